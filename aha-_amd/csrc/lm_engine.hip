// lm_engine.hip -- the B = 1 decoder layer's MLP half as ONE persistent launch on an LDS-DMA loader ring:
//     resid_norm (o_proj slabs + residual + post-attention RMSNorm)  ->  gate/up GEMM + SwiGLU  ->  down GEMM (split-K slabs)
// (reference: the per-frame model call, /root/reference/test/inference.py:217, i.e. the Qwen2 decoder layer's second half,
// video_head_live_llava_qwen.py:156-188 -> transformers Qwen2DecoderLayer).
//
// Why: at one stream (M = 36 rows) the layer is seven launches of 6-45 us, each paying ~3 us of ramp plus a boundary, and the three
// glue launches run with HBM idle - 34 of the layer's 109 us (profiles/r05_bench_kernel_stats.csv).  A grid barrier inside one
// launch costs as much as the boundary it replaces (rounds 2-4 measured that with the register-streaming GEMM body, whose weight
// stream is only two chunks deep).  What is different here is WHO waits: the weights do not depend on any hand-off, so one wave per CU
// does nothing but stream this CU's share of the layer's packed weights into a ring of LDS slots (global_load_lds ... nt, counted
// vmcnt, FULL / FREE words in LDS) and never looks at a seam; the MFMA consumer waves and a second, tiny loader for the activation
// panels are the only ones that wait for other CUs.  While a seam's round trips are in flight the weight loader fills the ring
// (6 x 20 KiB per CU = 30 MB chip-wide, ~5 us of HBM time), and behind the seam the consumers drain it at LDS speed.
//
// Geometry (one workgroup per CU, 5 waves): wave 0 = weight loader, wave 1 = activation (X) loader, waves 2-4 = consumers, one 16-row
// tile of the M <= 48 rows each.  A ring slot = two 32-deep k-steps of ALL of this CU's n-tiles ([tile][2][1 KiB] in MFMA-fragment
// order, exactly the blocks of the packed weight Wp[n_tile][k_step][lane][8]); an X slot = the same two k-steps of the activation
// panels ([k-step][48 rows][32], 16-byte chunks XOR-swizzled on the DMA's source address so the B-fragment ds_read_b128 is
// conflict-free, as gemm_wl.hip).  Every consumer reads every weight block of a slot (its A fragments) and its own row tile of X,
// frees the slot as soon as the fragments are in registers, then issues the 2 * NT MFMAs.  Every output element accumulates its
// k-steps in order in one accumulator with gemm_ws_kernel's split-K slice boundaries: the same bits as the launches it replaces.
//
// Work split: the chip is G = split_down groups of CPG = CUs / G workgroups (group = blockIdx % G: one XCD under round-robin
// placement - a speed bonus only, every hand-off is correct under any placement).  Group g owns down_proj's K slice g (its CUs split
// the slice's n-tiles) and, to keep the hand-off local, the gate/up column pairs that produce that slice of the activation.
// Hand-offs (cdna_hip_programming.md G16): producers store write-through (sc1), drain vmcnt, and ONE lane adds to a counter
// (rows done; per-slice pairs done); the X loader of a consuming CU polls that one word relaxed, does one agent-scope acquire, then
// DMAs the panels.  Every spin is bounded: a time-out sets the error word the heads kernel turns into NaN scores, and every other
// wait then falls through, so the grid always drains.
#include "aha_kernels.h"
#include "resid_norm_body.h"

namespace {

constexpr int ENG_ROWS = 48;                                   // rows of a hand-off panel: [K/32][48][32]
constexpr int ENG_NTMAX = 10;                                  // n-tiles a CU carries per phase (accumulators: 4 VGPRs each)
constexpr int ENG_R = 6, ENG_RX = 4;                           // ring depths: weight slots, X slots
constexpr int ENG_WSLOT = ENG_NTMAX * 2048;                    // two k-steps of ten tiles
constexpr int ENG_XSLOT = 2 * ENG_ROWS * 64;                   // two k-steps of the 48-row panel
constexpr int ENG_X_OFF = ENG_R * ENG_WSLOT;
constexpr int ENG_FLAG_OFF = ENG_X_OFF + ENG_RX * ENG_XSLOT;
constexpr int ENG_ASG_OFF = ENG_FLAG_OFF + 512;                // this workgroup's EngAssign rows (copied once: a global load in a loader would drain its DMA queue)
constexpr int ENG_LDS = ENG_ASG_OFF + 4 * 48;
constexpr int ENG_NCW = 3;                                     // consumer waves = row tiles
constexpr int ENG_NWL = 2;                                     // weight-loader waves (they take alternate slots)
constexpr int ENG_THREADS = 64 * (ENG_NWL + 1 + ENG_NCW);
constexpr int ENG_NROW = ENG_NCW + 1;                          // waves that share a row of the row phase: the consumers and the X loader
// flag words (unsigned) inside the 512-byte flag block
constexpr int F_WFULL = 0, F_WFREE = 8, F_XFULL = 32, F_XFREE = 40, F_CNT = 64, F_RED = 80, F_ABORT = 96;
constexpr unsigned ENG_SPIN_LIMIT = 1u << 21;                  // ~0.3 s of s_sleep polls: never hang the GPU

typedef __attribute__((address_space(3))) void* lptr_t;

static __device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
static __device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct EngWave {
    unsigned* fl;           // LDS flag block
    unsigned* gerr;         // global error word (sticky; heads_kernel poisons the scores when set)
    int lane;
    __device__ __forceinline__ bool aborted() const { return __builtin_amdgcn_readfirstlane(lds_ld(fl + F_ABORT)) != 0; }
    __device__ __forceinline__ void give_up() const {
        if (lane == 0) {
            lds_st(fl + F_ABORT, 1u);
            __hip_atomic_store(gerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // wait until *p >= v (wrap-safe); all lanes read the same word
    __device__ __forceinline__ void lds_wait_ge(const unsigned* p, unsigned v) const {
        unsigned spins = 0;
        while ((int)(__builtin_amdgcn_readfirstlane(lds_ld(p)) - v) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0) {
                if (aborted()) break;
                if (spins > ENG_SPIN_LIMIT) { give_up(); break; }
            }
        }
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ bool lds_is_ge(const unsigned* p, unsigned v) const {
        return (int)(__builtin_amdgcn_readfirstlane(lds_ld(p)) - v) >= 0;
    }
    // wait until the global counter *g >= target, then ONE agent-scope acquire (the bytes it publishes were stored write-through)
    __device__ __forceinline__ void global_wait_ge(unsigned* g, unsigned target) const {
        unsigned spins = 0;
        while ((int)(__builtin_amdgcn_readfirstlane(__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 63u) == 0) {
                if (aborted() || __builtin_amdgcn_readfirstlane(__hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) { give_up(); break; }
                if (spins > (ENG_SPIN_LIMIT >> 3)) { give_up(); break; }      // a global poll is ~1 us
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // the consumer waves meet at counter k: returns true on the wave that arrived last
    __device__ __forceinline__ bool cons_arrive(int k, int n_waves = ENG_NCW) const {
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(fl + F_CNT + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)__builtin_amdgcn_readfirstlane(old) == n_waves - 1;
    }
    __device__ __forceinline__ void cons_sync(int k, int n_waves = ENG_NCW) const {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's LDS stores (red[]) are done
        (void)cons_arrive(k, n_waves);
        lds_wait_ge(fl + F_CNT + k, (unsigned)n_waves);
    }
};

// diagnostic time stamps (100 MHz wall clock), one row of 16 per workgroup; a.stamps is null in the product
static __device__ __forceinline__ void eng_stamp(const EngArgs& a, int slot, int lane) {
    if (a.stamps && lane == 0) a.stamps[(long)blockIdx.x * 16 + slot] = wall_clock64();
}

// s_waitcnt vmcnt(n) for a run-time n: the loaders' slots carry different DMA counts per phase
static __device__ __forceinline__ void wait_vm(int n) {
#define VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        VMC(0) VMC(2) VMC(4) VMC(6) VMC(8) VMC(10) VMC(12) VMC(14) VMC(16) VMC(18) VMC(20) VMC(22) VMC(24) VMC(26) VMC(28) VMC(30)
        VMC(32) VMC(34) VMC(36) VMC(38) VMC(40)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef VMC
}

// Two LDS-DMA wave-instructions: 2 KiB from sbase + voff (+ 1024) to LDS address ldsaddr (+ 1024), lane-linear.  M0 is written in
// the statement that reads it (cdna_hip_programming.md 5.7); s_nop 3: a VALU-written SGPR base needs 5 wait states before a VMEM reads it.
template <bool NT>
static __device__ __forceinline__ void dma2(const void* sbase, unsigned voff, unsigned ldsaddr) {
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\t"
                     "global_load_lds_dwordx4 %1, %2 nt\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// wave 0: the weight loader.  Walks this CU's slots of every GEMM phase in order and never waits for anything but a free ring slot.
// ---------------------------------------------------------------------------------------------------------------------------------
static __device__ void eng_w_loader(const EngArgs& a, const EngAssign* __restrict__ asg, const unsigned lds0, const EngWave w, const int me) {
    const unsigned voff = w.lane * 16;
    unsigned n = 0;                                             // slot index over all phases (this wave takes n % ENG_NWL == me)
    int q0 = 0, q1 = 0, q2 = 0, pend = 0;                       // DMA counts of this wave's issued, not yet published slots (oldest first)
    unsigned i0 = 0, i1 = 0, i2 = 0;                            // ... and their slot indices
    auto retire = [&] {                                         // publish the oldest: everything but the younger ones has landed
        wait_vm(q1 + q2);
        if (w.lane == 0) lds_st(w.fl + F_WFULL + i0 % ENG_R, i0 + 1);
        q0 = q1; q1 = q2; q2 = 0; i0 = i1; i1 = i2; --pend;
    };
    if (me == 0) eng_stamp(a, 0, w.lane);
    // a row workgroup's loaders start behind its row: the row's loads would queue behind this CU's own ring-fill burst
    if (a.rn.H > 0 && (int)blockIdx.x < a.M && !(a.exp & 1)) w.lds_wait_ge(w.fl + F_CNT + 1, ENG_NROW);
    for (int ph = 0; ph < a.n_gemm; ++ph) {
        const EngGemm& g = a.gemm[ph];
        const EngAssign as = asg[ph];
        const char* wb = reinterpret_cast<const char*>(g.Wp) + ((long)as.tile0 * g.KS + as.ks0) * 1024;
        const long tstride = (long)g.KS * 1024;
        const int ns = as.nk >> 1;
        for (int s = 0; s < ns; ++s, ++n) {
            if ((int)(n % ENG_NWL) != me) continue;
            const unsigned pos = n % ENG_R;
            if (n >= ENG_R) {
                const unsigned need = n - ENG_R + 1;
                unsigned spins = 0;
                while (!(w.lds_is_ge(w.fl + F_WFREE + pos * 3, need) && w.lds_is_ge(w.fl + F_WFREE + pos * 3 + 1, need) && w.lds_is_ge(w.fl + F_WFREE + pos * 3 + 2, need))) {
                    if (pend) { retire(); continue; }
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 63u) == 0) {
                        if (w.aborted()) break;
                        if (spins > ENG_SPIN_LIMIT) { w.give_up(); break; }
                    }
                }
                asm volatile("" ::: "memory");
            }
            const unsigned la = lds0 + pos * ENG_WSLOT;
            const char* src = wb + (long)s * 2048;
            for (int t = 0; t < as.nt; ++t) dma2<true>(src + t * tstride, voff, la + t * 2048);
            const int cnt = 2 * as.nt;
            if (pend == 0) { q0 = cnt; i0 = n; } else if (pend == 1) { q1 = cnt; i1 = n; } else { q2 = cnt; i2 = n; }
            ++pend;
            if (pend == 3) retire();
        }
        if (me == 0) eng_stamp(a, 1 + ph, w.lane);              // 1, 2: the phase's last slot is issued
    }
    while (pend) retire();
    if (me == 0) eng_stamp(a, 3, w.lane);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// wave 1: the activation loader.  Per GEMM phase: wait for the hand-off that publishes this CU's X panels, acquire, then DMA them two
// k-steps at a time (six 1-KiB pieces) into the X ring.
// ---------------------------------------------------------------------------------------------------------------------------------
static __device__ void eng_x_loader(const EngArgs& a, const EngAssign* __restrict__ asg, const unsigned lds0, const EngWave w) {
    // piece = 16 rows x 64 B; LDS position p (lane-linear) holds row p >> 2, chunk (p & 3) ^ g(row >> 2), g = (0, 3, 2, 1)
    const unsigned voff = (w.lane >> 2) * 64 + (((w.lane & 3) ^ ((4 - ((w.lane >> 4) & 3)) & 3)) * 16);
    unsigned n = 0;
    int pend = 0;
    auto retire = [&] {
        wait_vm(6 * (pend - 1));
        const unsigned idx = n - pend;
        if (w.lane == 0) lds_st(w.fl + F_XFULL + idx % ENG_RX, idx + 1);
        --pend;
    };
    for (int ph = 0; ph < a.n_gemm; ++ph) {
        const EngGemm& g = a.gemm[ph];
        const EngAssign as = asg[ph];
        const int ns = as.nk >> 1;
        if (ns == 0) continue;
        while (pend) retire();                                  // the poll below drains vmcnt anyway: publish first
        w.global_wait_ge(a.sync + as.ready_idx * 32, (unsigned)(as.ready_target < 0 ? a.M : as.ready_target));
        eng_stamp(a, 4 + ph, w.lane);                           // 4, 5: this phase's X is published
        const char* xb = reinterpret_cast<const char*>(g.Xkb) + (long)as.ks0 * (ENG_ROWS * 64);
        for (int s = 0; s < ns; ++s) {
            const unsigned pos = n % ENG_RX;
            if (n >= ENG_RX) {
                const unsigned need = n - ENG_RX + 1;
                unsigned spins = 0;
                while (!(w.lds_is_ge(w.fl + F_XFREE + pos * 3, need) && w.lds_is_ge(w.fl + F_XFREE + pos * 3 + 1, need) && w.lds_is_ge(w.fl + F_XFREE + pos * 3 + 2, need))) {
                    if (pend) { retire(); continue; }
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 63u) == 0) {
                        if (w.aborted()) break;
                        if (spins > ENG_SPIN_LIMIT) { w.give_up(); break; }
                    }
                }
                asm volatile("" ::: "memory");
            }
            const unsigned la = lds0 + ENG_X_OFF + pos * ENG_XSLOT;
            const char* src = xb + (long)s * ENG_XSLOT;
#pragma unroll
            for (int i = 0; i < 3; ++i) dma2<false>(src + i * 2048, voff, la + i * 2048);
            ++pend; ++n;
            if (pend == 3) retire();
        }
    }
    while (pend) retire();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// the consumer waves
// ---------------------------------------------------------------------------------------------------------------------------------
struct EngCons {
    unsigned n;             // slots consumed so far (the W ring and the X ring advance together)
    int k_sync;             // next consumer-sync counter
};

// A consumer wave owns NTW of the CU's n-tiles (all three row tiles of them): it reads only those tiles' weight blocks of a slot and
// the whole X slot.  (A first version gave each wave one row tile of ALL tiles: every weight block was then read three times and the
// CU's LDS pipe - 66 KiB of reads per 20-KiB slot beside the DMA writes - capped the CU at ~26 GB/s, below what two loaders deliver.)
template <int NTW>
static __device__ __forceinline__ void eng_gemm_phase(const EngArgs& a, const EngGemm& g, const EngAssign& as, const char* lds, const EngWave& w,
                                                      const int cw, const int t0w, EngCons& st, const int ph) {
    const int lane = w.lane, q = lane >> 4, r16 = lane & 15;
    constexpr int NA = NTW > 0 ? NTW : 1;
    f32x4 acc[NA][3];
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int xslot = (r16 * 4 + (q ^ ((4 - (r16 >> 2)) & 3))) * 16;        // byte offset of this lane's B fragment inside a 16-row piece
    const int ns = as.nk >> 1;
    for (int s = 0; s < ns; ++s) {
        const unsigned n = st.n, pos = n % ENG_R, xpos = n % ENG_RX;
        w.lds_wait_ge(w.fl + F_WFULL + pos, n + 1);
        w.lds_wait_ge(w.fl + F_XFULL + xpos, n + 1);
        if (s == 0 && cw == 0) eng_stamp(a, 6 + 3 * ph, lane);              // 6, 9: first slot in hand
        if constexpr (NTW > 0) {
            const char* wb = lds + pos * ENG_WSLOT + t0w * 2048 + lane * 16;
            const char* xb = lds + ENG_X_OFF + xpos * ENG_XSLOT + xslot;
            bf16x8 xf[2][3], wf[NTW][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 3; ++m) xf[kk][m] = *reinterpret_cast<const bf16x8*>(xb + kk * (ENG_ROWS * 64) + m * 1024);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                wf[t][0] = *reinterpret_cast<const bf16x8*>(wb + t * 2048);
                wf[t][1] = *reinterpret_cast<const bf16x8*>(wb + t * 2048 + 1024);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the fragments are in registers: hand the slots back
            if (lane == 0) {
                lds_st(w.fl + F_WFREE + pos * 3 + cw, n + 1);
                lds_st(w.fl + F_XFREE + xpos * 3 + cw, n + 1);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc[t][m] = mfma16(wf[t][kk], xf[kk][m], acc[t][m]);
        } else if (lane == 0) {
            lds_st(w.fl + F_WFREE + pos * 3 + cw, n + 1);
            lds_st(w.fl + F_XFREE + xpos * 3 + cw, n + 1);
        }
        st.n = n + 1;
    }
    if (cw == 0) eng_stamp(a, 7 + 3 * ph, lane);                            // 7, 10: last slot consumed
    // ---- epilogue: acc[t][m][e] <-> row m*16 + r16, column (tile0 + t0w + t)*16 + q*4 + e
    if (g.epi == EPI_SWIGLU) {
        if constexpr (NTW > 0 && NTW % 2 == 0) {
#pragma unroll
            for (int p = 0; p < NTW / 2; ++p) {
                const int col = (((as.tile0 + t0w) >> 1) + p) * 16 + q * 4;
                if (col >= g.out_cols) continue;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int row = m * 16 + r16;
                    if (row >= a.M) continue;
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gg = rbf(acc[2 * p][m][e]);                // gate_proj output (bf16)
                        const float sg = rbf(gg / (1.0f + __expf(-gg)));       // silu output (bf16)
                        const float u = rbf(acc[2 * p + 1][m][e]);             // up_proj output (bf16)
                        o[e] = f2bf(sg * u);
                    }
                    store8_sc1(g.out_kb, (((long)(col >> 5) * ENG_ROWS + row) * 32 + (col & 31)) * 2, __builtin_bit_cast(unsigned long long, o));
                }
            }
        }
        // hand-off: every consumer drains its write-through stores, the last one to arrive signals the slices it fed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool last = w.cons_arrive(st.k_sync++);
        if (last && lane == 0) {
            if (as.sig0_cnt) __hip_atomic_fetch_add(a.sync + as.sig0_idx * 32, (unsigned)as.sig0_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (as.sig1_cnt) __hip_atomic_fetch_add(a.sync + as.sig1_idx * 32, (unsigned)as.sig1_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) eng_stamp(a, 8 + 3 * ph, lane);                           // 8: outputs published
    } else if constexpr (NTW > 0) {                                         // EPI_PARTIAL: this slice's slab
        float* base = g.partial + (long)as.slice * g.slab_stride;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = (as.tile0 + t0w + t) * 16 + q * 4;
            if (col >= g.ldp) continue;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int row = m * 16 + r16;
                if (row < a.M) *reinterpret_cast<f32x4*>(base + (long)row * g.ldp + col) = acc[t][m];
            }
        }
    }
}

// One row of resid_norm by four waves (the consumers and the X loader, which has nothing to do before the row exists): resid_norm_row's
// arithmetic and reduction order - a "virtual wave" v = chunks 64v .. 64v+63 is reduced by the same butterfly and the virtual waves'
// sums are added in order.  rw = 0..3; a wave takes virtual waves rw and rw + 4 and issues every load of both before any arithmetic
// (the row phase is one memory round trip long, and HBM idles once the weight ring is full).
static __device__ __forceinline__ void eng_row(const EngArgs& a, const int row, const EngWave& w, const int rw) {
    const ResidNormArgs& r = a.rn;
    const int nch = r.H >> 3, nv = (nch + 63) >> 6, S = r.S;
    float* red = reinterpret_cast<float*>(w.fl + F_RED);
    f32x4 p0[2][8], p1[2][8];
    bf16x8 wv[2], hh[2];
    bool valid[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int v = rw + ENG_NROW * u, c = min(v * 64 + w.lane, nch - 1);   // clamped: the loads are unconditional, surplus lanes discard them
        valid[u] = v < nv && v * 64 + w.lane < nch;
        const float* p = r.partial + (long)row * r.ldp + c * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long so = (long)min(j, S - 1) * r.slab_stride;
            p0[u][j] = *reinterpret_cast<const f32x4*>(p + so);
            p1[u][j] = *reinterpret_cast<const f32x4*>(p + so + 4);
        }
        hh[u] = *reinterpret_cast<const bf16x8*>(r.h + (long)row * r.ldh + c * 8);
        wv[u] = *reinterpret_cast<const bf16x8*>(r.w + c * 8);
    }
    if (rw == 0) eng_stamp(a, 14, w.lane);
    float f8[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int v = rw + ENG_NROW * u, c = v * 64 + w.lane;
        float lin[8], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) lin[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < S) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { lin[e] += p0[u][j][e]; lin[4 + e] += p1[u][j][e]; }
            }
        bf16x8 ho;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = rbf(bf2f(hh[u][e]) + rbf(lin[e]));
            f8[u][e] = f;
            ho[e] = f2bf(f);
            ss = __builtin_fmaf(f, f, ss);
        }
        if (valid[u]) *reinterpret_cast<bf16x8*>(r.h + (long)row * r.ldh + c * 8) = ho;
        else ss = 0.f;
        ss = wave_sum(ss);
        if (v < nv && w.lane == 0) red[v] = ss;
    }
    w.cons_sync(0, ENG_NROW);
    if (rw == 0) eng_stamp(a, 15, w.lane);
    float t = 0.f;
    for (int i = 0; i < nv; ++i) t += red[i];
    const float rstd = resid_rstd(r, t);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = (rw + ENG_NROW * u) * 64 + w.lane;
        if (valid[u]) {
            const bf16x8 o = resid_scale(f8[u], wv[u], rstd);
            const long xo = ((long)(c >> 2) * ENG_ROWS + row) * 32 + (c & 3) * 8;
            store16_sc1(r.xn, xo * 2, __builtin_bit_cast(u32x4_t, o));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool last = w.cons_arrive(1, ENG_NROW);
    if (last && w.lane == 0) __hip_atomic_fetch_add(a.sync + a.rows_idx * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (last) eng_stamp(a, 12, w.lane);
}

}  // namespace

__global__ __launch_bounds__(ENG_THREADS, 1) void lm_engine_kernel(EngArgs a) {
    extern __shared__ __attribute__((aligned(16))) char eng_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned* fl = reinterpret_cast<unsigned*>(eng_lds + ENG_FLAG_OFF);
    static_assert(sizeof(EngAssign) == 48, "EngAssign layout");
    if (threadIdx.x < 128) fl[threadIdx.x] = 0u;
    if (threadIdx.x >= 128 && threadIdx.x < 128 + 12 * a.n_gemm) {            // 12 words per phase row
        const int i = threadIdx.x - 128, ph = i / 12, wd = i % 12;
        reinterpret_cast<int*>(eng_lds + ENG_ASG_OFF)[i] = reinterpret_cast<const int*>(a.asg + (long)ph * a.grid + blockIdx.x)[wd];
    }
    __syncthreads();
    const EngWave w{fl, reinterpret_cast<unsigned*>(a.err), lane};
    const EngAssign* asg = reinterpret_cast<const EngAssign*>(eng_lds + ENG_ASG_OFF);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)eng_lds);
    const bool has_row = a.rn.H > 0 && (int)blockIdx.x < a.M;
    if (wave < ENG_NWL) {
        eng_w_loader(a, asg, lds0, w, wave);
    } else if (wave == ENG_NWL) {
        if (has_row) eng_row(a, blockIdx.x, w, ENG_NCW);
        eng_x_loader(a, asg, lds0, w);
    } else {
        const int cw = wave - ENG_NWL - 1;
        EngCons st{0u, 2};                                      // consumer-sync counters 0 and 1 belong to the row phase
        if (has_row) eng_row(a, blockIdx.x, w, cw);
        for (int ph = 0; ph < a.n_gemm; ++ph) {
            const EngGemm& g = a.gemm[ph];
            const EngAssign as = asg[ph];
            if (as.nk == 0) continue;
            // this wave's share of the CU's tiles: whole gate/up pairs for the SwiGLU epilogue, single tiles otherwise
            const int unit = g.epi == EPI_SWIGLU ? 2 : 1, nu = as.nt / unit;
            const int t0w = (cw * nu / ENG_NCW) * unit, ntw = ((cw + 1) * nu / ENG_NCW) * unit - t0w;
            switch (ntw) {
                case 0: eng_gemm_phase<0>(a, g, as, eng_lds, w, cw, t0w, st, ph); break;
                case 1: eng_gemm_phase<1>(a, g, as, eng_lds, w, cw, t0w, st, ph); break;
                case 2: eng_gemm_phase<2>(a, g, as, eng_lds, w, cw, t0w, st, ph); break;
                case 3: eng_gemm_phase<3>(a, g, as, eng_lds, w, cw, t0w, st, ph); break;
                default: eng_gemm_phase<4>(a, g, as, eng_lds, w, cw, t0w, st, ph); break;
            }
        }
        if (cw == 0) eng_stamp(a, 13, lane);
    }
    // the counters clean up after themselves: the last workgroup to get here zeroes them for the next launch on this block (a memset node in
    // front of the step's graph did not do under graph replay: tools/diag/race_screen.py)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.sync + 32 * 15, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            for (int s = 0; s < 15; ++s) __hip_atomic_store(a.sync + 32 * s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 32 * 15, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" int aha_lm_engine_lds_bytes() { return ENG_LDS; }
extern "C" int aha_lm_engine_rows() { return ENG_ROWS; }
extern "C" int aha_lm_engine_ntmax() { return ENG_NTMAX; }

extern "C" hipError_t aha_lm_engine(const EngArgs* a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)lm_engine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ENG_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (a->M < 1 || a->M > ENG_ROWS || a->n_gemm < 1 || a->n_gemm > 4 || a->grid < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lm_engine_kernel, dim3(a->grid), dim3(ENG_THREADS), ENG_LDS, st, *a);
    return hipGetLastError();
}
