// lm_stream.hip -- gate/up + SwiGLU -> down_proj of a single-stream LM step as ONE launch whose weight stream runs ahead of the seam in the
// REGISTER FILE (the Qwen2 decoder layer's MLP inside the per-frame model call, /root/reference/test/inference.py:217,
// video_head_live_llava_qwen.py:156-188).  Round-6 experiment, tuning "engine" = 2, off by default: it measures what the two launches measure.
//
// lm_engine.hip (the LDS-DMA loader ring VERDICT r5 asked for) showed what bounds a launch with in-kernel hand-offs on this chip
// (profiles/r06_engine_mlp_stamps.txt): a seam costs 5.5-11.5 us of round trips whatever runs around it, and the weights a CU can have fetched by
// the time a seam resolves are capped by where they wait - 120 KiB of LDS = 5 us of its HBM share.  The register file is 512 KB per CU.  So
// this kernel is gemm_ws_kernel's own body (weights straight from HBM into VGPRs as the MFMA A operand, X through LDS, the same k-step order
// and split-K slices: the same bits) made persistent over the two GEMMs, with NS = 5 rotating weight register sets instead of three and the
// down_proj phase's first four chunks (7 waves x 32 KiB = 224 KiB per CU, 9 us of stream) issued BEFORE the workgroup waits for its slice of
// the activation.  The work split is the launches' own (237 five-pair workgroups for gate/up; 32 x 8 seven-tile workgroups for down_proj = 256).
//
// What the stamps taught (profiles/r06_engine_mlp_stamps.txt, second half): (1) one acquire per WORKGROUP - with all eight waves executing
// buffer_inv sc1 (~1,900 per seam chip-wide) the first X chunk behind a seam came 10 us late; (2) the polling wave must own no tiles - a wave with
// weight loads in flight sees its poll's answer only behind them; (3) the row phase does NOT belong in the launch: in-kernel it took 12.8 us until
// the first X chunk (rows under the other CUs' prefetch traffic 7, counter seen +4, X +1) against 4.8 us for resid_norm as its own launch plus
// ~2.5 of ramp; (4) with all that the launch is HBM-bound end to end - 407 MB in 68.9 us from the first workgroup's start to the last one's end,
// 5.9 TB/s - and the trace gives it 71.3 us against 45.7 + 25.7 for the two launches: the seam is covered by prefetch, and what it saves is what
// one dispatch + ramp costs, which is what the seam's own round trips give back.
//
// Hand-off (cdna_hip_programming.md G16): gate/up stores the activation write-through (sc1), drains, one lane adds its pairs to the counter of
// the down_proj slice(s) they belong to; the consuming workgroup's LAST wave (no tiles in either phase) polls that word relaxed and does the
// workgroup's one agent-scope acquire, everyone meets at the barrier, plain loads stage X.  Spins are bounded; a time-out sets the error word the
// heads kernel turns into NaN.
#include "aha_kernels.h"
#include <type_traits>

namespace {

constexpr int ST_THREADS = 512;
constexpr unsigned ST_SPIN_LIMIT = 1u << 18;

template <int MT, int KC>
struct StCfg {
    static constexpr int MPAD = MT * 16;
    static constexpr int STRIDE = KC * 32 + 8;                 // bf16 elements per LDS row (gemm_ws.hip: conflict-free ds_read_b128)
    static constexpr int BUF = MPAD * STRIDE;
    static constexpr int LDS_BYTES = 2 * BUF * 2;
    static constexpr int XCH = MPAD * KC * 4;                  // 16-byte pieces of one X chunk
    static constexpr int XLD = (XCH + ST_THREADS - 1) / ST_THREADS;
};

// The workgroup waits until *g >= target: lane 0 of wave 0 polls, wave 0 does the ONE agent-scope acquire (buffer_inv sc1 drops the CU's L1 and
// the stale L2 lines for every wave of the workgroup; a first version let all eight waves acquire: ~1,900 invalidates per seam chip-wide cost
// ~10 us before the first X chunk could be staged), then everyone meets at the barrier and loads.
// The polling wave is the LAST one, which owns no tiles in either phase: a wave with weight loads in flight sees its poll's answer only behind
// them (loads return in order), i.e. a prefetch's length after the hand-off happened.
static __device__ __forceinline__ void st_wait(unsigned* g, unsigned target, unsigned* gerr) {
    if (threadIdx.x >= ST_THREADS - 64) {
      if (threadIdx.x == ST_THREADS - 64) {
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 63u) == 0 && (spins > ST_SPIN_LIMIT || __hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_store(gerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // never hang the GPU: poisoned scores instead
                break;
            }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// f(integral_constant<0>) ... f(integral_constant<N-1>): compile-time step indices without a loop variable
template <int N, int I = 0, typename F>
static __device__ __forceinline__ void unroll_steps(F& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        unroll_steps<N, I + 1>(f);
    }
}
// the NS - 1 sets still loaded after R steps of a group are sets R, R+1, ... (mod NS); the last one has no next X chunk to stage
template <int R, int NS, int I = 0, typename F>
static __device__ __forceinline__ void unroll_drain(F& f) {
    if constexpr (I < NS - 1) {
        f(std::integral_constant<int, (R + I) % NS>{}, std::integral_constant<bool, I == NS - 2>{});
        unroll_drain<R, NS, I + 1>(f);
    }
}

// One GEMM phase of one workgroup: gemm_ws_body's arithmetic with NS rotating weight sets.  `wave_tile0` < 0: this wave owns no tiles in
// this phase (it still stages X and meets the barriers).  The first NS - 1 weight chunks are issued, then `between()` runs (the seam), then X.
// Needs n = c1 - c0 >= NS - 1 chunks (the host checks).
// SEAM: the phase's X is handed over inside the launch (prefetch, wait, then X); otherwise X is there at launch and its first chunk goes out FIRST,
// in front of the weight prologue (loads return in order: behind the prologue the first X chunk arrived ~3 us later than it had to).
template <int MT, int NT, int KC, int EPI, int NS, unsigned RMASK, bool SEAM, typename Between, typename After>
static __device__ __forceinline__ void st_phase(const GemmWsArgs& a, const int wave_tile0, const int by, bf16* xs, Between between, After after, unsigned long long* stamps) {
    using C = StCfg<MT, KC>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = lane >> 4, r16 = lane & 15;
    const bool active = wave_tile0 >= 0;
    const int tile0 = active ? wave_tile0 : 0;
    int tl[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) tl[j] = min(tile0 + j, a.n_tiles - 1);
    const int NC8 = a.KS / 8;
    const int c0 = (int)(((long)by * NC8) / a.S) * (8 / KC), c1 = (int)(((long)(by + 1) * NC8) / a.S) * (8 / KC);
    const int n = c1 - c0;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 w[NS][KC][NT];
    bf16x8 xr[C::XLD];

    auto load_w = [&](bf16x8 (&ws)[KC][NT], int c) {
        const int ks0 = c * KC;
#pragma unroll
        for (int i = 0; i < KC; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                ws[i][j] = __builtin_nontemporal_load(&a.Wp[((long)tl[j] * a.KS + (ks0 + i)) * 64 + lane]);
    };
    auto stage_load = [&](int c) {
        const int kbase = c * KC * 32;
#pragma unroll
        for (int i = 0; i < C::XLD; ++i) {
            const int idx = min(tid + i * ST_THREADS, C::XCH - 1);
            const int row = min(idx / (KC * 4), a.M - 1);
            const int k = min(kbase + (idx % (KC * 4)) * 8, a.Kx - 8);
            xr[i] = *reinterpret_cast<const bf16x8*>(a.X + (long)row * a.ldx + k);
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < C::XLD; ++i) {
            const int idx = min(tid + i * ST_THREADS, C::XCH - 1);
            const int row = idx / (KC * 4), cc = idx % (KC * 4);
            *reinterpret_cast<bf16x8*>(xs + buf * C::BUF + row * C::STRIDE + cc * 8) = xr[i];
        }
    };
    auto compute = [&](bf16x8 (&ws)[KC][NT], int buf) {
        const bf16* xb = xs + buf * C::BUF + r16 * C::STRIDE + q * 8;
#pragma unroll
        for (int i = 0; i < KC; ++i) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xb + m * 16 * C::STRIDE + i * 32);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[m][j] = mfma16(ws[i][j], xf, acc[m][j]);
            }
        }
    };

    // ---- the next NS - 1 chunks of this wave's weight stream go out BEFORE the seam: they do not depend on it
    if constexpr (!SEAM) stage_load(c0);
    // (paced: at most two chunks of a wave in flight.  Issued all at once, the chip's prologues are 38-57 MB in the memory system's queues, and
    // every poll of the hand-off counter, the row phase's slab loads and the first X chunk wait ~6 us behind them.)
    if (active) {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) {
            load_w(w[s], c0 + s);
            if (SEAM && s >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KC * NT) : "memory");
        }
    }
    between();
    if constexpr (SEAM) stage_load(c0);
    stage_store(0);
    __syncthreads();
    if (stamps && tid == 0) stamps[(long)blockIdx.x * 16 + (EPI == EPI_SWIGLU ? 6 : 9)] = wall_clock64();

    int c = c0, buf = 0;
    const int steady = n - (NS - 1);                            // steps that still prefetch (chunk c + NS - 1)
    if (active) {
        // straight-line groups of NS steps whose set indices are compile-time, then the remainder R = steady % NS as the first R steps of
        // one more group followed by the NS - 1 chunks left in the sets - one instantiation per R, chosen by a switch, so that no step ever
        // copies a register set (a copy would wait for the load just issued into it: a memory latency per step)
        auto step = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
            stage_load(c + 1);
            __builtin_amdgcn_sched_barrier(0);
            load_w(w[(S + NS - 1) % NS], c + NS - 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(w[S], buf);
            stage_store(buf ^ 1);
            __syncthreads();
            ++c; buf ^= 1;
        };
        auto drain = [&](auto sc, auto lastc) {
            constexpr int S = decltype(sc)::value;
            constexpr bool LAST = decltype(lastc)::value;
            if constexpr (!LAST) stage_load(c + 1);
            compute(w[S], buf);
            if constexpr (!LAST) {
                stage_store(buf ^ 1);
                __syncthreads();
            }
            ++c; buf ^= 1;
        };
        for (int g = steady / NS; g > 0; --g) unroll_steps<NS>(step);
        auto finish = [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            unroll_steps<R>(step);
            unroll_drain<R, NS>(drain);
        };
        // only the remainders in RMASK are instantiated (every instantiation costs registers around the switch); the host launches this kernel
        // only for shapes whose chunk counts give one of them (aha_lm_mlp_stream_ok)
        const int rem = steady % NS;
        if ((RMASK >> 0 & 1u) && rem == 0) finish(std::integral_constant<int, 0>{});
        if constexpr (NS > 1 && (RMASK >> 1 & 1u)) { if (rem == 1) finish(std::integral_constant<int, 1>{}); }
        if constexpr (NS > 2 && (RMASK >> 2 & 1u)) { if (rem == 2) finish(std::integral_constant<int, 2>{}); }
        if constexpr (NS > 3 && (RMASK >> 3 & 1u)) { if (rem == 3) finish(std::integral_constant<int, 3>{}); }
        if constexpr (NS > 4 && (RMASK >> 4 & 1u)) { if (rem == 4) finish(std::integral_constant<int, 4>{}); }
        if constexpr (NS > 5 && (RMASK >> 5 & 1u)) { if (rem == 5) finish(std::integral_constant<int, 5>{}); }
    } else {
        // a wave without tiles: stages X and meets the same n - 1 barriers
        for (int i = 0; i < n - 1; ++i) {
            stage_load(c + 1);
            stage_store(buf ^ 1);
            __syncthreads();
            ++c; buf ^= 1;
        }
    }

    // ---- epilogue: acc[m][j][e] <-> row m*16 + r16, column (tile0 + j)*16 + q*4 + e (gemm_ws_body.h)
    if (active && tile0 < a.n_tiles) {
        if constexpr (EPI == EPI_PARTIAL) {
            float* base = a.partial + (long)by * a.slab_stride;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = m * 16 + r16;
                if (row >= a.M) continue;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = (tile0 + j) * 16 + q * 4;
                    if (tile0 + j < a.n_tiles && col < a.ldp) *reinterpret_cast<f32x4*>(base + (long)row * a.ldp + col) = acc[m][j];
                }
            }
        } else {
            static_assert(EPI != EPI_SWIGLU || NT == 2, "swiglu epilogue needs gate/up tile pairs");
            const int col = (tile0 / 2) * 16 + q * 4;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = m * 16 + r16;
                if (row >= a.M || col >= a.N) continue;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = rbf(acc[m][0][e]);              // gate_proj output (bf16)
                    const float sg = rbf(g / (1.0f + __expf(-g)));  // silu output (bf16)
                    const float u = rbf(acc[m][NT - 1][e]);         // up_proj output (bf16)
                    o[e] = f2bf(sg * u);
                }
                store8_sc1(a.out, ((long)row * a.ldo + col) * 2, __builtin_bit_cast(unsigned long long, o));   // handed off inside the launch
            }
        }
    }
    after();
}

}  // namespace

#define ST_STAMP(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[(long)blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)

template <int MT>
__global__ __launch_bounds__(ST_THREADS, 1) void lm_mlp_stream_kernel(MlpStreamArgs p) {
    constexpr int NS_GU = 5, NS_DN = 5, KC_GU = 4, KC_DN = 8;
    constexpr unsigned RM_GU = 1u << 4, RM_DN = (1u << 0) | (1u << 1);   // Qwen2-7B: gate/up 28 chunks of 4 k-steps (24 prefetching steps = 4 groups of 5 + 4); down_proj 9 or 10 chunks of 8 (5 or 6 = one group + 0 or 1)
    extern __shared__ __attribute__((aligned(16))) char st_lds[];
    bf16* xs = reinterpret_cast<bf16*>(st_lds);
    const int b = blockIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned* gerr = reinterpret_cast<unsigned*>(p.err);
    // pairs (16 activation columns each) in front of down_proj's K slice s: its slices are placed in units of 8 k-steps (gemm_ws.hip)
    const int pairs_total = p.gu.n_tiles / 2, NC8d = p.dn.KS / 8;
    auto slice_lo = [&](int s) { return min(2 * 8 * (int)(((long)s * NC8d) / p.dn.S), pairs_total); };

    ST_STAMP(0);
    // ---- gate/up + SwiGLU: workgroup b < gu_blocks, waves 0..4 own pair b*5 + wave (gemm_ws_kernel<MT,2,4,SWIGLU,5>'s split)
    if (b < p.gu_blocks) {
        const int tile0 = (b * p.gu_wpb + wave) * 2;
        const int wt = (wave < p.gu_wpb && tile0 < p.gu.n_tiles) ? tile0 : -1;
        auto gu_after = [&] {
                // hand-off: everyone's write-through stores have left, then one lane tells the slices this workgroup fed
                ST_STAMP(7);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) {
                    const int p0 = b * p.gu_wpb, p1 = min(p0 + p.gu_wpb, pairs_total);     // this workgroup's pairs = activation columns 16 p0 .. 16 p1
                    for (int s = 0; s < p.dn.S; ++s) {                                     // a k-step of down_proj = 32 columns = 2 pairs
                        const int lo = max(p0, slice_lo(s)), hi = min(p1, slice_lo(s + 1));
                        if (hi > lo) __hip_atomic_fetch_add(p.sync + 32 * (1 + s), (unsigned)(hi - lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                ST_STAMP(8);
        };
        st_phase<MT, 2, KC_GU, EPI_SWIGLU, NS_GU, RM_GU, false>(p.gu, wt, 0, xs, [&] { ST_STAMP(1); }, gu_after, p.stamps);
    }
    // ---- down_proj: workgroup b < dn_bx * S: tile dn_wpb * (b % dn_bx) + wave, K slice b / dn_bx (gemm_ws_kernel<MT,1,*,PARTIAL,*>'s split by
    // (n-tile, slice); seven tiles per workgroup put its 224 x 8 wave tasks on exactly 256 CUs and leave the last wave free to poll)
    if (b < p.dn_bx * p.dn.S) {
        const int bx = b % p.dn_bx, by = b / p.dn_bx;
        const int tile0 = bx * p.dn_wpb + wave;
        st_phase<MT, 1, KC_DN, EPI_PARTIAL, NS_DN, RM_DN, true>(
            p.dn, (wave < p.dn_wpb && tile0 < p.dn.n_tiles) ? tile0 : -1, by, xs, [&] { ST_STAMP(2); st_wait(p.sync + 32 * (1 + by), (unsigned)(slice_lo(by + 1) - slice_lo(by)), gerr); ST_STAMP(5); }, [&] { ST_STAMP(10); }, p.stamps);
    }
    ST_STAMP(13);
    // The counters clean up after themselves: the last workgroup to get here (every workgroup of the grid passes, with or without work) zeroes
    // them for the next launch that uses this block.  [r6] A memset node in front of the step's graph did not do: under graph replay the hand-offs
    // intermittently read early or timed out (tools/diag/race_screen.py), never with direct launches.
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.sync + 32 * 15, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            for (int s = 0; s < p.dn.S; ++s) __hip_atomic_store(p.sync + 32 * (1 + s), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.sync + 32 * 15, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Shapes the instantiated pipeline depths serve (lm_mlp_stream_kernel: gate/up NS 5 / KC 4 with remainder 4; down_proj NS 5 / KC 8 with remainder
// 0 or 1 per K slice); everything else runs the three launches.
extern "C" int aha_lm_mlp_stream_ok(int gu_KS, int dn_KS, int dn_S) {
    const int n_gu = gu_KS / 4;
    if (gu_KS % 8 || dn_KS % 8 || n_gu < 4 || (n_gu - 4) % 5 != 4 || dn_S < 1 || dn_S > 14) return 0;
    const int NC8 = dn_KS / 8;
    for (int s = 0; s < dn_S; ++s) {
        const int n = (int)(((long)(s + 1) * NC8) / dn_S) - (int)(((long)s * NC8) / dn_S);
        if (n < 4 || (n - 4) % 5 > 1) return 0;
    }
    return 1;
}

extern "C" hipError_t aha_lm_mlp_stream(const MlpStreamArgs* p, int grid, hipStream_t st) {
    constexpr int LDS = StCfg<3, 8>::LDS_BYTES;                 // the larger of the two phases' X staging
    static bool attr_set = false;
    auto kern = lm_mlp_stream_kernel<3>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (p->M < 1 || p->M > 48 || grid < p->gu_blocks || grid < p->dn_bx * p->dn.S || grid < p->M || p->gu_wpb > 7 || p->dn_wpb > 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(ST_THREADS), LDS, st, *p);
    return hipGetLastError();
}
