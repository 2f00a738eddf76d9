// Mid-M weight-streaming GEMM for the batched LM step:  Y[M,N] = X[M,K] * W[N,K]^T,  128 < M <= 320
// (BASELINE configs[3]: 8 streams x 36 tokens = 288 rows per weight pass; static-cache frame batching).
//
// gemm_ws.hip keeps a wave's weights in registers (three rotating sets) and stages only X through LDS.  That is the right
// shape for M <= 128, but at 18 row tiles the 36 accumulator tiles of a wave leave room for ONE k-step of weights per
// set: ~50 KB in flight per CU, i.e. ~25-33 GB/s of operand ingest per CU against the ~60 a CU can take in, and the wave
// re-reads 18 X fragments from LDS for every 36 MFMAs (LDS time = MFMA time).  Measured: gate/up at M = 288 116 us =
// 0.27 of the MFMA peak and 0.29 of HBM at an intensity on the ridge.  What bounds this regime is the per-CU ingest:
// with all M rows in one workgroup a CU takes in  K * 2 B * (M + its columns)  = 3.2 MB for gate/up at 237 workgroups
// (the decomposition is already near-square, so no tiling lowers it) - the kernel has to keep >= 100 KB of it in flight.
//
// So here BOTH operands go through LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers), five stages of one
// k-step (32 deep) each = four k-steps (~115 KB) in flight per CU behind a counted vmcnt and ONE raw s_barrier per k-step:
//   * W is already stored in MFMA-fragment order (Wp[n_tile][k_step][lane][8 bf16], gemm_ws.hip): a (tile, k-step) block
//     is 1 KiB contiguous, so one wave-instruction DMAs it lane-linear into LDS and every wave that needs it reads its
//     A fragment back with one conflict-free ds_read_b128 at lane*16.  Each weight byte still leaves HBM exactly once
//     (nontemporal), but is now shared by the workgroup's two row-halves instead of being held per wave.
//   * X rows (64 B per k-step) are DMAed as 1-KiB blocks of 16 rows x 4 chunks with the chunk index XOR-swizzled on the
//     SOURCE address (an LDS-DMA writes lane-linear) so that the B-fragment ds_read_b128 is conflict-free:
//     slot(row, c) = row*4 + (c ^ g(row >> 2)),  g = (0,3,2,1): within each of ds_read_b128's four 16-lane groups the
//     16 slots then fall on 16 different bank quads (checked against MI355X_MICROARCH.md section LDS).
//   * waves: 2 (row halves) x WN (column pairs); a wave owns MT/2 row tiles x 2 n-tiles (18 accumulator tiles at
//     M = 288): 11 fragment reads per 18 MFMAs.  WN = 5 for gate/up: 2368 n-tiles / 10 = 237 workgroups, one per CU
//     (296 eight-tile workgroups would need a second, 16 % full round); WN = 4 (8 tiles) for the split-K GEMMs.
// Every output element accumulates its k-steps in the same order, in one accumulator chain, with the same split-K slice
// boundaries as gemm_ws_kernel, so a batched step stays bit-identical to the same rows stepped alone
// (tests/test_gpu_parity.py::test_row_blocks_above_256_stay_bit_identical, test_full_size_batched_streams...).
#include "aha_kernels.h"

#ifndef AHA_WL_ABLATE
#define AHA_WL_ABLATE 0        // diagnostic builds only (tools/diag/wl_ablate.sh): 1 no MFMA, 2 no result store, 4 no W DMA, 8 no X DMA
#endif
template <int MT, int WN, int EPI, int STAGES>
__global__ __launch_bounds__(128 * WN) void gemm_wl_kernel(GemmWsArgs a) {
    constexpr int ABL = AHA_WL_ABLATE;
    constexpr int NW = 2 * WN, NTB = 2 * WN, MH = MT / 2;          // waves, n-tiles per workgroup, row tiles per wave
    constexpr int NB = MT + NTB;                                    // 1-KiB blocks per stage: MT of X, then NTB of W
    constexpr int PX = (MT + NW - 1) / NW, P = PX + 1;              // DMA wave-instructions per wave per stage: PX X blocks + its one W block
    constexpr int STAGE = NB * 512;                                 // bf16 elements per stage
    static_assert(MT % 2 == 0 && STAGES >= 4 && (STAGES - 2) * P <= 63, "geometry");
    static_assert(EPI == EPI_PARTIAL || EPI == EPI_SWIGLU, "mid-M kernel: split-K slabs or fused SwiGLU");
    extern __shared__ __attribute__((aligned(16))) char wl_smem[];
    bf16* lds = reinterpret_cast<bf16*>(wl_smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA bases are scalars
    const int q = lane >> 4, r16 = lane & 15;
    const int wm = wave / WN, wn = wave % WN;
    const int tile0 = blockIdx.x * NTB + wn * 2;                    // this wave's first n-tile
    // split-K slice in units of 8 k-steps, as gemm_ws_kernel places it (independent of M and of the tile configuration;
    // the products fit 32 bits: at most 16 slices x KS/8 chunks)
    const int NC8 = a.KS / 8;
    const int ks0 = (((int)blockIdx.y * NC8) / a.S) * 8, ks1 = ((((int)blockIdx.y + 1) * NC8) / a.S) * 8;
    const int nk = ks1 - ks0;
    const int ksx_last = a.Kx / 32 - 1;                             // X columns exist up to here; later k-steps meet zero weights

    // ---- per-lane DMA sources.  Wave w stages X blocks w, w + NW, .. (rows 16*o .. +15; surplus ops re-load block MT-1 and
    // surplus rows row M-1: identical bytes to the same place) and W block w = n-tile blockIdx.x*NTB + w (clamped: surplus
    // tiles of the last workgroup re-load the last tile and skip the store).
    const bf16* xsrc[PX];
    int xblk[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        xblk[i] = min(wave + i * NW, MT - 1);
        const int row = min(xblk[i] * 16 + (lane >> 2), a.M - 1);
        const int c = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
        xsrc[i] = a.X + (long)row * (a.xkb ? 32 : a.ldx) + c * 8;
    }
    const long xkstride = a.xkb ? (long)a.xkb * 32 : 32;            // elements between consecutive k-steps of one row's 64-byte piece
    const bf16* wsrc = reinterpret_cast<const bf16*>(a.Wp + ((long)min(blockIdx.x * NTB + wave, a.n_tiles - 1) * a.KS) * 64 + lane);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma = [&](int kt, int stage) {
        const int ks = ks0 + min(kt, nk - 1);                       // past the end: refill a dead stage (keeps the vmcnt counts fixed)
        const long xk = min(ks, ksx_last) * xkstride;
        bf16* sa = lds + stage * STAGE;
        if constexpr (!(ABL & 8)) {
#pragma unroll
            for (int i = 0; i < PX; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i] + xk), (lptr_t)(sa + xblk[i] * 512), 16, 0, 0);
        }
        if constexpr (!(ABL & 4))
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (long)ks * 512), (lptr_t)(sa + (MT + wave) * 512), 16, 0, 2);   // weights: read once, nontemporal
    };

    f32x4 acc[MH][2];
#pragma unroll
    for (int i = 0; i < MH; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
    const int xslot = (r16 * 4 + (q ^ ((4 - (r16 >> 2)) & 3))) * 8;   // element offset of this lane's B fragment inside an X block
    const int xoff = wm * MH * 512 + xslot, woff = (MT + wn * 2) * 512 + lane * 8;
    bf16x8 xf[MH], w0, w1;
    auto read_w = [&](int stage, bf16x8& a0, bf16x8& a1) {
        const bf16* sa = lds + stage * STAGE + woff;
        a0 = *reinterpret_cast<const bf16x8*>(sa);
        a1 = *reinterpret_cast<const bf16x8*>(sa + 512);
    };
    // One k-step: wait (this wave's blocks of k-step kt landed) -> barrier -> DMA k-step kt+STAGES-1 into the stage read last
    // k-step -> all 11 fragment reads -> 18 MFMAs.  Two other schedules were A/B-ed on MI355X and tie within noise
    // (profiles/r02_gemm_wl_ablation.txt): fragment reads software-pipelined one k-step ahead in place, and the DMA pieces
    // spread between the MFMAs.  The ablations there show why: the MFMAs alone take 60 of the 92 us (10 waves sit 3/3/2/2
    // on the 4 SIMDs and the chip holds ~1.6 GHz under the load); barrier, reads and DMA add 10 + 15 + 15.
    if (nk > 0) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s) dma(s, s);
        int st_cur = 0, st_new = STAGES - 1;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * P) : "memory");   // this wave's blocks of k-step kt have landed
            __builtin_amdgcn_s_barrier();                            // everyone's have; everyone is done reading the stage refilled next
            dma(kt + STAGES - 1, st_new);
            read_w(st_cur, w0, w1);
#pragma unroll
            for (int i = 0; i < MH; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(lds + st_cur * STAGE + xoff + i * 512);
            __builtin_amdgcn_sched_barrier(0);                      // keep the reads ahead of the MFMAs (hipcc otherwise reads 2, waits, issues 4)
            if constexpr (ABL & 1) {
#pragma unroll
                for (int i = 0; i < MH; ++i) { acc[i][0][0] += (float)xf[i][0] * (float)w0[0]; acc[i][1][0] += (float)xf[i][1] * (float)w1[0]; }
            } else {
#pragma unroll
                for (int i = 0; i < MH; ++i) {
                    acc[i][0] = mfma16(w0, xf[i], acc[i][0]);
                    acc[i][1] = mfma16(w1, xf[i], acc[i][1]);
                }
            }
            st_cur = st_cur == STAGES - 1 ? 0 : st_cur + 1;
            st_new = st_new == STAGES - 1 ? 0 : st_new + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing may still target LDS when the workgroup retires
    }

    // ---- epilogue (gemm_ws_body.h): acc[i][j][e] <-> row (wm*MH + i)*16 + r16, column (tile0 + j)*16 + q*4 + e
    if (tile0 >= a.n_tiles) return;
    if constexpr (ABL & 2) {                                        // keeps the arithmetic alive, never stores
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MH; ++i) s += acc[i][0][0] + acc[i][0][1] + acc[i][0][2] + acc[i][0][3] + acc[i][1][0] + acc[i][1][1] + acc[i][1][2] + acc[i][1][3];
        if (s != 12345.678f) return;
    }
    if constexpr (EPI == EPI_PARTIAL) {
        float* base = a.partial + (long)blockIdx.y * a.slab_stride;
#pragma unroll
        for (int i = 0; i < MH; ++i) {
            const int row = (wm * MH + i) * 16 + r16;
            if (row >= a.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = (tile0 + j) * 16 + q * 4;
                if (tile0 + j < a.n_tiles && col < a.ldp) *reinterpret_cast<f32x4*>(base + (long)row * a.ldp + col) = acc[i][j];
            }
        }
    } else {
        // tile0 = gate tile, tile0 + 1 = up tile of the same 16 output columns (weights interleaved at load)
        const int col = (tile0 / 2) * 16 + q * 4;
#pragma unroll
        for (int i = 0; i < MH; ++i) {
            const int row = (wm * MH + i) * 16 + r16;
            if (row >= a.M || col >= a.N) continue;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = rbf(acc[i][0][e]);                  // gate_proj output (bf16)
                const float sg = rbf(g / (1.0f + __expf(-g)));      // silu output (bf16)
                const float u = rbf(acc[i][1][e]);                  // up_proj output (bf16)
                o[e] = f2bf(sg * u);
            }
            bf16* dst = a.okb ? a.out + ((long)(col >> 5) * a.okb + row) * 32 + (col & 31) : a.out + (long)row * a.ldo + col;
            *reinterpret_cast<bf16x4*>(dst) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Gate/up + SwiGLU at 18 row tiles (M = 273..288: BASELINE configs[3]'s 8 streams x 36 tokens, and the 8-frame static batch):
// the same staging and the same per-element arithmetic as gemm_wl_kernel<18, 5, EPI_SWIGLU>, with the 90 (row tile, gate/up
// pair) units of a k-step dealt to the ten waves by where the waves SIT.  A workgroup's waves go round the four SIMDs in
// launch order (MI355X_MICROARCH.md, LDS section), so waves {0,4,8} and {1,5,9} share a SIMD three at a time and {2,6}, {3,7}
// two at a time: with nine units per wave the three-wave SIMDs carry 54 MFMAs per k-step and set the pace (the round-2
// ablation: MFMA-only time 60 of 90 us).  Here the three-wave SIMDs' waves take 7 units each (7 rows x 1 pair: 42 MFMAs per
// SIMD) and the two-wave SIMDs' waves 12 (6 rows x 2 pairs, or 4 rows x 3 pairs: 48 per SIMD); fragment reads drop from 110
// to 94 per k-step.  Every output element still sums its k-steps in the same order in one accumulator: bit-identical.
//   wave 2: rows  0-5  pairs 0-1     wave 3: rows 6-11 pairs 0-1     wave 6: rows 12-17 pairs 0-1     wave 7: rows 0-3 pairs 2-4
//   waves 0,4,8: rows 4-10 of pair 2,3,4         waves 1,5,9: rows 11-17 of pair 2,3,4
// ---------------------------------------------------------------------------------------------
template <int NR, int NP>
static __device__ __forceinline__ void wl_bal_body(const GemmWsArgs& a, bf16* lds, const int row0, const int pair0, const int lane, const int wave) {
    constexpr int MT = 18, WN = 5, STAGES = 5, NW = 10, NTB = 10;
    constexpr int NB = MT + NTB, PX = 2, P = PX + 1, STAGE = NB * 512;
    const int q = lane >> 4, r16 = lane & 15;
    const int KS = a.KS, ksx_last = a.Kx / 32 - 1;                  // S = 1: the whole K range
    const bf16* xsrc[PX];
    int xblk[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        xblk[i] = min(wave + i * NW, MT - 1);
        const int row = min(xblk[i] * 16 + (lane >> 2), a.M - 1);
        const int c = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
        xsrc[i] = a.X + (long)row * (a.xkb ? 32 : a.ldx) + c * 8;
    }
    const long xkstride = a.xkb ? (long)a.xkb * 32 : 32;
    const bf16* wsrc = reinterpret_cast<const bf16*>(a.Wp + ((long)min((int)blockIdx.x * NTB + wave, a.n_tiles - 1) * KS) * 64 + lane);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma = [&](int kt, int stage) {
        const int ks = min(kt, KS - 1);
        const long xk = min(ks, ksx_last) * xkstride;
        bf16* sa = lds + stage * STAGE;
        if constexpr (!(AHA_WL_ABLATE & 8)) {
#pragma unroll
            for (int i = 0; i < PX; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i] + xk), (lptr_t)(sa + xblk[i] * 512), 16, 0, 0);
        }
        if constexpr (!(AHA_WL_ABLATE & 4))
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (long)ks * 512), (lptr_t)(sa + (MT + wave) * 512), 16, 0, 2);
    };
    f32x4 acc[NR][NP][2];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) { acc[i][p][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][p][1] = acc[i][p][0]; }
    const int xslot = (r16 * 4 + (q ^ ((4 - (r16 >> 2)) & 3))) * 8;
    const int xoff = row0 * 512 + xslot, woff = (MT + pair0 * 2) * 512 + lane * 8;
    bf16x8 xf[NR], wf[NP][2];
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) dma(s, s);
    int st_cur = 0, st_new = STAGES - 1;
    for (int kt = 0; kt < KS; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * P) : "memory");
        __builtin_amdgcn_s_barrier();
        dma(kt + STAGES - 1, st_new);
        const bf16* sa = lds + st_cur * STAGE;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            wf[p][0] = *reinterpret_cast<const bf16x8*>(sa + woff + p * 1024);
            wf[p][1] = *reinterpret_cast<const bf16x8*>(sa + woff + p * 1024 + 512);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sa + xoff + i * 512);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (AHA_WL_ABLATE & 1) {
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) { acc[i][p][0][0] += (float)xf[i][0] * (float)wf[p][0][0]; acc[i][p][1][0] += (float)xf[i][1] * (float)wf[p][1][0]; }
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    acc[i][p][0] = mfma16(wf[p][0], xf[i], acc[i][p][0]);
                    acc[i][p][1] = mfma16(wf[p][1], xf[i], acc[i][p][1]);
                }
        }
        st_cur = st_cur == STAGES - 1 ? 0 : st_cur + 1;
        st_new = st_new == STAGES - 1 ? 0 : st_new + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue: SwiGLU of (gate tile, up tile) pairs, as gemm_wl_kernel
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int tile0 = (int)blockIdx.x * NTB + (pair0 + p) * 2;
        if (tile0 >= a.n_tiles) continue;
        const int col = (tile0 / 2) * 16 + q * 4;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = (row0 + i) * 16 + r16;
            if (row >= a.M || col >= a.N) continue;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = rbf(acc[i][p][0][e]);
                const float sg = rbf(g / (1.0f + __expf(-g)));
                const float u = rbf(acc[i][p][1][e]);
                o[e] = f2bf(sg * u);
            }
            bf16* dst = a.okb ? a.out + ((long)(col >> 5) * a.okb + row) * 32 + (col & 31) : a.out + (long)row * a.ldo + col;
            *reinterpret_cast<bf16x4*>(dst) = o;
        }
    }
}

__global__ __launch_bounds__(640) void gemm_wl_bal18_kernel(GemmWsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wl_smem[];
    bf16* lds = reinterpret_cast<bf16*>(wl_smem);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {                                                     // wave-uniform; every path runs the same barriers
        case 2: wl_bal_body<6, 2>(a, lds, 0, 0, lane, wave); break;
        case 3: wl_bal_body<6, 2>(a, lds, 6, 0, lane, wave); break;
        case 6: wl_bal_body<6, 2>(a, lds, 12, 0, lane, wave); break;
        case 7: wl_bal_body<4, 3>(a, lds, 0, 2, lane, wave); break;
        default: wl_bal_body<7, 1>(a, lds, (wave & 1) ? 11 : 4, 2 + (wave >> 2), lane, wave); break;   // waves 0,1,4,5,8,9
    }
}
static int g_wl_bal = 1;         // tuning "wl_bal": SIMD-balanced unit deal of the 18-row-tile gate/up kernel (0: nine units per wave)
extern "C" void aha_gemm_wl_set_balanced(int on) { g_wl_bal = on; }

template <int MT, int WN, int EPI>
static hipError_t launch_wl(const GemmWsArgs& a, hipStream_t st) {
    constexpr int NB = MT + 2 * WN;
    constexpr int STAGES = (5 * NB * 1024 <= 160 * 1024) ? 5 : 4;
    constexpr int LDS = STAGES * NB * 1024;
    static bool attr_set = false;
    auto kern = gemm_wl_kernel<MT, WN, EPI, STAGES>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(ceil_div(a.n_tiles, 2 * WN), a.S);
    hipLaunchKernelGGL(kern, grid, dim3(128 * WN), LDS, st, a);
    return hipGetLastError();
}

template <int WN, int EPI>
static hipError_t dispatch_wl(const GemmWsArgs& a, hipStream_t st) {
    const int mt = ceil_div(a.M, 16);
    if (mt <= 10) return launch_wl<10, WN, EPI>(a, st);
    if (mt <= 12) return launch_wl<12, WN, EPI>(a, st);
    if (mt <= 14) return launch_wl<14, WN, EPI>(a, st);
    if (mt <= 16) return launch_wl<16, WN, EPI>(a, st);
    if (mt <= 18) return launch_wl<18, WN, EPI>(a, st);
    if (mt <= 20) return launch_wl<20, WN, EPI>(a, st);
    return hipErrorInvalidValue;
}

// Shapes this kernel serves: 128 < M <= 320, whole 32-deep k-steps of X (Kx % 32 == 0), 16-B aligned rows.
extern "C" int aha_gemm_wl_supports(const GemmWsArgs* a, int epi) {
    return (epi == EPI_PARTIAL || epi == EPI_SWIGLU) && a->M > 128 && a->M <= 320 && a->Kx % 32 == 0 && a->Kx >= 32 && a->ldx % 8 == 0;
}

extern "C" hipError_t aha_gemm_wl(const GemmWsArgs* a, int epi, hipStream_t st) {
    if (!aha_gemm_wl_supports(a, epi)) return hipErrorInvalidValue;
    if (epi == EPI_SWIGLU && g_wl_bal && ceil_div(a->M, 16) == 18 && a->S == 1) {
        constexpr int LDS = 5 * 28 * 1024;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_wl_bal18_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_wl_bal18_kernel, dim3(ceil_div(a->n_tiles, 10)), dim3(640), LDS, st, *a);
        return hipGetLastError();
    }
    if (epi == EPI_SWIGLU) return dispatch_wl<5, EPI_SWIGLU>(*a, st);
    return dispatch_wl<4, EPI_PARTIAL>(*a, st);
}
