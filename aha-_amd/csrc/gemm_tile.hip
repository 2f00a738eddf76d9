// Tiled MFMA GEMM for the vision tower / projector (M = frames*patches is large, MFMA-bound):
//     C[M,N] = epilogue( A[M,K] * W[N,K]^T + bias )
// 128x128 output tile, BK = 64, 256 threads = 4 waves as 2(M) x 2(N), each wave 64x64
// (4x4 v_mfma_f32_16x16x32_bf16 accumulators).  Operands are staged global -> registers -> LDS
// (issue-early / write-late: the loads of tile k+1 are in flight during the MFMAs of tile k),
// LDS rows are 128 B with the 16-B chunk index XOR-swizzled by (row & 7) so both the staging
// ds_write_b128 and the fragment ds_read_b128 are bank-conflict free.  MFMA A operand = W rows,
// B operand = activation rows, so each lane's accumulator holds 4 consecutive n of one m and the
// epilogue stores 8-byte bf16x4 vectors.  Block ids are remapped so that the blocks one XCD
// receives (ids equal mod 8) walk consecutive n-tiles of the same m-panel (A panel stays in
// that XCD's L2).
#include "aha_kernels.h"
#include "tile_act.h"

#define TBK 64

// WT = 16x16 MFMA tiles per wave per dimension: WT = 4 -> 128x128 block tile (throughput shapes),
// WT = 2 -> 64x64 block tile (M <= ~1k rows: single-frame latency; 4x the workgroups).
template <int WT>
__global__ __launch_bounds__(256, 2) void gemm_tile_kernel(GemmTileArgs g) {
    constexpr int TBM = 32 * WT, TBN = 32 * WT;
    __shared__ __attribute__((aligned(16))) bf16 As[2][TBM * TBK];
    __shared__ __attribute__((aligned(16))) bf16 Ws[2][TBN * TBK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware remap (bijective only when nblk % 8 == 0; otherwise identity)
    const int tiles_n = ceil_div(g.N, TBN), tiles_m = ceil_div(g.M, TBM);
    const int nblk = tiles_n * tiles_m;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    // Grouped rasterisation: consecutive ids walk 8 m-panels x all n-tiles column by column, so the ~32
    // workgroups one XCD runs together share 8 A panels and ~4 W panels (fits its 4 MB L2).  With plain row-major
    // order they shared 1 A panel and 32 W panels (8 MB of W): W thrashed L2 and streamed from Infinity Cache
    // (measured: waves parked ~45 % on vmcnt/barrier, zero LDS conflicts).
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n, grp = bid / per_group, first_m = grp * GROUP_M;
    const int gmn = min(tiles_m - first_m, GROUP_M), inner = bid % per_group;
    const int bm = first_m + inner % gmn, bn = inner / gmn;
    const int m0 = bm * TBM, n0 = bn * TBN;

    f32x4 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: each thread moves 4 chunks of A and 4 of W per k-tile: row = tid/8 + 32*i, chunk = tid%8.
    // The loads of k-tile kt+1 are issued before the MFMAs of tile kt and written to the other LDS
    // buffer after them.  All loads are unconditional (indices clamped, out-of-range k chunks zeroed
    // by a select at the store) so the compiler keeps counted waits (see gemm_ws.hip for the
    // vmcnt(0) trap).  A second register set (two tiles in flight) was tried: it spills at
    // 2 waves/SIMD; the 8-phase LDS-DMA template is the planned replacement.
    const int srow = tid >> 3, sch = tid & 7;
    const int nk = ceil_div(g.K, TBK);
    int aoff[WT], woff[WT];                                              // element offsets fit 32 bits (checked on the host)
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        aoff[i] = min(m0 + srow + 32 * i, g.M - 1) * g.lda;
        woff[i] = min(n0 + srow + 32 * i, g.N - 1) * g.ldw;
    }
    bf16x8 raA[WT], rwA[WT];
    auto gload = [&](bf16x8 (&ra)[WT], bf16x8 (&rw)[WT], int kt) {
        // past the end of K: re-read the last tile / last chunk (zeroed at store time, or never stored to a live buffer)
        const int kc = min(min(kt, nk - 1) * TBK + sch * 8, g.K - 8);
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            ra[i] = *reinterpret_cast<const bf16x8*>(g.A + aoff[i] + kc);
            rw[i] = *reinterpret_cast<const bf16x8*>(g.W + woff[i] + kc);
        }
    };
    auto lstore = [&](bf16x8 (&ra)[WT], bf16x8 (&rw)[WT], int kt, int buf) {
        const bool kok = kt * TBK + sch * 8 < g.K;                     // the select sits at the STORE so it does not wait on the load early
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int row = srow + 32 * i;
            const int off = row * TBK + ((sch ^ (row & 7)) << 3);
            *reinterpret_cast<bf16x8*>(&As[buf][off]) = kok ? ra[i] : z;
            *reinterpret_cast<bf16x8*>(&Ws[buf][off]) = kok ? rw[i] : z;
        }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < TBK / 32; ++ks) {
            bf16x8 af[WT], wf[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                const int row = wm * (16 * WT) + i * 16 + r16;
                af[i] = *reinterpret_cast<const bf16x8*>(&As[buf][row * TBK + (((ks * 4 + q) ^ (row & 7)) << 3)]);
                const int wrow = wn * (16 * WT) + i * 16 + r16;
                wf[i] = *reinterpret_cast<const bf16x8*>(&Ws[buf][wrow * TBK + (((ks * 4 + q) ^ (wrow & 7)) << 3)]);
            }
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
        }
    };

    gload(raA, rwA, 0);
    lstore(raA, rwA, 0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {                                 // issue early / write late, one tile ahead
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(raA, rwA, kt + 1);                     // uniform branch around the whole batch of loads
        compute(buf);
        if (kt + 1 < nk) lstore(raA, rwA, kt + 1, buf ^ 1);
        __syncthreads();
    }

    // epilogue: acc[i][j][e] <-> m = m0 + wm*64 + i*16 + r16, n = n0 + wn*64 + j*16 + q*4 + e.
    // All side inputs (bias, residual, row-add) are fetched as 8-byte vectors and issued together
    // BEFORE they are used: one scalar bf16 load per element made hipcc wait vmcnt(0) ~130 times per
    // thread, which dominated the K=1024 GEMMs.  N % 4 == 0 and ld % 4 == 0 are required (checked on
    // the host), so a 4-wide group is either fully inside or fully outside N.
    const bf16x4 z4 = {0, 0, 0, 0};
    bf16x4 bv[WT];
#pragma unroll
    for (int j = 0; j < WT; ++j) {
        const int n = min(n0 + wn * (16 * WT) + j * 16 + q * 4, g.N - 4);
        bv[j] = g.bias ? *reinterpret_cast<const bf16x4*>(g.bias + n) : z4;
    }
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        const int m = m0 + wm * (16 * WT) + i * 16 + r16;
        const int mc = min(m, g.M - 1);
        bf16x4 rv[WT], pv[WT];
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int n = min(n0 + wn * (16 * WT) + j * 16 + q * 4, g.N - 4);
            rv[j] = g.residual ? *reinterpret_cast<const bf16x4*>(g.residual + (long)mc * g.ldr + n) : z4;
            pv[j] = g.rowadd ? *reinterpret_cast<const bf16x4*>(g.rowadd + (long)(mc % g.rowadd_period) * g.ldra + n) : z4;
        }
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int n = n0 + wn * (16 * WT) + j * 16 + q * 4;
            if (n >= g.N) continue;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = rbf(acc[i][j][e] + bf2f(bv[j][e]));            // Linear output (bf16)
                if (g.act == ACT_GELU_TANH) x = rbf(gelu_tanh_f(x));
                else if (g.act == ACT_GELU_ERF) x = rbf(gelu_erf_f(x));
                else if (g.act == ACT_QUICK_GELU) x = rbf(quick_gelu_bf16(x));
                if (g.residual) x = rbf(bf2f(rv[j][e]) + x);
                if (g.rowadd) x = rbf(x + bf2f(pv[j][e]));
                o[e] = f2bf(x);
            }
            *reinterpret_cast<bf16x4*>(g.C + (long)m * g.ldc + n) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Epilogue of the LDS-DMA kernels through LDS (round 2).  The accumulator layout gives a lane 4 consecutive output columns
// of one row per 16x16 tile, so a direct store is 16 (WI x WJ) wave-instructions of 8 B per lane, each touching 16 rows with
// 32 B: quarter cache lines, store-issue-bound.  The K sweep of tools/diag/tile_ablate.py put the FIXED cost of a 256x128
// workgroup (fill + this epilogue) at 8-12 us against 5.6-7.6 us per 64 of K - a third of the K = 1024 tower GEMMs.
// Here a wave first writes its finished bf16 tile (bias and activation applied: same rounding points) row-major into the
// stage memory, which is free once the k-loop has drained, and then moves it out in full 128-byte rows, 16 B per lane
// (8 rows per wave-instruction), adding the residual / row-add operand on the way with equally wide loads.  One rounding per
// step exactly as before, so every variant stays bit-identical to gemm_tile_kernel.
// Requirements (checked by the caller): N % 8 == 0, ldc % 8 == 0, ldr / ldra % 8 == 0 - otherwise the direct epilogue runs.
// ---------------------------------------------------------------------------------------------
template <int WI, int WJ>
static __device__ __forceinline__ void tile_epilogue_lds(const GemmTileArgs& g, f32x4 (&acc)[WI][WJ], bf16* stg, const int mw, const int nw,
                                                         const int lane) {
    constexpr int COLS = 16 * WJ, ST = COLS + 8, CPR = COLS / 8;      // staging row stride padded by 16 B; 16-B chunks per row
    const int q = lane >> 4, r16 = lane & 15;
    const bf16x4 z4 = {0, 0, 0, 0};
    bf16x4 bv[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int n = min(nw + j * 16 + q * 4, g.N - 4);
        bv[j] = g.bias ? *reinterpret_cast<const bf16x4*>(g.bias + n) : z4;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = rbf(acc[i][j][e] + bf2f(bv[j][e]));            // Linear output (bf16)
                if (g.act == ACT_GELU_TANH) x = rbf(gelu_tanh_f(x));
                else if (g.act == ACT_GELU_ERF) x = rbf(gelu_erf_f(x));
                else if (g.act == ACT_QUICK_GELU) x = rbf(quick_gelu_bf16(x));
                o[e] = f2bf(x);
            }
            *reinterpret_cast<bf16x4*>(&stg[(i * 16 + r16) * ST + j * 16 + q * 4]) = o;
        }
    // the staging area is private to the wave: its own LDS writes only need to have landed (the compiler's lgkmcnt wait)
    constexpr int ITERS = (16 * WI * CPR) / 64;
    static_assert((16 * WI * CPR) % 64 == 0, "wave tile must split into whole 1-KiB store instructions");
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = it * 64 + lane, row = idx / CPR, ch = idx % CPR;
        const int m = mw + row, n = nw + ch * 8;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(&stg[row * ST + ch * 8]);
        const int mc = min(m, g.M - 1), nc = min(n, g.N - 8);
        bf16x8 rv, pv;
        if (g.residual) rv = *reinterpret_cast<const bf16x8*>(g.residual + (long)mc * g.ldr + nc);
        if (g.rowadd) pv = *reinterpret_cast<const bf16x8*>(g.rowadd + (long)(mc % g.rowadd_period) * g.ldra + nc);
        if (g.residual)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f2bf(rbf(bf2f(rv[e]) + bf2f(v[e])));
        if (g.rowadd)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f2bf(rbf(bf2f(v[e]) + bf2f(pv[e])));
        if (m < g.M && n < g.N) *reinterpret_cast<bf16x8*>(g.C + (long)m * g.ldc + n) = v;
    }
}
static __device__ __forceinline__ bool tile_epilogue_wide_ok(const GemmTileArgs& g) {
    return g.wide_epi && !(g.N & 7) && !(g.ldc & 7) && (!g.residual || !(g.ldr & 7)) && (!g.rowadd || !(g.ldra & 7));
}

// The same epilogue with ONE staging image for the whole workgroup (BM x BN bf16, row stride BN + 8): for wave tiles narrower
// than 128 bytes (the 288 x 128 tile's waves own 144 x 32) the per-wave image would store half lines.  All waves write
// their part, one barrier, then every wave-instruction moves 1 KiB = whole BN-wide rows.  Same rounding points.
template <int BM, int BN, int WI, int WJ, int NT>
static __device__ __forceinline__ void tile_epilogue_lds_wg(const GemmTileArgs& g, f32x4 (&acc)[WI][WJ], bf16* stg, const int m0, const int n0,
                                                            const int mw, const int nw, const int tid) {
    constexpr int ST = BN + 8, CPR = BN / 8;                         // staging row stride in elements; 16-B chunks per row
    const int lane = tid & 63, q = lane >> 4, r16 = lane & 15;
    const bf16x4 z4 = {0, 0, 0, 0};
    bf16x4 bv[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int n = min(n0 + nw + j * 16 + q * 4, g.N - 4);
        bv[j] = g.bias ? *reinterpret_cast<const bf16x4*>(g.bias + n) : z4;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = rbf(acc[i][j][e] + bf2f(bv[j][e]));            // Linear output (bf16)
                if (g.act == ACT_GELU_TANH) x = rbf(gelu_tanh_f(x));
                else if (g.act == ACT_GELU_ERF) x = rbf(gelu_erf_f(x));
                else if (g.act == ACT_QUICK_GELU) x = rbf(quick_gelu_bf16(x));
                o[e] = f2bf(x);
            }
            *reinterpret_cast<bf16x4*>(&stg[(mw + i * 16 + r16) * ST + nw + j * 16 + q * 4]) = o;
        }
    __syncthreads();
    static_assert((BM * CPR) % NT == 0, "tile must split into whole rounds of 16-byte stores");
#pragma unroll
    for (int it = 0; it < (BM * CPR) / NT; ++it) {
        const int idx = it * NT + tid, row = idx / CPR, ch = idx % CPR;
        const int m = m0 + row, n = n0 + ch * 8;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(&stg[row * ST + ch * 8]);
        const int mc = min(m, g.M - 1), nc = min(n, g.N - 8);
        bf16x8 rv, pv;
        if (g.residual) rv = *reinterpret_cast<const bf16x8*>(g.residual + (long)mc * g.ldr + nc);
        if (g.rowadd) pv = *reinterpret_cast<const bf16x8*>(g.rowadd + (long)(mc % g.rowadd_period) * g.ldra + nc);
        if (g.residual)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f2bf(rbf(bf2f(rv[e]) + bf2f(v[e])));
        if (g.rowadd)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f2bf(rbf(bf2f(v[e]) + bf2f(pv[e])));
        if (m < g.M && n < g.N) *reinterpret_cast<bf16x8*>(g.C + (long)m * g.ldc + n) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA family (K % 64 == 0): block tile BM x BN = (WAVES_M*16*WI) x (WAVES_N*16*WJ), one wave per
// (16*WI) x (16*WJ) sub-tile, STAGES stages of (BM+BN) 128-byte rows filled by global_load_lds_dwordx4
// (no staging VGPRs, so STAGES-1 k-tiles are in flight while one computes - the register-staged kernel
// above keeps one, and a second register set spills).  Per k-tile: counted `s_waitcnt vmcnt((STAGES-2)*P)`
// (this wave's P DMA pieces of tile kt have landed, the younger tiles stay in flight) -> raw s_barrier
// (everyone's pieces landed; everyone finished reading the stage about to be refilled) -> issue tile
// kt+STAGES-1 -> ds_read + MFMA on tile kt.  LDS image = the same 128-B rows with the chunk index
// XOR-swizzled by (row & 7); an LDS-DMA writes lane-linear, so the swizzle is applied to the SOURCE
// address (guide rule 21).  One __shared__ object only; no ordinary global loads inside the loop.
//   <4,4,4,2,3>  256x128, 8 waves, 3 x 48 KB : throughput shapes (grids that fill their last round of CUs); ILV = the
//                DMA pieces of tile kt+2 issued between the MFMAs of tile kt instead of in a bunch after the barrier
//   <2,4,4,2,4>  128x128, 8 waves, 4 x 32 KB : (sweeps only: wins two mid-M shapes by < 10 %)
//   <2,2,2,2,4>   64x64,  4 waves, 4 x 16 KB : (sweeps only)
//   <2,2,2,2,3>   64x64,  4 waves, 3 x 16 KB : everything smaller, down to one frame (3 workgroups per CU)
// Every variant accumulates one output element's k-tiles in the same order, so they are bit-identical to each
// other and to gemm_tile_kernel (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------------
template <int WI, int WJ, int WAVES_M, int WAVES_N, int STAGES, bool ILV = false, int MINB = 1>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MINB) void gemm_tile_dma_kernel(GemmTileArgs g) {
    constexpr int BM = WAVES_M * 16 * WI, BN = WAVES_N * 16 * WJ, NT = 64 * WAVES_M * WAVES_N;
    constexpr int ROWS = BM + BN, STAGE = ROWS * TBK;              // elements per stage
    constexpr int P = ROWS * 8 / NT;                                // 16-B DMA pieces per thread per stage
    static_assert(ROWS * 8 % NT == 0, "stage image must split evenly over the threads");
    static_assert((STAGES - 2) * P <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char dsm_raw[];
    bf16* lds = reinterpret_cast<bf16*>(dsm_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tiles_n = ceil_div(g.N, BN), tiles_m = ceil_div(g.M, BM);
    const int nblk = tiles_n * tiles_m;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    // Grouped rasterisation: consecutive ids walk GROUP_M m-panels x all n-tiles column by column, so the ~32
    // workgroups one XCD runs together share a few A panels and a few W panels (fits its 4 MB L2).  With plain
    // row-major order they shared 1 A panel and 32 W panels (8 MB of W): W thrashed L2 and streamed from Infinity
    // Cache (measured: waves parked ~45 % on vmcnt/barrier, zero LDS conflicts).
    constexpr int GROUP_M = BM >= 256 ? 4 : 8;
    const int per_group = GROUP_M * tiles_n, grp = bid / per_group, first_m = grp * GROUP_M;
    const int gmn = min(tiles_m - first_m, GROUP_M), inner = bid % per_group;
    const int bm = first_m + inner % gmn, bn = inner / gmn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int nk = g.K / TBK;

    f32x4 acc[WI][WJ];
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // DMA piece p of a stage = NT consecutive 16-B chunks of the stage image (chunk c -> row c/8, slot c%8;
    // rows [0,BM) come from A, rows [BM,ROWS) from W); a lane's source chunk = slot ^ (row & 7).
    const bf16* src[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int c = i * NT + tid, row = c >> 3, ch = (c & 7) ^ (row & 7);
        src[i] = row < BM ? g.A + min(m0 + row, g.M - 1) * g.lda + ch * 8
                          : g.W + min(n0 + row - BM, g.N - 1) * g.ldw + ch * 8;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma = [&](int kt, int stage) {
        const int k0 = min(kt, nk - 1) * TBK;                      // past the end: refill a dead stage (keeps vmcnt counts fixed)
        bf16* sa = lds + stage * STAGE;
#pragma unroll
        for (int i = 0; i < P; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + k0), (lptr_t)(sa + (i * NT + wave * 64) * 8), 16, 0, 0);
    };
    auto compute = [&](int stage) {
        const bf16* sa = lds + stage * STAGE;
        const bf16* sb = sa + BM * TBK;
#pragma unroll
        for (int ks = 0; ks < TBK / 32; ++ks) {
            bf16x8 af[WI], wf[WJ];
#pragma unroll
            for (int i = 0; i < WI; ++i) {
                const int row = wm * (16 * WI) + i * 16 + r16;
                af[i] = *reinterpret_cast<const bf16x8*>(&sa[row * TBK + (((ks * 4 + q) ^ (row & 7)) << 3)]);
            }
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const int wrow = wn * (16 * WJ) + j * 16 + r16;
                wf[j] = *reinterpret_cast<const bf16x8*>(&sb[wrow * TBK + (((ks * 4 + q) ^ (wrow & 7)) << 3)]);
            }
#pragma unroll
            for (int i = 0; i < WI; ++i)
#pragma unroll
                for (int j = 0; j < WJ; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
        }
    };

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) dma(s, s);
    int st_cur = 0, st_new = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * P) : "memory");   // tile kt (this wave's pieces) landed
        __builtin_amdgcn_s_barrier();
        if constexpr (!ILV) {
            dma(kt + STAGES - 1, st_new);
            compute(st_cur);
        } else {
            // Interleaved issue: every fragment read of the k-tile first, then the DMA pieces of tile kt+STAGES-1 spread
            // between the MFMAs (a piece issued while the MFMA pipe is busy costs ~60 cycles; bunched after the barrier,
            // where both waves of a SIMD stand at the same point, 100-185 each with nothing to overlap).  The pieces
            // sit between the MFMA groups in SOURCE order (hipcc does not move an LDS-DMA across a ds_read) and the
            // group barriers keep the MFMAs from clumping.
            constexpr int KS = TBK / 32, NM = WI * WJ * KS, PER = NM / (P + 1);
            const bf16* sa = lds + st_cur * STAGE;
            const bf16* sb = sa + BM * TBK;
            bf16* sn = lds + st_new * STAGE;
            const int k0 = min(kt + STAGES - 1, nk - 1) * TBK;
            bf16x8 af[KS][WI], wf[KS][WJ];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < WI; ++i) {
                    const int row = wm * (16 * WI) + i * 16 + r16;
                    af[ks][i] = *reinterpret_cast<const bf16x8*>(&sa[row * TBK + (((ks * 4 + q) ^ (row & 7)) << 3)]);
                }
#pragma unroll
                for (int j = 0; j < WJ; ++j) {
                    const int wrow = wn * (16 * WJ) + j * 16 + r16;
                    wf[ks][j] = *reinterpret_cast<const bf16x8*>(&sb[wrow * TBK + (((ks * 4 + q) ^ (wrow & 7)) << 3)]);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < WI; ++i)
#pragma unroll
                    for (int j = 0; j < WJ; ++j) {
                        acc[i][j] = mfma16(wf[ks][j], af[ks][i], acc[i][j]);
                        constexpr int dummy = 0; (void)dummy;
                        const int n = (ks * WI + i) * WJ + j + 1;               // MFMAs issued so far (compile-time after unrolling)
                        if (n % PER == 0 && n / PER <= P) {
                            const int pc = n / PER - 1;
                            __builtin_amdgcn_global_load_lds((gptr_t)(src[pc] + k0), (lptr_t)(sn + (pc * NT + wave * 64) * 8), 16, 0, 0);
                        }
                    }
            __builtin_amdgcn_sched_group_barrier(0x100, (WI + WJ) * KS, 0);             // DS reads
#pragma unroll
            for (int i = 0; i < P; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);                    // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                      // VMEM read (LDS-DMA piece)
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - P * PER, 0);
        }
        st_cur = st_cur == STAGES - 1 ? 0 : st_cur + 1;
        st_new = st_new == STAGES - 1 ? 0 : st_new + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // nothing may still target LDS when the block retires

    if (tile_epilogue_wide_ok(g)) {                                // uniform
        __builtin_amdgcn_s_barrier();                              // every wave is done reading the stages: they become staging
        tile_epilogue_lds<WI, WJ>(g, acc, lds + wave * (16 * WI * (16 * WJ + 8)), m0 + wm * (16 * WI), n0 + wn * (16 * WJ), lane);
        return;
    }
    // direct epilogue (same rounding points as gemm_tile_kernel)
    const bf16x4 z4 = {0, 0, 0, 0};
    bf16x4 bv[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int n = min(n0 + wn * (16 * WJ) + j * 16 + q * 4, g.N - 4);
        bv[j] = g.bias ? *reinterpret_cast<const bf16x4*>(g.bias + n) : z4;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int m = m0 + wm * (16 * WI) + i * 16 + r16;
        const int mc = min(m, g.M - 1);
        bf16x4 rv[WJ], pv[WJ];
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int n = min(n0 + wn * (16 * WJ) + j * 16 + q * 4, g.N - 4);
            rv[j] = g.residual ? *reinterpret_cast<const bf16x4*>(g.residual + (long)mc * g.ldr + n) : z4;
            pv[j] = g.rowadd ? *reinterpret_cast<const bf16x4*>(g.rowadd + (long)(mc % g.rowadd_period) * g.ldra + n) : z4;
        }
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int n = n0 + wn * (16 * WJ) + j * 16 + q * 4;
            if (n >= g.N) continue;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = rbf(acc[i][j][e] + bf2f(bv[j][e]));
                if (g.act == ACT_GELU_TANH) x = rbf(gelu_tanh_f(x));
                else if (g.act == ACT_GELU_ERF) x = rbf(gelu_erf_f(x));
                else if (g.act == ACT_QUICK_GELU) x = rbf(quick_gelu_bf16(x));
                if (g.residual) x = rbf(bf2f(rv[j][e]) + x);
                if (g.rowadd) x = rbf(x + bf2f(pv[j][e]));
                o[e] = f2bf(x);
            }
            *reinterpret_cast<bf16x4*>(g.C + (long)m * g.ldc + n) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 32-deep k-stages: the same LDS-DMA structure with BK = 32, so a 256x128 stage is 24 KB and three of them (72 KB)
// let TWO workgroups share a CU (<= 128 VGPRs each): while one workgroup's waves are parked on their vmcnt/barrier
// (39-47 % of wave cycles in the 64-deep kernel, profiles/r01_pmc_mfma_vit32.json) the other one computes.
// LDS image: 64-byte rows (4 chunks), chunk index XOR (row >> 2) & 3: the 16 rows a ds_read_b128 group touches span
// four 256-byte bank rows and get four different 16-byte slots in each - conflict-free.  Same k order: bit-identical.
// ---------------------------------------------------------------------------------------------
template <int WI, int WJ, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 4) void gemm_tile_dma32_kernel(GemmTileArgs g) {
    constexpr int BK = 32;
    constexpr int BM = WAVES_M * 16 * WI, BN = WAVES_N * 16 * WJ, NT = 64 * WAVES_M * WAVES_N;
    constexpr int ROWS = BM + BN, STAGE = ROWS * BK;
    constexpr int NPIECE = ROWS * 4 / 64, NWV = NT / 64;           // 1-KiB wave-pieces per stage; waves
    constexpr int P = (NPIECE + NWV - 1) / NWV;                     // DMA wave-instructions per wave per stage (surplus ones re-load the last piece:
                                                                    // same bytes to the same place, so every wave's vmcnt counts stay equal)
    static_assert(ROWS % 16 == 0, "stage image must split into whole 1-KiB pieces");
    static_assert((STAGES - 2) * P <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char dsm_raw[];
    bf16* lds = reinterpret_cast<bf16*>(dsm_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tiles_n = ceil_div(g.N, BN), tiles_m = ceil_div(g.M, BM);
    const int nblk = tiles_n * tiles_m;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    constexpr int GROUP_M = BM >= 256 ? 4 : 8;
    const int per_group = GROUP_M * tiles_n, grp = bid / per_group, first_m = grp * GROUP_M;
    const int gmn = min(tiles_m - first_m, GROUP_M), inner = bid % per_group;
    const int bm = first_m + inner % gmn, bn = inner / gmn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int nk = g.K / BK;

    f32x4 acc[WI][WJ];
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // a piece (16 rows) lies entirely in A or entirely in W: wave-uniform base + a 32-bit per-lane byte offset (one VGPR per piece;
    // check_tile_args keeps M*lda and N*ldw below 2^31 elements)
    const char* pbase[P];
    unsigned poff[P];
    int pdst[P];                                                    // element offset of this wave's piece inside a stage (wave-uniform)
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int wp = min(i * NWV + __builtin_amdgcn_readfirstlane(wave), NPIECE - 1);
        const int c = wp * 64 + lane, row = c >> 2, ch = (c & 3) ^ ((row >> 2) & 3);
        const bool isA = wp * 16 < BM;                              // BM % 16 == 0
        pbase[i] = reinterpret_cast<const char*>(isA ? g.A : g.W);
        poff[i] = isA ? ((unsigned)min(m0 + row, g.M - 1) * (unsigned)g.lda + ch * 8) * 2u
                      : ((unsigned)min(n0 + row - BM, g.N - 1) * (unsigned)g.ldw + ch * 8) * 2u;
        pdst[i] = wp * 512;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
        const int k0 = min(s, nk - 1) * BK;
#pragma unroll
        for (int i = 0; i < P; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(pbase[i] + k0 * 2 + poff[i]), (lptr_t)(lds + s * STAGE + pdst[i]), 16, 0, 0);
    }
    int st_cur = 0, st_new = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * P) : "memory");
        __builtin_amdgcn_s_barrier();
        constexpr int NM = WI * WJ, PER = NM / (P + 1);
        const bf16* sa = lds + st_cur * STAGE;
        const bf16* sb = sa + BM * BK;
        bf16* sn = lds + st_new * STAGE;
        const int k0 = min(kt + STAGES - 1, nk - 1) * BK;
        bf16x8 af[WI], wf[WJ];
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int row = wm * (16 * WI) + i * 16 + r16;
            af[i] = *reinterpret_cast<const bf16x8*>(&sa[row * BK + ((q ^ ((row >> 2) & 3)) << 3)]);
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int wrow = wn * (16 * WJ) + j * 16 + r16;
            wf[j] = *reinterpret_cast<const bf16x8*>(&sb[wrow * BK + ((q ^ ((wrow >> 2) & 3)) << 3)]);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i)
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
                const int n = i * WJ + j + 1;
                if (n % PER == 0 && n / PER <= P) {
                    const int pc = n / PER - 1;
                    __builtin_amdgcn_global_load_lds((gptr_t)(pbase[pc] + k0 * 2 + poff[pc]), (lptr_t)(sn + pdst[pc]), 16, 0, 0);
                }
            }
        __builtin_amdgcn_sched_group_barrier(0x100, WI + WJ, 0);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - P * PER, 0);
        st_cur = st_cur == STAGES - 1 ? 0 : st_cur + 1;
        st_new = st_new == STAGES - 1 ? 0 : st_new + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (tile_epilogue_wide_ok(g)) {                                // uniform
        __builtin_amdgcn_s_barrier();
        if constexpr (WJ * 16 >= 64)
            tile_epilogue_lds<WI, WJ>(g, acc, lds + wave * (16 * WI * (16 * WJ + 8)), m0 + wm * (16 * WI), n0 + wn * (16 * WJ), lane);
        else
            tile_epilogue_lds_wg<BM, BN, WI, WJ, NT>(g, acc, lds, m0, n0, wm * (16 * WI), wn * (16 * WJ), tid);
        return;
    }
    const bf16x4 z4 = {0, 0, 0, 0};
    bf16x4 bv[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int n = min(n0 + wn * (16 * WJ) + j * 16 + q * 4, g.N - 4);
        bv[j] = g.bias ? *reinterpret_cast<const bf16x4*>(g.bias + n) : z4;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int m = m0 + wm * (16 * WI) + i * 16 + r16;
        const int mc = min(m, g.M - 1);
        bf16x4 rv[WJ], pv[WJ];
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int n = min(n0 + wn * (16 * WJ) + j * 16 + q * 4, g.N - 4);
            rv[j] = g.residual ? *reinterpret_cast<const bf16x4*>(g.residual + (long)mc * g.ldr + n) : z4;
            pv[j] = g.rowadd ? *reinterpret_cast<const bf16x4*>(g.rowadd + (long)(mc % g.rowadd_period) * g.ldra + n) : z4;
        }
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int n = n0 + wn * (16 * WJ) + j * 16 + q * 4;
            if (n >= g.N) continue;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = rbf(acc[i][j][e] + bf2f(bv[j][e]));
                if (g.act == ACT_GELU_TANH) x = rbf(gelu_tanh_f(x));
                else if (g.act == ACT_GELU_ERF) x = rbf(gelu_erf_f(x));
                else if (g.act == ACT_QUICK_GELU) x = rbf(quick_gelu_bf16(x));
                if (g.residual) x = rbf(bf2f(rv[j][e]) + x);
                if (g.rowadd) x = rbf(x + bf2f(pv[j][e]));
                o[e] = f2bf(x);
            }
            *reinterpret_cast<bf16x4*>(g.C + (long)m * g.ldc + n) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 64x64 tile with SOFTWARE-PIPELINED k-steps (round 3) for the latency path: one to four frames (M = 576 .. 2304), where a
// GEMM is 72-576 small tiles and every workgroup's k-loop is a serial chain.  gemm_tile_dma_kernel<2,2,2,2,3> spends
// ~600 cycles per 64-deep k-tile on 8 MFMAs per wave (wait -> barrier -> fragment reads -> MFMAs, nothing overlapped): fc2's
// K = 4096 alone is 22 us.  Here the k-step is split around ONE barrier like gemm_tile_p288s_kernel:
//   TOP(s): MFMAs of k-half 0 (fragments read during MID(s-1)) | k-half 1 fragments of stage s by inline-asm ds_read_b128
//           (issued behind the first MFMA, so that no compiler-inserted lgkmcnt wait can catch them) | DMA pieces 0,1 of s+3
//   MID(s): lgkmcnt(0) -> vmcnt(6) -> s_barrier -> MFMAs of k-half 1 | k-half 0 fragments of stage s+1 | DMA pieces 2,3 of s+3
// 4 stages of 16 KB (2 workgroups per CU), 4 waves as 2x2 with 32x32 wave tiles, four 1-KiB pieces per wave per stage.
// Same k order per output element as every other variant: bit-identical.  K % 64 == 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_tile_ps64_kernel(GemmTileArgs g) {
    constexpr int BM = 64, BN = 64, STAGES = 4, ROWS = BM + BN, STAGE = ROWS * TBK, P = 4;
    extern __shared__ __attribute__((aligned(16))) char dsm_raw[];
    bf16* lds = reinterpret_cast<bf16*>(dsm_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = ceil_div(g.N, BN), tiles_m = ceil_div(g.M, BM);
    const int nblk = tiles_n * tiles_m;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n, grp = bid / per_group, first_m = grp * GROUP_M;
    const int gmn = min(tiles_m - first_m, GROUP_M), inner = bid % per_group;
    const int bm = first_m + inner % gmn, bn = inner / gmn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int nk = g.K / TBK;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // DMA piece p = 8 rows of the stage image (rows [0,64) from A, [64,128) from W); wave w issues pieces w, w+4, w+8, w+12
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const bf16* src[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int row = (wave + 4 * i) * 8 + (lane >> 3), ch = (lane & 7) ^ (row & 7);
        src[i] = row < BM ? g.A + min(m0 + row, g.M - 1) * g.lda + ch * 8
                          : g.W + min(n0 + row - BM, g.N - 1) * g.ldw + ch * 8;
    }
    auto dma_piece = [&](int i, int kt, int stage) {
        const int k0 = min(kt, nk - 1) * TBK;                      // past the end: refill a dead stage (keeps the vmcnt counts fixed)
        __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + k0), (lptr_t)(lds + stage * STAGE + (wave + 4 * i) * 512), 16, 0, 0);
    };
    // fragment addresses: row = (multiple of 16) + r16, so the swizzle key (row & 7) is (r16 & 7) for every fragment
    const unsigned off0 = (unsigned)r16 * (TBK * 2) + ((q ^ (r16 & 7)) << 4), off1 = off0 ^ 64u;     // k-half 0 / 1: chunk bit 2
    const char* lds_b = reinterpret_cast<const char*>(lds);
    const unsigned lds_u32 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)dsm_raw;
    bf16x8 a0[2], w0[2], a1[2], w1[2];
    auto read_half0 = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a0[i] = *reinterpret_cast<const bf16x8*>(lds_b + stage * (STAGE * 2) + (wm * 32 + i * 16) * (TBK * 2) + off0);
            w0[i] = *reinterpret_cast<const bf16x8*>(lds_b + stage * (STAGE * 2) + (BM + wn * 32 + i * 16) * (TBK * 2) + off0);
        }
    };

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
#pragma unroll
        for (int i = 0; i < P; ++i) dma_piece(i, s, s);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * P) : "memory");
    __builtin_amdgcn_s_barrier();
    read_half0(0);
    int st_cur = 0, st_new = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int st_next = st_cur == STAGES - 1 ? 0 : st_cur + 1;
        // ---- TOP
        acc[0][0] = mfma16(w0[0], a0[0], acc[0][0]);
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned va = lds_u32 + st_cur * (STAGE * 2) + wm * (32 * TBK * 2) + off1;
            const unsigned vw = lds_u32 + st_cur * (STAGE * 2) + (BM + wn * 32) * (TBK * 2) + off1;
            asm volatile("ds_read_b128 %0, %1" : "=v"(a1[0]) : "v"(va) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a1[1]) : "v"(va) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(w1[0]) : "v"(vw) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(w1[1]) : "v"(vw) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][1] = mfma16(w0[1], a0[0], acc[0][1]);
        dma_piece(0, kt + STAGES - 1, st_new);
        acc[1][0] = mfma16(w0[0], a0[1], acc[1][0]);
        dma_piece(1, kt + STAGES - 1, st_new);
        acc[1][1] = mfma16(w0[1], a0[1], acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- MID
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + 2) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = mfma16(w1[0], a1[0], acc[0][0]);
        read_half0(st_next);
        acc[0][1] = mfma16(w1[1], a1[0], acc[0][1]);
        dma_piece(2, kt + STAGES - 1, st_new);
        acc[1][0] = mfma16(w1[0], a1[1], acc[1][0]);
        dma_piece(3, kt + STAGES - 1, st_new);
        acc[1][1] = mfma16(w1[1], a1[1], acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        st_cur = st_next;
        st_new = st_new == STAGES - 1 ? 0 : st_new + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // nothing may still target LDS when the block retires

    if (tile_epilogue_wide_ok(g)) {                                // uniform
        __builtin_amdgcn_s_barrier();                              // every wave is done reading the stages: they become staging
        tile_epilogue_lds<2, 2>(g, acc, lds + wave * (32 * (32 + 8)), m0 + wm * 32, n0 + wn * 32, lane);
        return;
    }
    const bf16x4 z4 = {0, 0, 0, 0};
    bf16x4 bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = min(n0 + wn * 32 + j * 16 + q * 4, g.N - 4);
        bv[j] = g.bias ? *reinterpret_cast<const bf16x4*>(g.bias + n) : z4;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + r16;
        const int mc = min(m, g.M - 1);
        bf16x4 rv[2], pv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = min(n0 + wn * 32 + j * 16 + q * 4, g.N - 4);
            rv[j] = g.residual ? *reinterpret_cast<const bf16x4*>(g.residual + (long)mc * g.ldr + n) : z4;
            pv[j] = g.rowadd ? *reinterpret_cast<const bf16x4*>(g.rowadd + (long)(mc % g.rowadd_period) * g.ldra + n) : z4;
        }
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + q * 4;
            if (n >= g.N) continue;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = tile_act(acc[i][j][e] + bf2f(bv[j][e]), g.act);
                if (g.residual) x = rbf(bf2f(rv[j][e]) + x);
                if (g.rowadd) x = rbf(x + bf2f(pv[j][e]));
                o[e] = f2bf(x);
            }
            *reinterpret_cast<bf16x4*>(g.C + (long)m * g.ldc + n) = o;
        }
    }
}

static hipError_t launch_ps64(const GemmTileArgs* g, hipStream_t st) {
    constexpr int lds_bytes = 4 * 128 * TBK * 2;                   // 64 KB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tile_ps64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nblk = ceil_div(g->N, 64) * ceil_div(g->M, 64);
    hipLaunchKernelGGL(gemm_tile_ps64_kernel, dim3(nblk), dim3(256), lds_bytes, st, *g);
    return hipGetLastError();
}

template <int WI, int WJ, int WAVES_M, int WAVES_N, int STAGES>
static hipError_t launch_dma32(const GemmTileArgs* g, hipStream_t st) {
    constexpr int BM = WAVES_M * 16 * WI, BN = WAVES_N * 16 * WJ;
    constexpr int lds_bytes = STAGES * (BM + BN) * 32 * 2;
    static_assert(lds_bytes >= (WJ * 16 >= 64 ? WAVES_M * WAVES_N * 16 * WI * (16 * WJ + 8) : BM * (BN + 8)) * 2, "stage ring must hold the epilogue staging image");
    static bool attr_set = false;
    auto kern = gemm_tile_dma32_kernel<WI, WJ, WAVES_M, WAVES_N, STAGES>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nblk = ceil_div(g->N, BN) * ceil_div(g->M, BM);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WAVES_M * WAVES_N), lds_bytes, st, *g);
    return hipGetLastError();
}

// tuning "tile_dma": 0 disables the LDS-DMA family, 1 auto, >= 2 forces variant id (tests, sweeps) whenever K allows
static int g_tile_dma = 1;
extern "C" void aha_gemm_tile_set_dma(int on) { g_tile_dma = on; }
static int g_tile_p288 = 1;      // tuning "tile_p288": persistent 288x256 kernel where its decomposition fits (0: round-2 variants only; n > 1: efficiency threshold n %)
extern "C" void aha_gemm_tile_set_p288(int on) { g_tile_p288 = on; }
static int g_tile_epi = 1;       // tuning "tile_epi": 1 = LDS-transposed wide epilogue (default), 0 = direct 8-byte stores
extern "C" void aha_gemm_tile_set_epi(int on) { g_tile_epi = on; }

template <int WI, int WJ, int WAVES_M, int WAVES_N, int STAGES, bool ILV = false, int MINB = 1>
static hipError_t launch_dma(const GemmTileArgs* g, hipStream_t st) {
    constexpr int BM = WAVES_M * 16 * WI, BN = WAVES_N * 16 * WJ;
    constexpr int lds_bytes = STAGES * (BM + BN) * TBK * 2;
    static_assert(lds_bytes <= 160 * 1024, "stage ring exceeds the CU's LDS");
    static_assert(lds_bytes >= WAVES_M * WAVES_N * 16 * WI * (16 * WJ + 8) * 2, "stage ring must hold the epilogue staging image");
    static bool attr_set = false;
    auto kern = gemm_tile_dma_kernel<WI, WJ, WAVES_M, WAVES_N, STAGES, ILV, MINB>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nblk = ceil_div(g->N, BN) * ceil_div(g->M, BM);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WAVES_M * WAVES_N), lds_bytes, st, *g);
    return hipGetLastError();
}

static hipError_t launch_dma_variant(int v, const GemmTileArgs* g, hipStream_t st) {
    switch (v) {
        case 2: return launch_dma<4, 4, 4, 2, 3, true>(g, st);   // 256x128, 8 waves, 144 KB (1 per CU), DMA/MFMA interleave
        case 5: return launch_dma<2, 2, 2, 2, 3>(g, st);     //  64x64,  4 waves,  48 KB (3 per CU)
        case 8: return launch_dma32<4, 4, 4, 2, 3>(g, st);       // 256x128, 32-deep stages, 72 KB (2 per CU)
        case 11: return launch_dma32<9, 2, 2, 4, 3>(g, st);      // 288x128, 32-deep stages, 78 KB (2 per CU): 576-patch towers tile M exactly
        case 21: return launch_dma<3, 2, 2, 2, 3>(g, st);     //  96x64,  4 waves, 3 x 20 KB (2 per CU): fc1 of the latency path (M = 576: 384 tiles instead of 576)
        case 14: return launch_ps64(g, st);                      //  64x64, software-pipelined k-steps, 64 KB (2 per CU): the latency path
        case 12: return aha_gemm_tile_p288_ok(g) && g->wide_epi ? aha_gemm_tile_p288(g, st) : launch_dma32<4, 4, 4, 2, 3>(g, st);   // persistent 288x256 (gemm_tile_p.hip)
        default: return hipErrorInvalidValue;
    }
}

static hipError_t check_tile_args(const GemmTileArgs* g) {
    if ((long)g->M * g->lda >= (1L << 31) || (long)g->N * g->ldw >= (1L << 31)) return hipErrorInvalidValue;
    if ((g->K & 7) || (g->lda & 7) || (g->ldw & 7) || (g->N & 3) || (g->ldc & 3) || (g->residual && (g->ldr & 3)) ||
        (g->rowadd && (g->ldra & 3)))
        return hipErrorInvalidValue;
    return hipSuccess;
}

// the predicate of aha_gemm_tile's choice of the persistent 288 x 256 kernel, for callers that want to hand it k-blocked operands
static bool tile_picks_p288(const GemmTileArgs* g) {
    return g_tile_dma == 1 && (g->K % TBK) == 0 && g->K >= 2 * TBK && g_tile_p288 && g_tile_epi && aha_gemm_tile_p288_ok(g) &&
           aha_gemm_tile_p288_efficiency(g, 256) >= (g_tile_p288 > 1 ? 0.01f * g_tile_p288 : 0.70f);
}
extern "C" int aha_gemm_tile_will_use_p288(const GemmTileArgs* g) {
    if (g->M <= 0 || g->N <= 0 || check_tile_args(g) != hipSuccess) return 0;
    GemmTileArgs gg = *g;
    gg.wide_epi = g_tile_epi;
    return tile_picks_p288(&gg) ? 1 : 0;
}

extern "C" hipError_t aha_gemm_tile(const GemmTileArgs* g_, hipStream_t st) {
    GemmTileArgs gg = *g_;
    gg.wide_epi = g_tile_epi;
    const GemmTileArgs* g = &gg;
    if (g->M <= 0 || g->N <= 0) return hipSuccess;
    if (hipError_t e = check_tile_args(g); e != hipSuccess) return e;
    const int nblk128 = ceil_div(g->N, 128) * ceil_div(g->M, 128);
    if (g_tile_dma && (g->K % TBK) == 0 && g->K >= 2 * TBK) {
        // Measured per shape (tools/diag/gemm_tile_sweep.py, M = 576 ... 18432 x the tower's four GEMM shapes): a CU
        // ingests ~55 GB/s by LDS-DMA whatever the tile or stage depth (8 stages were no faster than 3), so small
        // grids go to the variant with the most resident workgroups per CU (64x64, 3 per CU).  The 256x128 tile
        // (one per CU) wins when its grid fills most of its last round of 256 CUs, except on the thin N = K = 1024 GEMM;
        // with short K and a large grid the 32-deep variant (two workgroups per CU hide each other's waits) beats it.
        const int nblk_l = ceil_div(g->N, 128) * ceil_div(g->M, 256);
        const float util = (float)nblk_l / (256.f * ceil_div(nblk_l, 256));
        int v = g_tile_dma;
        if ((g->akb || g->ckb) && !(v == 1 && tile_picks_p288(g))) return hipErrorInvalidValue;   // k-blocked operands: the persistent kernel only (caller asks aha_gemm_tile_will_use_p288 first)
        if (v == 1 && tile_picks_p288(g)) {
            // throughput shapes whose 288 x 256 decomposition keeps the chip busy (tile padding x round quantisation >= 0.70:
            // every tower and projector GEMM from 8 frames of 576 patches up): the persistent kernel (gemm_tile_p.hip).
            // Tuning "tile_p288" = 0 keeps the round-2 selection below.
            v = 12;
        }
        if (v == 1) {
            // 576-patch towers: M is a multiple of 288.  The 288x128 tile is ~10 % slower per flop than 256x128 (11 fragment
            // reads per 18 MFMAs, 128 VGPRs) and wins exactly where it turns "one round of the chip's 512 resident workgroups
            // plus a stub" into one round: out-proj at 32 frames (576 -> 512 tiles: 63 -> 52 us), fc1 at 8 frames (65 -> 53 us).
            const int t288 = (g->M % 288 == 0) ? (g->M / 288) * ceil_div(g->N, 128) : 0;
            if (g->K < 2048 && t288 > 400 && t288 <= 512 && nblk_l > 512) v = 11;
            else if (g->K < 2048 && nblk_l >= 400) v = 8;     // 32-deep stages, two workgroups per CU: +7-10 % on the K = 1024 GEMMs
            else v = ((util >= 0.6f || (g->K >= 2048 && util >= 0.5f)) && (g->N > 1024 || g->K > 1024)) ? 2 : 5;
            // latency path (one or two frames): the narrow GEMMs (out-proj, fc2, patch embedding: N <= 1024) are <= 288 64x64 tiles,
            // one per CU, each a serial k-chain - the software-pipelined 64x64 kernel (M = 576: fc2 21.9 -> 16.4 us, out-proj 8.1 -> 6.6)
            if (v == 5 && g->N <= 1024 && g->M <= 1536) v = 14;
            // ... and its widest GEMM (fc1: N = 4096, K = 1024) takes 96-row tiles when they divide M: a CU then ingests (96 + 64) rows of
            // operands per 96 x 64 outputs instead of (64 + 64) per 64 x 64 - the latency path's GEMMs are bound by what a CU can take in
            // by LDS-DMA.  Measured at M = 576: 14.2 -> 11.1 us (QKV at N = 3072 is faster on the 64 x 64 tiles: 10.2 vs 11.7; the 64 x 32 and
            // 32 x 64 forms lose on every shape - profiles/r05_tile_sweep_latency_path.txt).
            if (v == 5 && g->N >= 4096 && g->K <= 1024 && g->M <= 1152 && g->M % 96 == 0) v = 21;
        }
        return launch_dma_variant(v, g, st);
    }
    if (g->akb || g->ckb) return hipErrorInvalidValue;
    if (nblk128 >= 384) {            // enough 128x128 tiles to fill 256 CUs at 2 workgroups each
        hipLaunchKernelGGL((gemm_tile_kernel<4>), dim3(nblk128), dim3(256), 0, st, *g);
    } else {
        const int nblk64 = ceil_div(g->N, 64) * ceil_div(g->M, 64);
        hipLaunchKernelGGL((gemm_tile_kernel<2>), dim3(nblk64), dim3(256), 0, st, *g);
    }
    return hipGetLastError();
}

// Developer hook (not part of include/aha_amd.h; used by tools/diag/gemm_tile_sweep.py): one plain GEMM
// C[M,N] = bf16(A[M,K] W[N,K]^T) with a forced kernel variant (0 = register-staged, >= 2 = LDS-DMA variant id).
extern "C" int aha_dev_gemm_tile(const void* A, const void* W, void* C, int M, int N, int K, int variant, void* st) {
    GemmTileArgs g{};
    g.A = (const bf16*)A; g.lda = K; g.M = M;
    g.W = (const bf16*)W; g.ldw = K; g.N = N;
    g.K = K; g.C = (bf16*)C; g.ldc = N;
    if (check_tile_args(&g) != hipSuccess || (variant >= 2 && (K % TBK || K < 2 * TBK))) return -22;
    const int saved = g_tile_dma;
    g_tile_dma = variant;
    hipError_t e = aha_gemm_tile(&g, (hipStream_t)st);
    g_tile_dma = saved;
    return e == hipSuccess ? 0 : -5;
}
