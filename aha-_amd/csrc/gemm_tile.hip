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


#define TBK 64

static __device__ __forceinline__ float gelu_tanh_f(float x) {
    // torch gelu(approximate='tanh'): 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))
    const float k = 0.7978845608028654f;
    const float inner = k * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}
static __device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }

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
                if (g.residual) x = rbf(bf2f(rv[j][e]) + x);
                if (g.rowadd) x = rbf(x + bf2f(pv[j][e]));
                o[e] = f2bf(x);
            }
            *reinterpret_cast<bf16x4*>(g.C + (long)m * g.ldc + n) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant for the throughput shapes (M >= 1024, K % 64 == 0): 256x128 block tile, 8 waves
// (4 x 2, each 64x64), THREE 48 KB stages filled by global_load_lds_dwordx4 (no staging VGPRs, so
// two k-tiles are in flight while one computes - the register-staged kernel above can only keep
// one, and a second register set spills).  Per k-tile: counted `s_waitcnt vmcnt(6)` (this wave's
// six DMA pieces of tile kt have landed, tile kt+1 stays in flight) -> raw s_barrier (everyone's
// pieces landed; everyone finished reading the stage about to be refilled) -> issue tile kt+2 ->
// ds_read + MFMA on tile kt.  LDS image = the same 128-B rows with the chunk index XOR-swizzled by
// (row & 7); an LDS-DMA writes lane-linear, so the swizzle is applied to the SOURCE address
// (guide rule 21).  One __shared__ object only; no ordinary global loads inside the loop.
// ---------------------------------------------------------------------------------------------
#define DBM 256
#define DBN 128
#define DSTAGES 3

__global__ __launch_bounds__(512, 2) void gemm_tile_dma_kernel(GemmTileArgs g) {
    extern __shared__ __attribute__((aligned(16))) char dsm_raw[];
    bf16* lds = reinterpret_cast<bf16*>(dsm_raw);
    constexpr int STAGE = (DBM + DBN) * TBK;                       // elements per stage

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = ceil_div(g.N, DBN), tiles_m = ceil_div(g.M, DBM);
    const int nblk = tiles_n * tiles_m;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    // Grouped rasterisation: consecutive ids walk 4 m-panels x all n-tiles column by column, so the ~32
    // workgroups one XCD runs together share 4 A panels and ~8 W panels (fits its 4 MB L2).  With plain row-major
    // order they shared 1 A panel and 32 W panels (8 MB of W): W thrashed L2 and streamed from Infinity Cache
    // (measured: waves parked ~45 % on vmcnt/barrier, zero LDS conflicts).
    constexpr int GROUP_M = 4;
    const int per_group = GROUP_M * tiles_n, grp = bid / per_group, first_m = grp * GROUP_M;
    const int gmn = min(tiles_m - first_m, GROUP_M), inner = bid % per_group;
    const int bm = first_m + inner % gmn, bn = inner / gmn;
    const int m0 = bm * DBM, n0 = bn * DBN;
    const int nk = g.K / TBK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // DMA piece p of a stage = 512 consecutive 16-B chunks of the stage image (chunk c -> row c/8,
    // slot c%8); lane's source chunk = slot ^ (row & 7).  A: 4 pieces, W: 2 pieces per thread.
    int aoff[4], woff[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 512 + tid, row = c >> 3, ch = (c & 7) ^ (row & 7);
        aoff[i] = min(m0 + row, g.M - 1) * g.lda + ch * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = i * 512 + tid, row = c >> 3, ch = (c & 7) ^ (row & 7);
        woff[i] = min(n0 + row, g.N - 1) * g.ldw + ch * 8;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma = [&](int kt, int stage) {
        const int k0 = min(kt, nk - 1) * TBK;                      // past the end: refill a dead stage (keeps vmcnt counts fixed)
        bf16* sa = lds + stage * STAGE;
        bf16* sb = sa + DBM * TBK;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(g.A + aoff[i] + k0), (lptr_t)(sa + (i * 512 + wave * 64) * 8), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(g.W + woff[i] + k0), (lptr_t)(sb + (i * 512 + wave * 64) * 8), 16, 0, 0);
    };
    auto compute = [&](int stage) {
        const bf16* sa = lds + stage * STAGE;
        const bf16* sb = sa + DBM * TBK;
#pragma unroll
        for (int ks = 0; ks < TBK / 32; ++ks) {
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + r16;
                af[i] = *reinterpret_cast<const bf16x8*>(&sa[row * TBK + (((ks * 4 + q) ^ (row & 7)) << 3)]);
                const int wrow = wn * 64 + i * 16 + r16;
                wf[i] = *reinterpret_cast<const bf16x8*>(&sb[wrow * TBK + (((ks * 4 + q) ^ (wrow & 7)) << 3)]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
        }
    };

    dma(0, 0);
    dma(1, 1);
    int st_cur = 0, st_new = 2;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           // tile kt (this wave's pieces) landed; kt+1 in flight
        __builtin_amdgcn_s_barrier();
        dma(kt + 2, st_new);
        compute(st_cur);
        st_cur = st_cur == 2 ? 0 : st_cur + 1;
        st_new = st_new == 2 ? 0 : st_new + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // nothing may still target LDS when the block retires

    // epilogue (same rounding points as gemm_tile_kernel)
    const bf16x4 z4 = {0, 0, 0, 0};
    bf16x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = min(n0 + wn * 64 + j * 16 + q * 4, g.N - 4);
        bv[j] = g.bias ? *reinterpret_cast<const bf16x4*>(g.bias + n) : z4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + r16;
        const int mc = min(m, g.M - 1);
        bf16x4 rv[4], pv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = min(n0 + wn * 64 + j * 16 + q * 4, g.N - 4);
            rv[j] = g.residual ? *reinterpret_cast<const bf16x4*>(g.residual + (long)mc * g.ldr + n) : z4;
            pv[j] = g.rowadd ? *reinterpret_cast<const bf16x4*>(g.rowadd + (long)(mc % g.rowadd_period) * g.ldra + n) : z4;
        }
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + q * 4;
            if (n >= g.N) continue;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = rbf(acc[i][j][e] + bf2f(bv[j][e]));
                if (g.act == ACT_GELU_TANH) x = rbf(gelu_tanh_f(x));
                else if (g.act == ACT_GELU_ERF) x = rbf(gelu_erf_f(x));
                if (g.residual) x = rbf(bf2f(rv[j][e]) + x);
                if (g.rowadd) x = rbf(x + bf2f(pv[j][e]));
                o[e] = f2bf(x);
            }
            *reinterpret_cast<bf16x4*>(g.C + (long)m * g.ldc + n) = o;
        }
    }
}

static int g_tile_dma = 1;       // tuning: 0 disables the LDS-DMA variant, 2 forces it whenever the shape allows (tests)
extern "C" void aha_gemm_tile_set_dma(int on) { g_tile_dma = on; }

extern "C" hipError_t aha_gemm_tile(const GemmTileArgs* g, hipStream_t st) {
    if (g->M <= 0 || g->N <= 0) return hipSuccess;
    if ((long)g->M * g->lda >= (1L << 31) || (long)g->N * g->ldw >= (1L << 31)) return hipErrorInvalidValue;
    if ((g->K & 7) || (g->lda & 7) || (g->ldw & 7) || (g->N & 3) || (g->ldc & 3) || (g->residual && (g->ldr & 3)) ||
        (g->rowadd && (g->ldra & 3)))
        return hipErrorInvalidValue;
    const int nblk128 = ceil_div(g->N, 128) * ceil_div(g->M, 128);
    const int nblk_dma = ceil_div(g->N, DBN) * ceil_div(g->M, DBM);
    if (g_tile_dma && (g->K % TBK) == 0 && g->K >= 2 * TBK && (nblk_dma >= 256 || g_tile_dma == 2)) {
        static bool attr_set = false;
        constexpr int lds_bytes = DSTAGES * (DBM + DBN) * TBK * 2;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_tile_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_tile_dma_kernel, dim3(nblk_dma), dim3(512), lds_bytes, st, *g);
    } else if (nblk128 >= 384) {     // enough 128x128 tiles to fill 256 CUs at 2 workgroups each
        hipLaunchKernelGGL((gemm_tile_kernel<4>), dim3(nblk128), dim3(256), 0, st, *g);
    } else {
        const int nblk64 = ceil_div(g->N, 64) * ceil_div(g->M, 64);
        hipLaunchKernelGGL((gemm_tile_kernel<2>), dim3(nblk64), dim3(256), 0, st, *g);
    }
    return hipGetLastError();
}
