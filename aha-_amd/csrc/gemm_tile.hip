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
    const int bm = bid / tiles_n, bn = bid % tiles_n;
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

extern "C" hipError_t aha_gemm_tile(const GemmTileArgs* g, hipStream_t st) {
    if (g->M <= 0 || g->N <= 0) return hipSuccess;
    if ((long)g->M * g->lda >= (1L << 31) || (long)g->N * g->ldw >= (1L << 31)) return hipErrorInvalidValue;
    if ((g->K & 7) || (g->lda & 7) || (g->ldw & 7) || (g->N & 3) || (g->ldc & 3) || (g->residual && (g->ldr & 3)) ||
        (g->rowadd && (g->ldra & 3)))
        return hipErrorInvalidValue;
    const int nblk128 = ceil_div(g->N, 128) * ceil_div(g->M, 128);
    if (nblk128 >= 384) {            // enough 128x128 tiles to fill 256 CUs at 2 workgroups each
        hipLaunchKernelGGL((gemm_tile_kernel<4>), dim3(nblk128), dim3(256), 0, st, *g);
    } else {
        const int nblk64 = ceil_div(g->N, 64) * ceil_div(g->M, 64);
        hipLaunchKernelGGL((gemm_tile_kernel<2>), dim3(nblk64), dim3(256), 0, st, *g);
    }
    return hipGetLastError();
}
