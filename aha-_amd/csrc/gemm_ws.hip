// Weight-streaming skinny GEMM for the LM step:  Y[M,N] = X[M,K] * W[N,K]^T,  M = B*T <= 256.
//
// This is the kernel the LM step's HBM roofline is about (13.05 GB of decoder
// weights streamed once per step and shared by the B streams).  Design for gfx950:
//   * W is repacked ONCE at load into MFMA-fragment order  Wp[n_tile][k_step][lane][8 bf16]
//     (element W[nt*16 + (lane&15)][ks*32 + 8*(lane>>4) + j]), so a wave's weight stream for its
//     n-tile is a linear sequence of 1 KiB wave-loads (16 B/lane, perfectly coalesced, each byte
//     read exactly once, nontemporal) that goes straight to VGPRs as the MFMA A operand - no LDS
//     round trip for the operand that is streamed once (guide: "GEMV / M<=16 decode weights").
//   * X (M x K, L2-resident, re-read by every workgroup) goes through LDS in full rows
//     (coalesced 16-B loads, padded row stride => conflict-free ds_read_b128 B-fragments).
//   * v_mfma_f32_16x16x32_bf16 with A = W fragment, B = X^T fragment: the accumulator holds
//     4 consecutive n for one m per lane => 16-B (fp32) / 8-B (bf16) vector epilogue stores.
//   * K is split across blockIdx.y (split-K) so every CU streams; partial sums go to fp32 slabs
//     that the NEXT kernel's prologue reduces (qkv_finish / resid_norm) - no extra launch, no
//     atomics, bitwise reproducible.
//   * Software pipeline, three rotating weight register sets: while chunk c computes, chunks c+1
//     AND c+2 are in flight (a wave's throughput is in-flight bytes / HBM latency, and these
//     kernels have only 4-9 waves per CU).  Per step, in this order because vmcnt is in-order:
//     X(c+1) loads -> W(c+2) loads -> MFMAs of chunk c (wait: W(c), the oldest) -> LDS store of
//     X(c+1) (wait: X(c+1); the younger W(c+2) stays in flight across the barrier) -> barrier.
//   * What the ISA taught (ROCm 7.2 hipcc): (1) a guarded load becomes its own basic block and
//     the waits degrade to vmcnt(0): every load here is unconditional, indices are clamped into
//     valid memory instead; (2) a prefetch under an `if` makes the count of younger loads unknown
//     at the join (vmcnt(0) again): the steady-state body is one straight-line block of 3 steps,
//     the remainder is peeled at the FRONT as rotate-by-copy steps, the tail is fixed;
//     (3) sched_barrier(0) pins the X-before-W issue order the counted waits rely on.
#include "aha_kernels.h"

template <int MT, int NT, int KC, int WPB>
struct WsCfg {
    static constexpr int THREADS = WPB * 64;
    static constexpr int MPAD = MT * 16;
    static constexpr int STRIDE = KC * 32 + 8;                 // bf16 elements per LDS row
    static constexpr int BUF = MPAD * STRIDE;                  // elements per buffer
    static constexpr int LDS_BYTES = 2 * BUF * 2;
    static constexpr int XCH = MPAD * KC * 4;                  // 16-B chunks per X chunk tile
    static constexpr int XLD = (XCH + THREADS - 1) / THREADS;  // staging loads per thread
};

template <int MT, int NT, int KC, int EPI, int WPB>
__global__ __launch_bounds__(WPB * 64, 2) void gemm_ws_kernel(GemmWsArgs a) {
    using C = WsCfg<MT, NT, KC, WPB>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16* xs = reinterpret_cast<bf16*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const int tile0 = (blockIdx.x * WPB + wave) * NT;          // first n-tile of this wave
    const bool wave_active = tile0 < a.n_tiles;
    // Clamping instead of guarding is exact because: the packed weight has KS % 8 == 0 (zero-padded
    // k-steps, every chunk whole); surplus waves redo the last tile and skip the store; surplus X
    // rows / lanes duplicate valid chunks (identical bytes rewritten); X columns beyond Kx meet
    // zero weights.
    int tl[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) tl[j] = min(tile0 + j, a.n_tiles - 1);
    // split-K slice boundaries are placed in units of 8 k-steps whatever KC is, so every element is
    // summed in the same order for every M (tile configuration): a batched step is bit-identical to
    // the same rows stepped alone.
    const int NC8 = a.KS / 8;
    const int c0 = (int)(((long)blockIdx.y * NC8) / a.S) * (8 / KC), c1 = (int)(((long)(blockIdx.y + 1) * NC8) / a.S) * (8 / KC);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 wA[KC][NT], wB[KC][NT], wC[KC][NT];
    bf16x8 xr[C::XLD];

    auto load_w = [&](bf16x8 (&w)[KC][NT], int c) {
        const int ks0 = c * KC;
#pragma unroll
        for (int i = 0; i < KC; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                w[i][j] = __builtin_nontemporal_load(&a.Wp[((long)tl[j] * a.KS + (ks0 + i)) * 64 + lane]);
    };
    auto stage_load = [&](int c) {
        const int kbase = c * KC * 32;
#pragma unroll
        for (int i = 0; i < C::XLD; ++i) {
            const int idx = min(tid + i * C::THREADS, C::XCH - 1);
            const int row = min(idx / (KC * 4), a.M - 1);
            const int k = min(kbase + (idx % (KC * 4)) * 8, a.Kx - 8);
            xr[i] = *reinterpret_cast<const bf16x8*>(a.X + (long)row * a.ldx + k);
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < C::XLD; ++i) {
            const int idx = min(tid + i * C::THREADS, C::XCH - 1);
            const int row = idx / (KC * 4), cc = idx % (KC * 4);
            *reinterpret_cast<bf16x8*>(xs + buf * C::BUF + row * C::STRIDE + cc * 8) = xr[i];
        }
    };
    auto compute = [&](bf16x8 (&w)[KC][NT], int buf) {
        const bf16* xb = xs + buf * C::BUF + r16 * C::STRIDE + q * 8;
#pragma unroll
        for (int i = 0; i < KC; ++i) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xb + m * 16 * C::STRIDE + i * 32);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[m][j] = mfma16(w[i][j], xf, acc[m][j]);
            }
        }
    };
    // one steady-state step: chunk c computes from WCUR, chunk c+2 starts loading into WNEW
#define WS_STEP(WCUR, WNEW, BUFX)                  \
    stage_load(c + 1);                             \
    __builtin_amdgcn_sched_barrier(0);             \
    load_w(WNEW, c + 2);                           \
    __builtin_amdgcn_sched_barrier(0);             \
    compute(WCUR, (BUFX));                         \
    stage_store((BUFX) ^ 1);                       \
    __syncthreads();                               \
    ++c;

    const int n = c1 - c0;
    if (n > 0) {
        stage_load(c0);
        load_w(wA, c0);
        if (n > 1) load_w(wB, c0 + 1);
        stage_store(0);
        __syncthreads();
        int c = c0, buf = 0;
        const int steady = n > 2 ? n - 2 : 0;                   // steps that prefetch chunk c+2
        const int pre = steady % 3;
        for (int i = 0; i < pre; ++i) {                         // remainder, rotate-by-copy form
            WS_STEP(wA, wC, buf)
#pragma unroll
            for (int ii = 0; ii < KC; ++ii)
#pragma unroll
                for (int j = 0; j < NT; ++j) { wA[ii][j] = wB[ii][j]; wB[ii][j] = wC[ii][j]; }
            buf ^= 1;
        }
        for (int g = steady / 3; g > 0; --g) {                  // straight-line group of 3 steps
            WS_STEP(wA, wC, buf)
            WS_STEP(wB, wA, buf ^ 1)
            WS_STEP(wC, wB, buf)
            buf ^= 1;
        }
        // fixed tail: current = wA, next (if any) = wB
        if (n > 1) {
            stage_load(c + 1);
            compute(wA, buf);
            stage_store(buf ^ 1);
            __syncthreads();
            compute(wB, buf ^ 1);
        } else {
            compute(wA, buf);
        }
    }
#undef WS_STEP
    if (!wave_active) return;

    // ---- epilogue: acc[m][j][e] <-> row m*16 + r16, column (tile0+j)*16 + q*4 + e
    if constexpr (EPI == EPI_PARTIAL) {
        float* base = a.partial + (long)blockIdx.y * a.slab_stride;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = m * 16 + r16;
            if (row >= a.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = (tile0 + j) * 16 + q * 4;
                if (tile0 + j < a.n_tiles && col < a.ldp)
                    *reinterpret_cast<f32x4*>(base + (long)row * a.ldp + col) = acc[m][j];
            }
        }
    } else if constexpr (EPI == EPI_BF16) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = m * 16 + r16;
            if (row >= a.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = (tile0 + j) * 16 + q * 4;
                if (tile0 + j >= a.n_tiles || col >= a.N) continue;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[m][j][e];
                    if (a.bias) v += bf2f(a.bias[col + e]);
                    o[e] = f2bf(v);
                }
                *reinterpret_cast<bf16x4*>(a.out + (long)row * a.ldo + col) = o;
            }
        }
    } else if constexpr (EPI == EPI_SWIGLU) {
        // NT == 2: tile0 = gate tile, tile0+1 = up tile of the same 16 output columns
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
            *reinterpret_cast<bf16x4*>(a.out + (long)row * a.ldo + col) = o;
        }
    } else {  // EPI_F32_RBF: fp32 logits that passed through a bf16 Linear output
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = m * 16 + r16;
            if (row >= a.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = (tile0 + j) * 16 + q * 4;
                if (tile0 + j >= a.n_tiles) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < a.N) a.outf[(long)row * a.ldof + col + e] = rbf(acc[m][j][e]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Repack W[N][K] (row-major, ld = ldw) into fragment order.  dst tile index = nt*tile_stride +
// tile_off lets the caller interleave several matrices (gate/up pairs, q|k|v concatenation).
// KS may exceed ceil(K/32): the extra k-steps are zero-filled.
// ---------------------------------------------------------------------------------------------
__global__ void pack_w_kernel(const bf16* __restrict__ W, int N, int K, int ldw, bf16x8* __restrict__ Wp,
                              int KS, int n_tiles_src, int tile_stride, int tile_off) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_tiles_src * KS * 64;
    if (gid >= total) return;
    const int lane = (int)(gid & 63);
    const long rest = gid >> 6;
    const int ks = (int)(rest % KS), nt = (int)(rest / KS);
    const int n = nt * 16 + (lane & 15), k0 = ks * 32 + 8 * (lane >> 4);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (n < N && k0 + j < K) ? W[(long)n * ldw + k0 + j] : (bf16)0.0f;
    Wp[((long)(nt * tile_stride + tile_off) * KS + ks) * 64 + lane] = v;
}

// ---------------------------------------------------------------------------------------------
// Host-side dispatch
// ---------------------------------------------------------------------------------------------
template <int MT, int NT, int KC, int EPI, int WPB>
static hipError_t launch_ws_w(const GemmWsArgs& a, hipStream_t st) {
    using C = WsCfg<MT, NT, KC, WPB>;
    static bool attr_set = false;
    auto kern = gemm_ws_kernel<MT, NT, KC, EPI, WPB>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(ceil_div(a.n_tiles, WPB * NT), a.S);
    hipLaunchKernelGGL(kern, grid, dim3(C::THREADS), C::LDS_BYTES, st, a);
    return hipGetLastError();
}

static thread_local int g_wpb = 4;      // waves per workgroup for the next dispatch (2 or 4)

template <int MT, int NT, int KC, int EPI>
static hipError_t launch_ws(const GemmWsArgs& a, hipStream_t st) {
    if constexpr (MT <= 4) {            // single-stream shapes: 2-, 4- and 8-wave workgroups (tuning knob)
        if (g_wpb == 2) return launch_ws_w<MT, NT, KC, EPI, 2>(a, st);
        if (g_wpb == 8) return launch_ws_w<MT, NT, KC, EPI, 8>(a, st);
        return launch_ws_w<MT, NT, KC, EPI, 4>(a, st);
    } else {                            // batched shapes: always 8-wave workgroups (X image shared by 8 waves)
        return launch_ws_w<MT, NT, KC, EPI, 8>(a, st);
    }
}

template <int NT, int EPI>
static hipError_t dispatch_mt(const GemmWsArgs& a, hipStream_t st) {
    // KC (k-steps of 32 per pipeline chunk, one barrier per chunk) is the largest of {8,4,2,1} for which
    // three weight register sets + accumulators + X staging stay under 256 VGPRs without spills and the
    // two X buffers fit the 160 KB LDS (checked with -Rpass-analysis=kernel-resource-usage).
    const int mt = ceil_div(a.M, 16);
    if constexpr (NT == 1) {
        if (mt <= 1) return launch_ws<1, 1, 8, EPI>(a, st);
        if (mt <= 2) return launch_ws<2, 1, 8, EPI>(a, st);
        if (mt <= 3) return launch_ws<3, 1, 8, EPI>(a, st);
        if (mt <= 4) return launch_ws<4, 1, 4, EPI>(a, st);
        if (mt <= 6) return launch_ws<6, 1, 4, EPI>(a, st);
        if (mt <= 8) return launch_ws<8, 1, 4, EPI>(a, st);
        if (mt <= 12) return launch_ws<12, 1, 4, EPI>(a, st);
        if (mt <= 16) return launch_ws<16, 1, 2, EPI>(a, st);
        if (mt <= 20) return launch_ws<20, 1, 2, EPI>(a, st);
        if (mt <= 26) return launch_ws<26, 1, 1, EPI>(a, st);
    } else {
        if (mt <= 1) return launch_ws<1, 2, 4, EPI>(a, st);
        if (mt <= 2) return launch_ws<2, 2, 4, EPI>(a, st);
        if (mt <= 3) return launch_ws<3, 2, 4, EPI>(a, st);
        if (mt <= 4) return launch_ws<4, 2, 4, EPI>(a, st);
        if (mt <= 6) return launch_ws<6, 2, 4, EPI>(a, st);
        if (mt <= 8) return launch_ws<8, 2, 2, EPI>(a, st);
        if (mt <= 12) return launch_ws<12, 2, 1, EPI>(a, st);
        if (mt <= 16) return launch_ws<16, 2, 1, EPI>(a, st);
    }
    return hipErrorInvalidValue;               // caller chunks M
}

// Largest M each epilogue supports in one launch (the host loops over row chunks beyond it).
extern "C" int aha_gemm_ws_max_m(int epi) { return epi == EPI_SWIGLU ? 256 : 416; }

extern "C" hipError_t aha_gemm_ws(const GemmWsArgs* a, int epi, int wpb, hipStream_t st) {
    g_wpb = (wpb == 2 || wpb == 8) ? wpb : 4;
    switch (epi) {
        case EPI_PARTIAL: return dispatch_mt<1, EPI_PARTIAL>(*a, st);
        case EPI_BF16: return dispatch_mt<1, EPI_BF16>(*a, st);
        case EPI_SWIGLU: return dispatch_mt<2, EPI_SWIGLU>(*a, st);
        case EPI_F32_RBF: return dispatch_mt<1, EPI_F32_RBF>(*a, st);
    }
    return hipErrorInvalidValue;
}

extern "C" hipError_t aha_pack_w(const bf16* W, int N, int K, int ldw, bf16x8* Wp, int KS, int tile_stride,
                                 int tile_off, hipStream_t st) {
    const int nts = ceil_div(N, 16);
    const long total = (long)nts * KS * 64;
    hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, N, K, ldw, Wp, KS, nts,
                       tile_stride, tile_off);
    return hipGetLastError();
}
