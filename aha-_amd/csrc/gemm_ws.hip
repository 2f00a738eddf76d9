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

#include "gemm_ws_body.h"

template <int MT, int NT, int KC, int EPI, int WPB>
__global__ __launch_bounds__(WPB * 64, 2) void gemm_ws_kernel(GemmWsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    gemm_ws_body<MT, NT, KC, EPI, WPB>(a, blockIdx.x, blockIdx.y, reinterpret_cast<bf16*>(smem_raw));
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

// tuning "kc_small": k-steps per pipeline chunk of the NT = 1, MT <= 3 kernels (QKV, O, down at T <= 48).  Their per-wave
// streams are short (5-9 chunks of 8 k-steps), so finer chunks fill and drain the 3-deep pipeline faster: measured LM step
// 3.26 -> 3.20 ms with 4 (tools/tune_lm.py --sweep kc_small:8,4).  Sums do not depend on it (slices are in units of 8 k-steps).
static int g_kc_small = 4;
extern "C" void aha_gemm_ws_set_kc_small(int v) { g_kc_small = v == 4 ? 4 : 8; }
static thread_local int g_wpb = 4;      // waves per workgroup for the next dispatch (2 or 4)

template <int MT, int NT, int KC, int EPI>
static hipError_t launch_ws(const GemmWsArgs& a, hipStream_t st) {
    if constexpr (MT <= 4) {            // single-stream shapes: 2- to 8-wave workgroups (tuning knob)
        // The width decides how many CUs a GEMM's wave-tasks land on: e.g. gate/up = 1184 wave-tasks = 148 x 8, 198 x 6 or
        // 237 x 5 workgroups (one per CU); each CU ingests at most ~43-55 GB/s, so fewer waves per CU on more CUs can beat
        // the 8-wave shape even though 8 waves share one X tile best.
        if (g_wpb == 2) return launch_ws_w<MT, NT, KC, EPI, 2>(a, st);
        if (g_wpb == 3) return launch_ws_w<MT, NT, KC, EPI, 3>(a, st);
        if (g_wpb == 5) return launch_ws_w<MT, NT, KC, EPI, 5>(a, st);
        if (g_wpb == 6) return launch_ws_w<MT, NT, KC, EPI, 6>(a, st);
        if (g_wpb == 7) return launch_ws_w<MT, NT, KC, EPI, 7>(a, st);
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
        if (g_kc_small == 4 && mt <= 3) {       // default: finer pipeline chunks for the short split-K streams
            if (mt <= 1) return launch_ws<1, 1, 4, EPI>(a, st);
            if (mt <= 2) return launch_ws<2, 1, 4, EPI>(a, st);
            return launch_ws<3, 1, 4, EPI>(a, st);
        }
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
        if (mt <= 18) return launch_ws<18, 2, 1, EPI>(a, st);
        if (mt <= 20) return launch_ws<20, 2, 1, EPI>(a, st);
    }
    return hipErrorInvalidValue;               // caller chunks M
}

// Largest M each epilogue supports in one launch (the host loops over row chunks beyond it).
extern "C" int aha_gemm_ws_max_m(int epi) { return epi == EPI_SWIGLU ? 320 : 416; }

extern "C" hipError_t aha_gemm_ws(const GemmWsArgs* a, int epi, int wpb, hipStream_t st) {
    g_wpb = (wpb >= 2 && wpb <= 8) ? wpb : 4;
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
