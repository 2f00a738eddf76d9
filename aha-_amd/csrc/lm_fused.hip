// Fused LM-layer phases in ONE launch (first step towards a persistent layer kernel):
//     resid_norm (o_proj slabs + residual + post-attention RMSNorm)  ->  gate/up GEMM + SwiGLU  ->  down GEMM (slabs)
// Three of the layer's eight launches become one.  Each launch costs ~4.5 us of fixed time (boundary + ramp-up + tail)
// on top of its work, ~36 us of a 124 us layer; inside one launch the phases are separated by a device-scope grid
// barrier and the NEXT phase's first weight chunks are issued between arriving at the barrier and waiting on it, so the
// weight stream restarts while the slowest workgroup is still finishing.
//
// Grid barrier (MI355X_MICROARCH.md "Correctness boundaries", cdna_hip_programming.md G16): per-CU L1s and per-XCD L2s
// are not coherent, so a hand-off is: every wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane
// does an agent-scope release fence + drain and a relaxed agent-scope fetch_add on a monotonic 64-bit counter; waiters
// poll it relaxed, then ONE lane does an agent-scope acquire fence + drain, workgroup barrier, plain loads.  The spin is
// BOUNDED: a workgroup that waits too long sets *err and carries on (wrong data, flagged as NaN scores by the heads
// kernel) instead of hanging the GPU.  Every workgroup of the grid takes part in every barrier; the grid is at most one
// workgroup per CU (512 threads, <= 256 VGPRs), so all of them are resident together.
//
// The phase bodies are the SAME device functions the stand-alone kernels run (gemm_ws_body.h, resid_norm_body.h):
// bit-identical results (tests/test_gpu_parity.py::test_fused_mlp_block_is_bit_identical).
#include "gemm_ws_body.h"
#include "resid_norm_body.h"

namespace {

// Two-level arrival: workgroup w adds to group counter g = w % NG (each on its own 128-byte line); the arrival that
// completes a group (old % per_group == per_group - 1; every barrier adds exactly per_group to every group counter) adds to
// the top counter, which is the only word the waiters poll.  A flat counter made 256 workgroups serialise their
// agent-scope atomics on one address (~15 us per barrier measured); this needs grid % NG == 0.
struct GridBarrier {
    unsigned long long* ctr;         // [0]: top counter; [16 * (1 + g)]: group g
    unsigned long long target;       // top-counter value at which everyone has arrived
    int* err;
    int per_group;                   // workgroups per group (grid / NG), 0 = flat counter (ctr[0] counts workgroups)
    bool release;                    // false: every handed-off store was write-through (sc1), no release fence needed
    static constexpr int NG = 16;
    __device__ __forceinline__ void arrive() const {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's stores of the finished phase
        __syncthreads();
        if (threadIdx.x == 0) {
            if (release) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (per_group > 0) {
                const unsigned long long old = __hip_atomic_fetch_add(ctr + 16 * (1 + (blockIdx.x % NG)), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(old % (unsigned long long)per_group) == per_group - 1)
                    __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __device__ __forceinline__ void wait() const {
        if (threadIdx.x == 0) {
            int spins = 0;
            // once any wait has timed out the flag is sticky: later waits fall through at once (results are poisoned anyway)
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target &&
                   __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 18)) {                              // ~0.3 s: never hang the GPU
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
};

}  // namespace

template <int MT>
struct MlpCfg {
    static constexpr int KC_GU = 4;                                     // gemm_ws dispatch table, NT = 2, MT <= 4
    static constexpr int KC_DN = MT <= 3 ? 8 : 4;                       // NT = 1
    using GU = WsCfg<MT, 2, KC_GU, 8>;
    using DN = WsCfg<MT, 1, KC_DN, 8>;
    static constexpr int XS_BYTES = GU::LDS_BYTES > DN::LDS_BYTES ? GU::LDS_BYTES : DN::LDS_BYTES;
    static constexpr int LDS_BYTES = XS_BYTES + 64;
};

// SC1: the two tensors that cross a barrier (xn: A -> B, act: B -> C) are stored write-through and the barriers skip the
// release fence (one L2 write-back per workgroup per barrier); the waiters' acquire stays.
template <int MT, bool SC1>
__global__ __launch_bounds__(512, 2) void lm_mlp_block_kernel(MlpBlockArgs p) {
    using C = MlpCfg<MT>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16* xs = reinterpret_cast<bf16*>(smem_raw);
    float* red = reinterpret_cast<float*>(smem_raw + C::XS_BYTES);
    const int wg = blockIdx.x, P = gridDim.x;

    // ---- phase A: o_proj slabs + residual -> h, post-attention RMSNorm -> xn (one row per workgroup)
    for (int row = wg; row < p.M; row += P) resid_norm_row<SC1>(p.rn, row, red);

    // ---- phase B: gate/up + SwiGLU (needs every row of xn): arrive, prefetch weights, wait
    const int pg = p.per_group;
    const unsigned long long per_bar = pg > 0 ? GridBarrier::NG : (unsigned long long)P;   // top-counter increments per barrier
    GridBarrier b1{p.ctr, p.base + per_bar, p.err, pg, !SC1};
    b1.arrive();
    auto w1 = [&] { b1.wait(); };
    if (wg < p.gu_blocks) gemm_ws_body<MT, 2, C::KC_GU, EPI_SWIGLU, 8, true, decltype(w1), SC1>(p.gu, wg, 0, xs, w1);
    else b1.wait();

    // ---- phase C: down projection, split-K slabs (needs every column of act)
    GridBarrier b2{p.ctr, p.base + 2 * per_bar, p.err, pg, !SC1};
    b2.arrive();
    if (wg < p.dn_blocks_x * p.dn.S) gemm_ws_body<MT, 1, C::KC_DN, EPI_PARTIAL, 8, true>(p.dn, wg % p.dn_blocks_x, wg / p.dn_blocks_x, xs, [&] { b2.wait(); });
    else b2.wait();
}

template <int MT, bool SC1>
static hipError_t launch_mlp(const MlpBlockArgs& p, int grid, hipStream_t st) {
    using C = MlpCfg<MT>;
    static bool attr_set = false;
    auto kern = lm_mlp_block_kernel<MT, SC1>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C::LDS_BYTES, st, p);
    return hipGetLastError();
}

// How far one launch advances the top counter: 2 barriers x (16 group completions, or `grid` arrivals when flat).
extern "C" int aha_lm_mlp_block_counter_step(int grid) { return 2 * (grid % GridBarrier::NG == 0 ? GridBarrier::NG : grid); }

// grid = workgroups launched (<= CUs, >= the widest phase); p.gu_blocks / p.dn_blocks_x are filled here.
extern "C" hipError_t aha_lm_mlp_block(MlpBlockArgs* p, int grid, hipStream_t st) {
    const int mt = ceil_div(p->M, 16);
    if (mt < 1 || mt > 4) return hipErrorInvalidValue;
    p->gu_blocks = ceil_div(p->gu.n_tiles, 8 * 2);
    p->dn_blocks_x = ceil_div(p->dn.n_tiles, 8);
    p->per_group = grid % GridBarrier::NG == 0 ? grid / GridBarrier::NG : 0;
    if (p->gu.S != 1 || p->gu_blocks > grid || p->dn_blocks_x * p->dn.S > grid || p->M > grid) return hipErrorInvalidValue;
    if (p->sc1) {
        switch (mt) {
            case 1: return launch_mlp<1, true>(*p, grid, st);
            case 2: return launch_mlp<2, true>(*p, grid, st);
            case 3: return launch_mlp<3, true>(*p, grid, st);
            default: return launch_mlp<4, true>(*p, grid, st);
        }
    }
    switch (mt) {
        case 1: return launch_mlp<1, false>(*p, grid, st);
        case 2: return launch_mlp<2, false>(*p, grid, st);
        case 3: return launch_mlp<3, false>(*p, grid, st);
        default: return launch_mlp<4, false>(*p, grid, st);
    }
}
