// Shared device/host helpers for the gfx950 kernels of the streaming path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (16 B)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;      // one 16x16 MFMA accumulator

#define AHA_WAVE 64
#define AHA_MAX_B 16            // streams batched in one lm_step
#define AHA_MAX_KEY_SPLITS 64    // key splits of the cache attention (the combine kernels merge them in chunks of 16)

static __device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
static __device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }
// Round to bf16 and back: the rounding point a torch bf16 op output has.  Done with explicit
// integer round-to-nearest-even: hipcc (ROCm 7.2) may keep excess precision across a
// (float)(__bf16)x round trip inside one expression and silently skip the rounding (observed in
// the RoPE re-rotation kernel: results equal to UNROUNDED arithmetic).  Finite inputs only.
static __device__ __forceinline__ float rbf(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}

// 16-byte write-through store (buffer_store_dwordx4 ... sc1): the bytes reach memory without a release fence, for tensors
// handed to other workgroups inside one launch (cdna_hip_programming.md G16, recipe R1).  byte_off < 4 GB.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
static __device__ __forceinline__ void store16_sc1(void* base, long byte_off, u32x4_t v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16);
}
static __device__ __forceinline__ void store8_sc1(void* base, long byte_off, unsigned long long v) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, (int)byte_off, 0, 16);
}

static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    // D[16x16] += A[16x32] * B[32x16]; lane l: A[row l&15][k 8(l>>4)+j], B[k 8(l>>4)+j][col l&15];
    // C/D: col = l&15, row = (l>>4)*4 + reg
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static __host__ __device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }
static __host__ __device__ __forceinline__ int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ---------------------------------------------------------------------------------------------
// Per-step description of the streams an lm_step works on.  It is DEVICE-RESIDENT: the host fills one (plan_stream),
// uploads it from a pinned ring slot once per step (1 KB, asynchronous) and every kernel reads it through the same constant
// pointer - so the launches of a step do not bake the stream state in and a captured HIP graph can be replayed.  Logical key index j of a stream maps to a
// physical cache slot through (n_fixed, ring_head, ring_cap):
//     j <  n_fixed : slot j                      (NONE/STATIC: everything; SINK: the sink tokens)
//     j >= n_fixed : slot n_fixed + (ring_head + (j - n_fixed)) % ring_cap
// ---------------------------------------------------------------------------------------------
struct StreamStep {
    bf16* k_base;        // [layers][kv_heads][cap][head_dim]
    bf16* v_base;
    int cap;             // slots per (layer, kv head)
    int n_fixed;         // see mapping above
    int ring_head;       // after this step's eviction
    int ring_cap;        // >= 1
    int len_after;       // keys the attention sees this step (Lk)
    int pos_base;        // RoPE position of new token 0 (= cache length before the step)
    int causal_off;      // key j visible to new token t  iff  j <= causal_off + t
    int write_base;      // logical index of new token 0's K/V slot; < 0: K/V of this step not stored
    int write_count;     // how many of the T new tokens are stored (STATIC first call: min(T, W))
    int n_rerot;         // SINK: number of kept keys to re-rotate this step (logical n_fixed .. +n_rerot)
    int rerot_row0;      // first row of the re-rotation table to use for kept key 0
    int pad_;
};

struct StepDesc {
    int B, T;
    StreamStep s[AHA_MAX_B];
};

static __device__ __forceinline__ int phys_slot(const StreamStep& s, int j) {
    if (j < s.n_fixed) return j;
    int r = s.ring_head + (j - s.n_fixed);
    if (r >= s.ring_cap) r -= s.ring_cap;       // ring_head < ring_cap and j - n_fixed < ring_cap
    return s.n_fixed + r;
}
