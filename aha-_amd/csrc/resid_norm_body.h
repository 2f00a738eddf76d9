// resid_norm_body.h -- split-K reduce + residual add + RMSNorm of one row (elementwise.hip: resid_norm_kernel).
#pragma once
#include "aha_kernels.h"

// One 16-B chunk of the row per thread (blockDim = H/8 rounded up to waves, <= 1024): every slab
// load of the thread is independent and issued back to back, so the kernel costs about two
// memory latencies instead of S of them.
static __device__ __forceinline__ float block_sum_any(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// The 8 elements of chunk c of `row`: split-K slab sum (or the bf16 Linear output), residual add, the updated residual stored; f8 = the
// new residual values, return = their sum of squares.  Shared by resid_norm_row and by the layer engine's row phase (lm_engine.hip),
// which must give the same bits: the squares are accumulated with explicit FMAs (left to the compiler, one kernel fused them and
// another did not).
static __device__ __forceinline__ float resid_chunk(const ResidNormArgs& a, const int row, const int c, float (&f8)[8]) {
    float lin[8];
    if (a.partial) {
#pragma unroll
        for (int e = 0; e < 8; ++e) lin[e] = 0.f;
        const float* p = a.partial + (long)row * a.ldp + c * 8;
        for (int s0 = 0; s0 < a.S; s0 += 8) {
            f32x4 p0[8], p1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < a.S) {
                    p0[j] = *reinterpret_cast<const f32x4*>(p + (s0 + j) * a.slab_stride);
                    p1[j] = *reinterpret_cast<const f32x4*>(p + (s0 + j) * a.slab_stride + 4);
                }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < a.S) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lin[e] += p0[j][e]; lin[4 + e] += p1[j][e]; }
                }
        }
    } else {
        const bf16x8 lv = *reinterpret_cast<const bf16x8*>(a.lin_bf16 + (long)row * a.ldl + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) lin[e] = bf2f(lv[e]);
    }
    const bf16x8 hh = *reinterpret_cast<const bf16x8*>(a.h + (long)row * a.ldh + c * 8);
    bf16x8 ho;
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float f = rbf(bf2f(hh[e]) + rbf(lin[e]));
        f8[e] = f;
        ho[e] = f2bf(f);
        ss = __builtin_fmaf(f, f, ss);
    }
    *reinterpret_cast<bf16x8*>(a.h + (long)row * a.ldh + c * 8) = ho;
    return ss;
}
static __device__ __forceinline__ float resid_rstd(const ResidNormArgs& a, const float ss) { return rsqrtf(ss / (float)a.H + a.eps); }
static __device__ __forceinline__ bf16x8 resid_scale(const float (&f8)[8], const bf16x8 wv, const float rstd) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(wv[e]) * rbf(f8[e] * rstd));
    return o;
}

// One row of resid_norm as a device function (`red`: 16 floats of LDS); called by resid_norm_kernel (one workgroup per
// row).  All threads of the workgroup must call it together.
// XN_SC1: xn stored write-through (sc1) for consumer workgroups of the SAME launch (lm_stream.hip)
template <bool XN_SC1 = false>
static __device__ __forceinline__ void resid_norm_row(const ResidNormArgs& a, const int row, float* red) {
    const int nch = a.H >> 3;
    float ss = 0.f;
    for (int c0 = 0; c0 < nch; c0 += blockDim.x) {          // one pass when H/8 <= blockDim
        const int c = c0 + threadIdx.x;
        float f8[8];
        // the norm weight of this thread's chunk is fetched with the slabs, not behind the row reduction's barriers (the compiler keeps a
        // load on its side of a barrier: it was one more dependent L2 round trip in a kernel that is nothing but round trips)
        bf16x8 wv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c < nch && nch <= (int)blockDim.x) wv = *reinterpret_cast<const bf16x8*>(a.w + c * 8);
        if (c < nch) ss += resid_chunk(a, row, c, f8);
        if (nch <= (int)blockDim.x) {                       // common case: keep the row in registers
            ss = block_sum_any(ss, red);
            const float rstd = resid_rstd(a, ss);
            if (c < nch) {
                const bf16x8 o = resid_scale(f8, wv, rstd);
                // xkb: xn k-blocked [H/32][xkb rows][32] for the mid-M GEMM (gemm_wl.hip), else row-major
                const long xo = a.xkb ? ((long)(c >> 2) * a.xkb + row) * 32 + (c & 3) * 8 : (long)row * a.ldx + c * 8;
                if constexpr (XN_SC1) store16_sc1(a.xn, xo * 2, __builtin_bit_cast(u32x4_t, o));
                else *reinterpret_cast<bf16x8*>(a.xn + xo) = o;
            }
            return;
        }
    }
    // H/8 > blockDim (H > 8192): second pass re-reads the updated residual row
    ss = block_sum_any(ss, red);
    const float rstd = rsqrtf(ss / (float)a.H + a.eps);
    for (int c = threadIdx.x; c < nch; c += blockDim.x) {
        const bf16x8 hh = *reinterpret_cast<const bf16x8*>(a.h + (long)row * a.ldh + c * 8);
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(a.w + c * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(wv[e]) * rbf(bf2f(hh[e]) * rstd));
        const long xo = a.xkb ? ((long)(c >> 2) * a.xkb + row) * 32 + (c & 3) * 8 : (long)row * a.ldx + c * 8;
        *reinterpret_cast<bf16x8*>(a.xn + xo) = o;
    }
}

