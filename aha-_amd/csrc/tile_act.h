// tile_act.h -- activation functions of the tiled-GEMM epilogues (vision tower / projector), shared by every tile kernel so
// that all of them round at the same points and give an output element the same bits.
#pragma once
#include "aha_kernels.h"

// torch gelu(approximate='tanh'): 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715*x^3).  Evaluated through the identity
// 0.5*(1+tanh(u)) = 1/(1+exp(-2u)): x / (1 + 2^(-2u*log2(e))) - one v_exp_f32 and one v_rcp_f32 instead of libm's tanhf
// (~40 instructions; the epilogue of the fc1 GEMM applies it to 144 elements per lane), no cancellation anywhere (the
// 1 - 2/(e^2u + 1) form loses the small tail of 1 + tanh for x < -4), within a few fp32 ulps of the fp32 evaluation torch
// performs, i.e. the bf16-rounded result differs from it in about 1 element in 10^4, by one bf16 ulp.
static __device__ __forceinline__ float gelu_tanh_f(float x) {
    const float k = 0.7978845608028654f;
    const float u = k * (x + 0.044715f * x * x * x);
    const float t = __builtin_amdgcn_exp2f(-2.8853900817779268f * u);      // exp(-2u); +inf for u -> -inf gives x / inf = -0
    return x * __builtin_amdgcn_rcpf(1.0f + t);                             // v_rcp_f32: 1 ulp
}
static __device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
// transformers QuickGELUActivation on bf16 tensors: input * sigmoid(1.702 * input), each op rounded to bf16
static __device__ __forceinline__ float quick_gelu_bf16(float x) {
    const float t = rbf(1.702f * x);
    const float s = rbf(1.0f / (1.0f + __expf(-t)));
    return x * s;                                     // the caller rounds the product
}
// one output element of a Linear: bf16(acc + bias) -> activation -> bf16
static __device__ __forceinline__ float tile_act(float lin_plus_bias, int act) {
    float x = rbf(lin_plus_bias);                     // Linear output (bf16)
    if (act == ACT_GELU_TANH) x = rbf(gelu_tanh_f(x));
    else if (act == ACT_GELU_ERF) x = rbf(gelu_erf_f(x));
    else if (act == ACT_QUICK_GELU) x = rbf(quick_gelu_bf16(x));
    return x;
}
