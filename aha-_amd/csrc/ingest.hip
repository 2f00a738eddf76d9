// Frame ingest on the GPU: uint8 [h,w,3] (RGB or BGR, as decoded) -> aspect-preserving resize -> centred zero pad ->
// uint8 [3,S,S] RGB, the canvas aha_vit_encode consumes.  Integer arithmetic throughout, bit-exact with the
// reference's two resamplers (oracle/frame_ingest.py restates both; the Pillow one is pinned against Pillow):
//   method 0: Pillow Image.resize default (BICUBIC, 22-bit fixed-point coefficients, horizontal pass to uint8 then
//             vertical pass) + ImageOps.expand       -- LiveInferForDemo.load_one_frame, test/live_infer_for_video.py:98-121
//   method 1: OpenCV cv2.resize default (INTER_LINEAR, 11-bit fixed point) + copyMakeBorder + BGR2RGB
//             -- load_video_for_testing / load_video, test/inference.py:538-562, test/live_infer_for_video.py:49-71
// The coefficient tables depend only on (source size, method); they are built once per geometry on the host in the
// same double / float arithmetic as the libraries (no FMA contraction) and cached on the device.  One thread computes
// one canvas pixel (3 channels): for Pillow it evaluates the horizontal pass on the fly for the <= ksize rows its
// vertical pass needs (each clipped to uint8 exactly like Pillow's intermediate image), so no source-sized workspace
// exists.  HBM-bound in principle (h*w*3 bytes in, 3*S*S out); at 1080p the taps make it ALU/L2-bound: ~1300 integer
// MACs per pixel x 83k pixels.
#include <cmath>
#include <vector>

#include "aha_kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;      // Pillow Resample.c

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> PRECISION_BITS;         // arithmetic shift, then Pillow's clip8 lookup (clamp)
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void ingest_pil_kernel(IngestArgs a) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= a.S || y >= a.S) return;
    const int ox = x - a.left, oy = y - a.top;
    int r[3] = {0, 0, 0};
    if (ox >= 0 && ox < a.new_w && oy >= 0 && oy < a.new_h) {
        const int xmin = a.need_h ? a.xb[2 * ox] : ox, xn = a.need_h ? a.xb[2 * ox + 1] : 1;
        const int ymin = a.need_v ? a.yb[2 * oy] : oy, yn = a.need_v ? a.yb[2 * oy + 1] : 1;
        const int* xk = a.xk + (long)ox * a.xks;
        const int* yk = a.yk + (long)oy * a.yks;
        int av[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
        for (int j = 0; j < yn; ++j) {
            const uint8_t* row = a.src + ((long)(ymin + j) * a.w + xmin) * 3;
            int hv[3];
            if (a.need_h) {
                int ah[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
                for (int i = 0; i < xn; ++i) {
                    const int k = xk[i];
                    ah[0] += row[3 * i] * k; ah[1] += row[3 * i + 1] * k; ah[2] += row[3 * i + 2] * k;
                }
                hv[0] = clip8(ah[0]); hv[1] = clip8(ah[1]); hv[2] = clip8(ah[2]);    // Pillow's uint8 intermediate image
            } else {
                hv[0] = row[0]; hv[1] = row[1]; hv[2] = row[2];
            }
            if (a.need_v) {
                const int k = yk[j];
                av[0] += hv[0] * k; av[1] += hv[1] * k; av[2] += hv[2] * k;
            } else {
                r[0] = hv[0]; r[1] = hv[1]; r[2] = hv[2];
            }
        }
        if (a.need_v) { r[0] = clip8(av[0]); r[1] = clip8(av[1]); r[2] = clip8(av[2]); }
    }
    const long plane = (long)a.S * a.S, o = (long)y * a.S + x;
    a.out[(a.src_bgr ? 2 : 0) * plane + o] = (uint8_t)r[0];
    a.out[plane + o] = (uint8_t)r[1];
    a.out[(a.src_bgr ? 0 : 2) * plane + o] = (uint8_t)r[2];
}

__global__ __launch_bounds__(256) void ingest_cv_kernel(IngestArgs a) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= a.S || y >= a.S) return;
    const int ox = x - a.left, oy = y - a.top;
    int r[3] = {0, 0, 0};
    if (ox >= 0 && ox < a.new_w && oy >= 0 && oy < a.new_h) {
        if (!a.need_h && !a.need_v) {                       // same size: cv2.resize copies
            const uint8_t* p = a.src + ((long)oy * a.w + ox) * 3;
            r[0] = p[0]; r[1] = p[1]; r[2] = p[2];
        } else {
            const int x0 = a.xb[4 * ox], x1 = a.xb[4 * ox + 1], a0 = a.xb[4 * ox + 2], a1 = a.xb[4 * ox + 3];
            const int r0 = a.yb[4 * oy], r1 = a.yb[4 * oy + 1], b0 = a.yb[4 * oy + 2], b1 = a.yb[4 * oy + 3];
            const uint8_t* p00 = a.src + ((long)r0 * a.w + x0) * 3;
            const uint8_t* p01 = a.src + ((long)r0 * a.w + x1) * 3;
            const uint8_t* p10 = a.src + ((long)r1 * a.w + x0) * 3;
            const uint8_t* p11 = a.src + ((long)r1 * a.w + x1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int h0 = p00[c] * a0 + p01[c] * a1, h1 = p10[c] * a0 + p11[c] * a1;      // HResizeLinear (int rows)
                const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;   // VResizeLinear, 8-bit
                r[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
        }
    }
    const long plane = (long)a.S * a.S, o = (long)y * a.S + x;
    a.out[(a.src_bgr ? 2 : 0) * plane + o] = (uint8_t)r[0];
    a.out[plane + o] = (uint8_t)r[1];
    a.out[(a.src_bgr ? 0 : 2) * plane + o] = (uint8_t)r[2];
}

double bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

}  // namespace

// Pillow precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size): bounds [out][2] = (xmin, count), kk [out][ksize]
void aha_ingest_pil_tables(int in_size, int out_size, int* ksize_out, std::vector<int>* bounds, std::vector<int>* kk) {
    double scale = (double)in_size / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    bounds->assign((size_t)out_size * 2, 0);
    kk->assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            const double v = k[x] * (1 << PRECISION_BITS);
            (*kk)[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        (*bounds)[2 * xx] = xmin;
        (*bounds)[2 * xx + 1] = xmax;
    }
    *ksize_out = ksize;
}

// OpenCV resize.cpp INTER_LINEAR tables for one axis: [out][4] = (i0, i1, c0, c1) with 11-bit coefficients.
// horizontal: border indices clamp AND the fraction is zeroed; vertical: indices clamp, coefficients stay.
void aha_ingest_cv_tables(int src_size, int dst_size, bool horizontal, std::vector<int>* tab) {
    const double scale = 1.0 / ((double)dst_size / src_size);
    tab->assign((size_t)dst_size * 4, 0);
    for (int d = 0; d < dst_size; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        int i0, i1;
        if (horizontal) {
            if (s < 0) { s = 0; f = 0.f; }
            if (s >= src_size - 1) { s = src_size - 1; f = 0.f; }
            i0 = s;
            i1 = s + 1 < src_size ? s + 1 : src_size - 1;
        } else {
            i0 = s < 0 ? 0 : (s > src_size - 1 ? src_size - 1 : s);
            i1 = s + 1 < 0 ? 0 : (s + 1 > src_size - 1 ? src_size - 1 : s + 1);
        }
        const float c0 = (1.f - f) * 2048.f, c1 = f * 2048.f;
        (*tab)[4 * d] = i0;
        (*tab)[4 * d + 1] = i1;
        (*tab)[4 * d + 2] = (int)std::nearbyintf(c0);      // saturate_cast<short>(float) = cvRound: round half to even
        (*tab)[4 * d + 3] = (int)std::nearbyintf(c1);
    }
}

hipError_t aha_ingest_launch(const IngestArgs* a, int method, hipStream_t st) {
    dim3 grid(ceil_div(a->S, 16), ceil_div(a->S, 16));
    if (method == 0) hipLaunchKernelGGL(ingest_pil_kernel, grid, dim3(256), 0, st, *a);
    else hipLaunchKernelGGL(ingest_cv_kernel, grid, dim3(256), 0, st, *a);
    return hipGetLastError();
}
