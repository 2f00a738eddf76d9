// gemm_ws_body.h -- the weight-streaming skinny-GEMM workgroup body of gemm_ws.hip (one launch per GEMM).  See gemm_ws.hip for the
// design notes.
#pragma once
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

// The kernel body as a device function: (bx, by) = the launch's block indices; `xs` is the workgroup's LDS staging area
// (WsCfg::LDS_BYTES).  X is loaded first, so that the first LDS store does not wait for the weights.
template <int MT, int NT, int KC, int EPI, int WPB>
static __device__ __forceinline__ void gemm_ws_body(const GemmWsArgs& a, const int bx, const int by, bf16* xs) {
    using C = WsCfg<MT, NT, KC, WPB>;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const int tile0 = (bx * WPB + wave) * NT;          // first n-tile of this wave
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
    const int c0 = (int)(((long)by * NC8) / a.S) * (8 / KC), c1 = (int)(((long)(by + 1) * NC8) / a.S) * (8 / KC);

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
        for (int g = steady / 3; g > 0; --g) {                  // straight-line group of 3 steps
            WS_STEP(wA, wC, buf)
            WS_STEP(wB, wA, buf ^ 1)
            WS_STEP(wC, wB, buf)
            buf ^= 1;
        }
        // The remainder (steady % 3 prefetching steps) runs as the first steps of one more group, then the last two chunks - straight-line
        // code per remainder, the register sets never copied.  [r6] Rounds 1-5 peeled the remainder at the FRONT in rotate-by-copy form
        // (wA = wB; wB = wC after each step): the copy reads the set whose load was issued a few instructions earlier, i.e. every such step
        // waited a full memory latency - two of QKV's four chunks, two of o_proj's, two of gate/up's 28.
        // tail: two chunks left, W0 = current, W1 = next
#define WS_TAIL2(W0, W1, BUFX)                     \
    stage_load(c + 1);                             \
    compute(W0, (BUFX));                           \
    stage_store((BUFX) ^ 1);                       \
    __syncthreads();                               \
    compute(W1, (BUFX) ^ 1);
        if (n > 1) {
            const int r = steady % 3;
            if (r == 0) {
                WS_TAIL2(wA, wB, buf)
            } else if (r == 1) {
                WS_STEP(wA, wC, buf)
                WS_TAIL2(wB, wC, buf ^ 1)
            } else {
                WS_STEP(wA, wC, buf)
                WS_STEP(wB, wA, buf ^ 1)
                WS_TAIL2(wC, wA, buf)
            }
        } else {
            compute(wA, buf);
        }
    }
#undef WS_TAIL2
#undef WS_STEP
    if (!wave_active) return;

    // ---- epilogue: acc[m][j][e] <-> row m*16 + r16, column (tile0+j)*16 + q*4 + e
    if constexpr (EPI == EPI_PARTIAL) {
        float* base = a.partial + (long)by * a.slab_stride;
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

