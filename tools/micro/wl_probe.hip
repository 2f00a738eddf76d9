// Structural probes of the mid-M weight-streaming GEMM loop (down_proj at M = 288: X [288][18944] k-blocked, W [3584][18944] packed,
// split-K 8 -> fp32 slabs), outside the library so that a variant costs a few lines.  Variants:
//   0  the shipped loop: 8 waves, every wave issues its share of the LDS-DMA pieces, one barrier per k-step
//   1  producer / consumer: 8 MFMA waves + 4 DMA waves (one per SIMD); the MFMA waves never touch vmcnt
//   2  as 1 with the fragment reads of k-step k+1 issued before the MFMAs of k-step k
//   3  as 0 with 16 waves (4 per SIMD), wave tile 9 row tiles x 1 n-tile
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/wl_probe tools/micro/wl_probe.hip && /tmp/wl_probe
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

constexpr int M = 288, MT = 18, N = 3584, K = 18944, KS = K / 32, S = 8, NTILES = N / 16;
constexpr int NTB = 8;                       // n-tiles per workgroup
constexpr int NB = MT + NTB, STAGE = NB * 512;

struct Args { const bf16* X; const bf16x8* Wp; float* partial; };

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int VAR, int STAGES, int ABL = 0>
__global__ __launch_bounds__(VAR == 1 || VAR == 2 ? 768 : (VAR == 3 ? 1024 : 512)) void probe(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;
    const int ks0 = blockIdx.y * (KS / S), nk = KS / S;                 // 74 k-steps per slice
    constexpr int NCONS = VAR == 3 ? 16 : 8;                            // MFMA waves
    constexpr bool SPLIT = VAR == 1 || VAR == 2;                        // dedicated DMA waves
    constexpr int NPROD = SPLIT ? 4 : NCONS;                            // waves that issue DMA
    constexpr int P = (NB + NPROD - 1) / NPROD;                         // pieces per DMA wave per k-step (surplus: re-load the last piece)
    const bool producer = !SPLIT || wave >= NCONS;
    const int pw = SPLIT ? wave - NCONS : wave;
    const long xkstride = (long)M * 32;
    // piece p < MT: X block p (16 rows x 64 B, chunk swizzle on the source); p >= MT: W tile p - MT
    const bf16* src[P]; long kstr[P]; int dst[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int p = min(pw + i * NPROD, NB - 1);
        if (p < MT) {
            const int row = p * 16 + (lane >> 2), c = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
            src[i] = a.X + (long)row * 32 + c * 8; kstr[i] = xkstride;
        } else {
            src[i] = reinterpret_cast<const bf16*>(a.Wp + ((long)(blockIdx.x * NTB + p - MT) * KS) * 64 + lane); kstr[i] = 512;
        }
        dst[i] = p * 512;
    }
    auto dma = [&](int kt, int stage) {
        const int ks = ks0 + min(kt, nk - 1);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const bool isx = min(pw + i * NPROD, NB - 1) < MT;      // wave-uniform
            // ablations 16 / 32: after the prologue X (W) pieces re-load k-step 0 (L2-hot, no new bytes) instead of streaming
            const int kk = ((ABL & 16) && isx && kt >= STAGES - 1) || ((ABL & 32) && !isx && kt >= STAGES - 1) ? ks0 : ks;
            __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (long)kk * kstr[i]), (lptr_t)(lds + stage * STAGE + dst[i]), 16, 0, 0);
        }
        return;
#pragma unroll
        for (int i = 0; i < P; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (long)ks * kstr[i]), (lptr_t)(lds + stage * STAGE + dst[i]), 16, 0, 0);
    };
    constexpr int MH = VAR == 3 ? 9 : 9, NJ = VAR == 3 ? 1 : 2;        // wave tile: 9 row tiles x NJ n-tiles
    const int wm = VAR == 3 ? wave / 8 : wave / 4, wn = VAR == 3 ? wave % 8 : wave % 4;
    f32x4 acc[MH][NJ];
#pragma unroll
    for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int xslot = (r16 * 4 + (q ^ ((4 - (r16 >> 2)) & 3))) * 8;
    const int xoff = wm * MH * 512 + xslot, woff = (MT + wn * NJ) * 512 + lane * 8;
    bf16x8 xf[MH], wf[NJ], xg[MH], wg[NJ];
    auto reads = [&](int stage, bf16x8 (&x)[MH], bf16x8 (&w)[NJ]) {
        const bf16* sa = lds + stage * STAGE;
#pragma unroll
        for (int j = 0; j < NJ; ++j) w[j] = *reinterpret_cast<const bf16x8*>(sa + woff + j * 512);
#pragma unroll
        for (int i = 0; i < MH; ++i) x[i] = *reinterpret_cast<const bf16x8*>(sa + xoff + i * 512);
    };
    auto mfmas = [&](bf16x8 (&x)[MH], bf16x8 (&w)[NJ]) {
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(w[j], x[i], acc[i][j]);
    };
    if (producer)
#pragma unroll
        for (int s = 0; s < (VAR == 4 ? STAGES : STAGES - 1); ++s) dma(s, s);
    int st_cur = 0, st_new = STAGES - 1;
    if constexpr (VAR == 4) {
        // fragments of k-step h are read during k-step h-1 (two register sets); a wave drains its own LDS reads (lgkmcnt) before
        // each barrier, so the stage whose fragments everyone holds in registers can be refilled right behind the barrier
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 1) * P) : "memory");
        __builtin_amdgcn_s_barrier();
        reads(0, xf, wf);
        for (int h = 0; h < nk; h += 2) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((STAGES - 2) * P) : "memory");   // stage h+1 landed (mine); my reads of stage h done
            __builtin_amdgcn_s_barrier();
            dma(h + STAGES, h % STAGES);
            reads((h + 1) % STAGES, xg, wg);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(xf, wf);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((STAGES - 2) * P) : "memory");   // stage h+2 landed
            __builtin_amdgcn_s_barrier();
            dma(h + 1 + STAGES, (h + 1) % STAGES);
            reads((h + 2) % STAGES, xf, wf);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(xg, wg);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (!SPLIT) {
        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (ABL & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * P) : "memory");
            if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
            if constexpr (!(ABL & 2)) dma(kt + STAGES - 1, st_new);
            if constexpr (!(ABL & 4)) reads(st_cur, xf, wf); else if (kt == 0) reads(0, xf, wf);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 1)) mfmas(xf, wf); else { acc[0][0][0] += (float)xf[kt % MH][0] + (float)wf[kt % NJ][1]; }
            st_cur = st_cur == STAGES - 1 ? 0 : st_cur + 1;
            st_new = st_new == STAGES - 1 ? 0 : st_new + 1;
        }
    } else if (producer) {
        // VAR 2 lets the consumers read stage kt+1 during iteration kt: it must have landed at barrier kt, so one stage less in flight
        constexpr int AHEAD = VAR == 2 ? STAGES - 2 : STAGES - 1;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((VAR == 2 ? STAGES - 3 : STAGES - 2) * P) : "memory");
            __builtin_amdgcn_s_barrier();
            if (VAR == 2 && kt == 0) { st_new = STAGES - 1; }
            dma(kt + STAGES - 1, st_new);
            st_new = st_new == STAGES - 1 ? 0 : st_new + 1;
            (void)AHEAD;
        }
    } else if constexpr (VAR == 1) {
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();
            reads(st_cur, xf, wf);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(xf, wf);
            st_cur = st_cur == STAGES - 1 ? 0 : st_cur + 1;
        }
    } else {                                                           // VAR 2 consumer: two fragment sets, reads one k-step ahead
        __builtin_amdgcn_s_barrier();                                  // barrier 0: stages 0 and 1 landed
        reads(0, xf, wf);
        for (int kt = 0; kt < nk; kt += 2) {
            reads((kt + 1) % STAGES, xg, wg);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(xf, wf);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                              // barrier kt+1: stage kt+2 landed, stage kt free
            reads((kt + 2) % STAGES, xf, wf);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(xg, wg);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < nk) __builtin_amdgcn_s_barrier();             // barrier kt+2
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SPLIT && producer) return;
    float* base = a.partial + (long)blockIdx.y * M * N;
#pragma unroll
    for (int i = 0; i < MH; ++i) {
        const int row = (wm * MH + i) * 16 + r16;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = (blockIdx.x * NTB + wn * NJ + j) * 16 + q * 4;
            *reinterpret_cast<f32x4*>(base + (long)row * N + col) = acc[i][j];
        }
    }
}

template <int VAR, int STAGES, int ABL = 0>
static float run(const char* name, Args* args, int ncopy, std::vector<float>* out) {
    constexpr int threads = VAR == 1 || VAR == 2 ? 768 : (VAR == 3 ? 1024 : 512);
    constexpr int LDS = STAGES * NB * 1024;
    auto kern = probe<VAR, STAGES, ABL>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    dim3 grid(NTILES / NTB, S);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(threads), LDS, 0, args[i % ncopy]);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 24;
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, grid, dim3(threads), LDS, 0, args[i % ncopy]);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipError_t e = hipGetLastError();
    hipLaunchKernelGGL(kern, grid, dim3(threads), LDS, 0, args[0]);
    out->resize((size_t)S * M * N);
    hipMemcpy(out->data(), args[0].partial, out->size() * 4, hipMemcpyDeviceToHost);
    printf("%-64s %7.1f us  (%s)\n", name, ms / n * 1e3, hipGetErrorString(e));
    return ms / n * 1e3f;
}

int main() {
    const int ncopy = 3;
    std::vector<Args> args(ncopy);
    std::vector<bf16> hx((size_t)M * K), hw((size_t)N * K);
    srand(1);
    for (auto& v : hx) v = (bf16)((rand() % 2001 - 1000) * 1e-3f);
    for (auto& v : hw) v = (bf16)((rand() % 2001 - 1000) * 3e-5f);
    // X k-blocked [K/32][M][32]; W packed [n_tile][k_step][lane][8]: lane (r16, q) holds W[n_tile*16 + r16][k_step*32 + q*8 .. +7]
    std::vector<bf16> xkb((size_t)M * K), wp((size_t)N * K);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) xkb[((size_t)(k / 32) * M + m) * 32 + k % 32] = hx[(size_t)m * K + k];
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const int nt = n / 16, r = n % 16, ks = k / 32, qq = (k % 32) / 8, e = k % 8;
            wp[(((size_t)nt * KS + ks) * 64 + (qq * 16 + r)) * 8 + e] = hw[(size_t)n * K + k];
        }
    bf16* dx; hipMalloc(&dx, xkb.size() * 2); hipMemcpy(dx, xkb.data(), xkb.size() * 2, hipMemcpyHostToDevice);
    for (int c = 0; c < ncopy; ++c) {
        bf16* dw; float* dp;
        hipMalloc(&dw, wp.size() * 2); hipMemcpy(dw, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
        hipMalloc(&dp, (size_t)S * M * N * 4);
        args[c] = Args{dx, reinterpret_cast<const bf16x8*>(dw), dp};
    }
    std::vector<float> r0, r;
    run<0, 5>("0  shipped loop, 8 waves, 5 stages", args.data(), ncopy, &r0);
    auto same = [&](const std::vector<float>& v) { size_t bad = 0; for (size_t i = 0; i < v.size(); ++i) bad += v[i] != r0[i]; return bad; };
    run<1, 5>("1  4 DMA waves + 8 MFMA waves, 5 stages", args.data(), ncopy, &r); printf("     mismatches vs 0: %zu\n", same(r));
    run<1, 6>("1  4 DMA waves + 8 MFMA waves, 6 stages", args.data(), ncopy, &r); printf("     mismatches vs 0: %zu\n", same(r));
    run<2, 6>("2  as 1, fragment reads one k-step ahead, 6 stages", args.data(), ncopy, &r); printf("     mismatches vs 0: %zu\n", same(r));
    run<3, 5>("3  16 waves of 9x1 tiles, 5 stages", args.data(), ncopy, &r); printf("     mismatches vs 0: %zu\n", same(r));
    run<4, 6>("4  all waves DMA, fragment reads one k-step ahead, 6 stages", args.data(), ncopy, &r); printf("     mismatches vs 0: %zu\n", same(r));
    run<4, 5>("4  same, 5 stages", args.data(), ncopy, &r); printf("     mismatches vs 0: %zu\n", same(r));
    run<0, 5>("0  again", args.data(), ncopy, &r); printf("     mismatches vs 0: %zu\n", same(r));
    run<0, 5, 16>("0  ablation: X pieces re-load one hot panel (W streams)", args.data(), ncopy, &r);
    run<0, 5, 32>("0  ablation: W pieces re-load one hot tile (X streams)", args.data(), ncopy, &r);
    run<0, 5, 48>("0  ablation: both re-load hot data (DMA issue + LDS writes, no new bytes)", args.data(), ncopy, &r);
    run<0, 5, 2>("0  ablation: no DMA in the loop", args.data(), ncopy, &r);
    run<0, 5, 4>("0  ablation: no fragment reads", args.data(), ncopy, &r);
    run<0, 5, 6>("0  ablation: no DMA, no fragment reads (MFMA + barrier)", args.data(), ncopy, &r);
    run<0, 5, 14>("0  ablation: MFMAs only (no DMA, reads, barrier)", args.data(), ncopy, &r);

    // spot check against a host dot product
    double worst = 0;
    for (int t = 0; t < 200; ++t) {
        const int m = rand() % M, n = rand() % N;
        double ref = 0; for (int k = 0; k < K; ++k) ref += (double)(float)hx[(size_t)m * K + k] * (double)(float)hw[(size_t)n * K + k];
        double got = 0; for (int s = 0; s < S; ++s) got += r0[((size_t)s * M + m) * N + n];
        worst = fmax(worst, fabs(got - ref));
    }
    printf("max |sum of slabs - fp64 dot| over 200 samples: %.3e\n", worst);
    return 0;
}
