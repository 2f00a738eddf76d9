// How do ds_read_b128 fragment reads and MFMAs share a SIMD on gfx950?  (round 4: every LDS-fed 16x16x32 loop in this repo -
// gemm_wl, gemm_wl_bal18, gemm_tile_p288s - sits at 0.4-0.6 of its MFMA-issue bound, and a software-pipelined gemm_wl measured
// no better than the burst form.)  One workgroup per CU, W waves, each iteration = NR fragment reads (1 KiB each, conflict-free,
// lane * 16) + NM MFMAs on independent accumulators, fragments consumed by the MFMAs:
//   mode 0  burst:      reads -> lgkmcnt(0) -> MFMAs                       (gemm_wl's round-2 loop)
//   mode 1  pipelined:  lgkmcnt(0) -> reads of the NEXT iteration -> MFMAs (round-4 experiment)
//   mode 2  interleaved: next iteration's reads spread evenly between this iteration's MFMAs
//   mode 3  MFMAs only (reads once, outside the loop)
//   mode 4  reads only
// MF = 0: v_mfma_f32_16x16x32_bf16 (16 cycles), MF = 1: v_mfma_f32_32x32x16_bf16 (32 cycles, NM counts these).
// Prints cycles per iteration per wave (s_memtime, median over workgroups) and the ideal NM * 16 (or 32) * waves-per-SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/lds_mfma_probe tools/micro/lds_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ unsigned long long g_stamp[256 * 4];

template <int NR, int NM, int MODE, int MF>
__global__ __launch_bounds__(512) void probe(int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 40 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)((i * 2654435761u) >> 20) * 1e-6f;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 2048 + lane * 16;
    bf16x8 fa[NR], fb[NR];
    constexpr int NACC = MF ? (NM < 8 ? NM : 8) : (NM < 36 ? NM : 36);
    f32x4 acc4[MF ? 1 : NACC];
    f32x16 acc16[MF ? NACC : 1];
    for (auto& a : acc4) a = (f32x4){0, 0, 0, 0};
    for (auto& a : acc16) for (int e = 0; e < 16; ++e) a[e] = 0.f;
    auto rd = [&](bf16x8& f, int i) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(base), "n"((i % 16) * 1024) : "memory"); };
    auto mm = [&](bf16x8 (&f)[NR], int j) {
        if constexpr (MF == 0) acc4[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[j % NR], f[(j + 1) % NR], acc4[j % NACC], 0, 0, 0);
        else acc16[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[j % NR], f[(j + 1) % NR], acc16[j % NACC], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < NR; ++i) { rd(fa[i], i); rd(fb[i], i + 3); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto step = [&](bf16x8 (&cur)[NR], bf16x8 (&nxt)[NR]) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) rd(cur[i], i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NM; ++j) mm(cur, j);
        } else if constexpr (MODE == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NR; ++i) rd(nxt[i], i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NM; ++j) mm(cur, j);
        } else if constexpr (MODE == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            int r = 0;
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                mm(cur, j);
                if (r < NR && (j + 1) * NR >= (r + 1) * NM) {
                    __builtin_amdgcn_sched_barrier(0);
                    rd(nxt[r], r);
                    ++r;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int j = 0; j < NM; ++j) mm(cur, j);
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) rd(cur[i], i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int it = 0; it < iters; it += 2) {
        step(fa, fb);
        step(fb, fa);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { g_stamp[blockIdx.x * 4] = t0; g_stamp[blockIdx.x * 4 + 1] = t1; g_stamp[blockIdx.x * 4 + 2] = r0; g_stamp[blockIdx.x * 4 + 3] = r1; }
    float s = 0.f;
    for (auto& a : acc4) s += a[0] + a[1] + a[2] + a[3];
    for (auto& a : acc16) for (int e = 0; e < 16; ++e) s += a[e];
    for (int i = 0; i < NR; ++i) s += (float)fa[i][0] + (float)fb[i][0];
    if (s == 123.456f) sink[tid] = s;
}

template <int NR, int NM, int MODE, int MF>
static void run(int waves, const char* what) {
    float* sink;
    hipMalloc(&sink, 4096);
    auto k = probe<NR, NM, MODE, MF>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    // sustained: one long launch (tens of ms: power management has settled), host-timed; the in-kernel stamps give the clock it ran at
    const int big = 100000;
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 100 * 1024, 0, 2000, sink);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 100 * 1024, 0, big, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h));
    std::vector<double> ghz;
    for (int b = 0; b < 256; ++b) ghz.push_back((double)(h[4 * b + 1] - h[4 * b]) / ((double)(h[4 * b + 3] - h[4 * b + 2]) * 10.0));
    std::sort(ghz.begin(), ghz.end());
    const double ns = ms * 1e6 / big, clk = ghz[128], cyc = ns * clk;
    const double mf_cyc = NM * (MF ? 32.0 : 16.0) * (waves / 4.0);
    const double pf = MODE == 4 ? 0.0 : 256.0 * waves * NM * (MF ? 32768.0 : 16384.0) / ns * 1e9 / 1e15;
    printf("%-24s NR=%2d NM=%2d %s waves/SIMD=%d: %7.1f ns/step host-timed, in-kernel clock %.2f GHz -> %6.0f cycles/step; MFMA issue bound %5.0f cycles -> pipe busy %.2f; %.2f PFLOP/s chip\n",
           what, NR, NM, MF ? "32x32x16" : "16x16x32", waves / 4, ns, clk, cyc, mf_cyc, MODE == 4 ? 0.0 : mf_cyc / cyc, pf);
    hipFree(sink);
}

int main() {
    run<11, 18, 3, 0>(4, "mfma only");
    run<10, 9, 3, 1>(4, "mfma only");
    // gemm_wl trio: 11 reads, 18 MFMAs; p288s: 13 reads, 36 MFMAs
    run<11, 18, 3, 0>(8, "mfma only");
    run<11, 18, 4, 0>(8, "reads only");
    run<11, 18, 0, 0>(8, "burst");
    run<11, 18, 1, 0>(8, "pipelined (reads first)");
    run<11, 18, 2, 0>(8, "interleaved");
    run<11, 18, 0, 0>(4, "burst");
    run<11, 18, 1, 0>(4, "pipelined (reads first)");
    run<11, 18, 2, 0>(4, "interleaved");
    run<13, 36, 3, 0>(8, "mfma only");
    run<13, 36, 0, 0>(8, "burst");
    run<13, 36, 1, 0>(8, "pipelined (reads first)");
    run<13, 36, 2, 0>(8, "interleaved");
    run<13, 36, 2, 0>(4, "interleaved");
    // 32x32x16: wave tile 288 x 32 -> 10 reads per 9 MFMAs; 160 x 64 -> 7 reads per 10
    run<10, 9, 3, 1>(8, "mfma only");
    run<10, 9, 0, 1>(8, "burst");
    run<10, 9, 1, 1>(8, "pipelined (reads first)");
    run<10, 9, 2, 1>(8, "interleaved");
    run<7, 10, 2, 1>(8, "interleaved");
    run<7, 10, 2, 1>(4, "interleaved");
    return 0;
}
