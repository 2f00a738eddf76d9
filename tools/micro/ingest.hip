// Per-CU operand ingest on MI355X: the same L2-resident panel stream (k-blocked X of a mid-M GEMM: [kt][288 rows][32] bf16, 18 KiB per
// k-step, every CU reads all of it) pulled (a) by global_load_dwordx4 into registers, (b) by global_load_lds_dwordx4 into an LDS ring.
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ingest tools/micro/ingest.hip ; run: gpurun_out/ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int PANEL = 18 * 1024;    // bytes per k-step
constexpr int NW = 8;

__global__ __launch_bounds__(512) void ingest_reg(const char* __restrict__ x, int nkt, int reps, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
#pragma unroll 4
        for (int kt = 0; kt < nkt; ++kt) {
            const char* p = x + (long)kt * PANEL + lane * 16;
            // wave w takes pieces w, w + 8, (w + 16 for w < 2): 18 pieces per k-step
            u32x4 a = *reinterpret_cast<const u32x4*>(p + wave * 1024);
            u32x4 b = *reinterpret_cast<const u32x4*>(p + (wave + 8) * 1024);
            acc ^= a; acc ^= b;
            if (wave < 2) { u32x4 c = *reinterpret_cast<const u32x4*>(p + (wave + 16) * 1024); acc ^= c; }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1;
}

template <int STAGES>
__global__ __launch_bounds__(512) void ingest_dma(const char* __restrict__ x, int nkt, int reps, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int P = 3;            // pieces per wave per k-step (waves >= 2 re-load piece 17: equal vmcnt counts)
    auto dma = [&](int kt, int st) {
        const char* p = x + (long)kt * PANEL + lane * 16;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int pc = min(wave + i * NW, 17);
            __builtin_amdgcn_global_load_lds((gptr_t)(p + pc * 1024), (lptr_t)(smem + st * PANEL + pc * 1024), 16, 0, 0);
        }
    };
    for (int r = 0; r < reps; ++r) {
        for (int s = 0; s < STAGES - 1; ++s) dma(s, s);
        int st = STAGES - 1;
        for (int kt = 0; kt < nkt; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * P) : "memory");
            __builtin_amdgcn_s_barrier();
            dma(min(kt + STAGES - 1, nkt - 1), st);
            st = st == STAGES - 1 ? 0 : st + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (smem[threadIdx.x] == 0x7f && smem[threadIdx.x + 512] == 0x7e) out[blockIdx.x] = 1;
}

int main() {
    const int nkt = 592, reps = 8;                       // K = 18944: 10.9 MB of panels, L2 / MALL resident
    char* x; unsigned* out;
    hipMalloc(&x, (size_t)nkt * PANEL); hipMalloc(&out, 4096);
    hipMemset(x, 1, (size_t)nkt * PANEL);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch, int wgs) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)nkt * PANEL * reps;
        printf("%-34s %4d workgroups: %8.1f us  -> %6.1f GB/s per CU, %6.2f TB/s chip\n", name, wgs, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * wgs / (ms * 1e-3) / 1e12);
    };
    for (int wgs : {1, 64, 256}) {
        timeit("global_load_dwordx4 -> VGPR", [&] { hipLaunchKernelGGL(ingest_reg, dim3(wgs), dim3(512), 0, 0, x, nkt, reps, out); }, wgs);
        hipFuncSetAttribute((const void*)ingest_dma<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * PANEL);
        timeit("global_load_lds_dwordx4, 5 stages", [&] { hipLaunchKernelGGL(ingest_dma<5>, dim3(wgs), dim3(512), 5 * PANEL, 0, x, nkt, reps, out); }, wgs);
        hipFuncSetAttribute((const void*)ingest_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PANEL);
        timeit("global_load_lds_dwordx4, 8 stages", [&] { hipLaunchKernelGGL(ingest_dma<8>, dim3(wgs), dim3(512), 8 * PANEL, 0, x, nkt, reps, out); }, wgs);
    }
    return 0;
}
