// In-kernel clock of the persistent tile GEMM under load (MI355X_MICROARCH.md "DVFS give-back" item 6): a diagnostic build of
// gemm_tile_p.hip with s_memtime / s_memrealtime stamps around each workgroup's tile stream, >= 2 s of back-to-back launches on
// random data per shape, clock = d(memtime) / d(memrealtime) x 100 MHz, median over the 256 workgroups of the last launch.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aha-_amd/csrc -o gpurun_out/tile_clock tools/micro/tile_clock.hip ; run it.
#define AHA_CLOCK_STAMP 1
#include "../../aha-_amd/csrc/gemm_tile_p.hip"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

static void fill(std::vector<unsigned short>& v, unsigned seed, float scale) {
    unsigned s = seed;
    for (auto& x : v) {
        s = s * 1664525u + 1013904223u;
        const float f = (((s >> 8) & 0xFFFF) / 32768.0f - 1.0f) * scale;
        unsigned u; __builtin_memcpy(&u, &f, 4);
        x = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
}

int main() {
    const int M = 18432;
    const int shapes[4][2] = {{3072, 1024}, {1024, 1024}, {4096, 1024}, {1024, 4096}};
    const char* names[4] = {"qkv", "out", "fc1", "fc2"};
    for (int sh = 0; sh < 4; ++sh) {
        const int N = shapes[sh][0], K = shapes[sh][1];
        std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K), hb(N);
        fill(ha, 1 + sh, 0.5f); fill(hw, 7 + sh, 0.05f); fill(hb, 3, 0.1f);
        bf16 *A, *W, *C, *B;
        hipMalloc(&A, ha.size() * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&B, N * 2);
        hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(B, hb.data(), N * 2, hipMemcpyHostToDevice);
        GemmTileArgs g{};
        g.A = A; g.lda = K; g.M = M; g.W = W; g.ldw = K; g.N = N; g.K = K; g.C = C; g.ldc = N; g.bias = B; g.act = 0; g.wide_epi = 1;
        if (!aha_gemm_tile_p288_ok(&g)) { printf("shape not eligible\n"); return 1; }
        hipStream_t st; hipStreamCreate(&st);
        for (int i = 0; i < 20; ++i) aha_gemm_tile_p288(&g, st);
        hipStreamSynchronize(st);
        const auto t0 = std::chrono::steady_clock::now();
        long launches = 0;
        double secs = 0;
        while (secs < 2.5) {
            for (int i = 0; i < 500; ++i) aha_gemm_tile_p288(&g, st);
            hipStreamSynchronize(st);
            launches += 500;
            secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        unsigned long long h[4 * 256];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(aha_clock_stamps), sizeof(h));
        std::vector<double> clk;
        for (int b = 0; b < 256; ++b) {
            const double dt = (double)(h[4 * b + 2] - h[4 * b + 0]), dr = (double)(h[4 * b + 3] - h[4 * b + 1]);
            if (dr > 0) clk.push_back(dt / dr * 0.1);          // GHz
        }
        std::sort(clk.begin(), clk.end());
        const double us = secs / launches * 1e6, pf = 2.0 * M * N * K / (us * 1e-6) / 1e15;
        const double ghz = clk[clk.size() / 2];
        const double peak = 1024.0 * 16384.0 / 16.0 * ghz * 1e9 / 1e15;        // 1024 SIMDs x one 16x16x32 MFMA (16,384 flop) per 16 cycles
        printf("%s M=%d N=%d K=%d: %.1f us per launch = %.3f PFLOP/s; in-kernel clock median %.3f GHz (min %.3f max %.3f) -> matrix-pipe peak at that clock %.2f PFLOP/s, achieved %.2f of it\n",
               names[sh], M, N, K, us, pf, ghz, clk.front(), clk.back(), peak, pf / peak);
        hipFree(A); hipFree(W); hipFree(C); hipFree(B); hipStreamDestroy(st);
    }
    return 0;
}
