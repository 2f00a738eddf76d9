// Persistent 288 x 256 tiled MFMA GEMM for the vision tower's throughput shapes (round 3):
//     C[M,N] = epilogue( A[M,K] * W[N,K]^T + bias )          (same contract as gemm_tile.hip: GemmTileArgs)
//
// Why another tile.  profiles/r02_gemm_tile_epilogue.txt: the 256x128 LDS-DMA kernels run the tower's K = 1024 GEMMs at
// 0.60-0.88 PF with a fixed cost of ~44 us inside a 153 us launch, and their LDS is ~88 % busy: a 64x64 wave tile reads
// (64+64) operand rows per 64x64 outputs.  Here:
//   * wave tile 144 x 64 (9 x 4 accumulator tiles, 36 MFMAs per 13 fragment reads per 32-deep k-step): 0.65x the LDS read
//     bytes per flop of the 64x64 wave tile; block tile 288 x 256, 8 waves as 2 (M) x 4 (N), one workgroup per CU.
//   * 288 divides the 576 patches of a ViT-L/14@336 frame: M = 576 n has no ragged m-tile, and at 32 frames the tower's
//     GEMMs are 768 (QKV), 256 (out-proj, fc2) and 1024 (fc1) tiles = exactly 3 / 1 / 4 rounds of the 256 CUs.
//   * PERSISTENT: a workgroup walks its tiles (t = blockIdx + j * gridDim) as ONE continuous stream of k-steps.  The LDS-DMA
//     prefetch (4 stages of 34 KB, three k-steps ahead, counted vmcnt, one raw s_barrier per k-step) runs across tile
//     boundaries, so the first stages of the next tile are already in flight while the finished tile's epilogue runs - no
//     pipeline fill per tile, no launch tail between the tiles of a GEMM.
//   * epilogue through a private 2.3 KB LDS slab per wave, outside the stage ring: one 16-row tile at a time, bf16 tile
//     written row-major, read back as whole 128-byte rows (one cache line per row per wave: the wave's 64 columns) and stored
//     with the residual / row-add operand added on the way.  Same rounding points as every other tile kernel.
//   * a stage = 288 A rows + 256 W rows of 64 B = 34 pieces of 1 KiB.  Every wave issues the same five DMA instructions per
//     k-step - two W pieces, two A pieces and a QUARTER A piece (global_load_lds_dword: 64 lanes x 4 B = four rows) - so
//     the 34 pieces split evenly over 8 waves with no duplicate bytes and one vmcnt constant for all waves.
//   * XCD-aware order: the 32 workgroups an XCD runs in a round (blockIdx equal mod 8) take 8 m-panels x 4 n-panels.
// One output element accumulates its k-steps in the same sequential order as every other variant: bit-identical results
// (tests/test_gpu_parity.py::test_tiled_gemm_variants_are_bit_identical).
#include <mutex>
#include <unordered_map>
#include "aha_kernels.h"
#include "tile_act.h"
#include <unordered_map>
#ifdef AHA_CLOCK_STAMP
__device__ unsigned long long aha_clock_stamps[4 * 256];   // written by the diagnostic build only; no other code reads it
#endif

namespace {
constexpr int PBM = 288, PBN = 256, PBK = 32, PSTAGES = 4;
constexpr int PROWS = PBM + PBN;                   // 544 rows of 64 B per stage
constexpr int PSTAGE = PROWS * PBK;                // elements per stage (34,816 B)
constexpr int PWI = 9, PWJ = 4;                    // 16x16 accumulator tiles per wave: 144 rows x 64 columns
constexpr int PSTG_ST = 64 + 8;                    // epilogue slab row stride (elements): 144 B, keeps ds_read_b128 rows 16-B aligned
constexpr int PSTG = 16 * PSTG_ST;                 // elements per wave slab
constexpr int PBIAS = 128;                         // elements per wave of the bias line (256 B: one global_load_lds_dword)
constexpr int PLDS_BYTES = PSTAGES * PSTAGE * 2 + 8 * PSTG * 2 + 8 * PBIAS * 2;   // 139,264 + 18,432 + 2,048 = 159,744
static_assert(PLDS_BYTES <= 160 * 1024, "stage ring + epilogue slabs exceed the CU's LDS");
constexpr int PPIECES = 5;                         // DMA instructions per wave per k-step
}

// Tile s of the locality-ordered sequence -> (bm, bn): GROUP_M = 8 m-panels x all n-tiles, column-major inside a group.
static __device__ __forceinline__ void p288_tile_coords(int s, int tiles_m, int tiles_n, int* bm, int* bn) {
    const int per_group = 8 * tiles_n, grp = s / per_group, first_m = grp * 8;
    const int gmn = min(tiles_m - first_m, 8), inner = s - grp * per_group;
    *bm = first_m + inner % gmn;
    *bn = inner / gmn;
}

// Epilogue of one finished 144 x 64 wave tile, one 16-row tile at a time through the wave's private LDS slab; zeroes the
// accumulators for the next tile.  ACT < 0: activation / operands decided at run time (rare combinations).
template <int ACT, bool RES, bool ROWADD>
static __device__ __forceinline__ void p288_epilogue(const GemmTileArgs& g, f32x4 (&acc)[PWI][PWJ], bf16* stg, const bf16* bias_lds, const int mw,
                                                     const int nw, const int lane) {
    const int q = lane >> 4, r16 = lane & 15;
#ifdef AHA_ABL_NOEPI
    {   // ablation build only (tools/micro/tile_clock.hip): no epilogue at all - one store keeps the accumulators live
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PWI; ++i)
#pragma unroll
            for (int j = 0; j < PWJ; ++j) { s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3]; acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        if (s == 12345.678f) g.C[lane] = f2bf(s);
        return;
    }
#endif
    const int act = ACT < 0 ? g.act : ACT;
    const bool res = ACT < 0 ? g.residual != nullptr : RES, radd = ACT < 0 ? g.rowadd != nullptr : ROWADD;
    const bf16x4 z4 = {0, 0, 0, 0};
    bf16x4 bv[PWJ];
#pragma unroll
    for (int j = 0; j < PWJ; ++j)      // this wave's 64 bias values were DMA'd into LDS a tile ago (no VMEM load here: an ordinary load's use
        bv[j] = g.bias ? *reinterpret_cast<const bf16x4*>(bias_lds + j * 16 + q * 4) : z4;   // would make hipcc drain the prefetch queue)
#pragma unroll
    for (int i = 0; i < PWI; ++i) {
#pragma unroll
        for (int j = 0; j < PWJ; ++j) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(tile_act(acc[i][j][e] + bf2f(bv[j][e]), act));
            *reinterpret_cast<bf16x4*>(&stg[r16 * PSTG_ST + j * 16 + q * 4]) = o;
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // the slab is private to the wave: its own LDS writes only need to have landed (the compiler's lgkmcnt wait)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * 64 + lane, row = idx >> 3, ch = idx & 7;
            const int m = mw + i * 16 + row, n = nw + ch * 8;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(&stg[row * PSTG_ST + ch * 8]);
            const int mc = min(m, g.M - 1), nc = min(n, g.N - 8);
            if (res) {
                const bf16x8 rv = *reinterpret_cast<const bf16x8*>(g.residual + (long)mc * g.ldr + nc);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = f2bf(rbf(bf2f(rv[e]) + bf2f(v[e])));
            }
            if (radd) {
                const bf16x8 pv = *reinterpret_cast<const bf16x8*>(g.rowadd + (long)(mc % g.rowadd_period) * g.ldra + nc);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = f2bf(rbf(bf2f(v[e]) + bf2f(pv[e])));
            }
#ifdef AHA_ABL_NOSTORE
            if (m < g.M && n < g.N && v[0] == (bf16)12345.0f) *reinterpret_cast<bf16x8*>(g.C + (long)m * g.ldc + n) = v;   // ablation build only
#else
            if (m < g.M && n < g.N) {
                bf16* dst = g.ckb ? g.C + ((long)(n >> 5) * g.ckb + m) * 32 + (n & 31) : g.C + (long)m * g.ldc + n;   // k-blocked: the next GEMM's A
                *reinterpret_cast<bf16x8*>(dst) = v;
            }
#endif
        }
    }
}

__global__ __launch_bounds__(512) void gemm_tile_p288s_kernel(GemmTileArgs g, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char dsm_raw[];
    bf16* lds = reinterpret_cast<bf16*>(dsm_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;
    bf16* stg = lds + PSTAGES * PSTAGE + wave * PSTG;
    bf16* bias_lds = lds + PSTAGES * PSTAGE + 8 * PSTG + wave * PBIAS;

    const int T = tiles_m * tiles_n, G = gridDim.x, bid = blockIdx.x;
    const int nk = g.K / PBK;
    const int my_tiles = bid < T ? (T - bid + G - 1) / G : 0;
    if (my_tiles == 0) return;
#ifdef AHA_CLOCK_STAMP
    // diagnostic build only (tools/micro/tile_clock.hip): shader-clock and 100 MHz reference stamps around the workgroup's whole tile stream
    if (tid == 0) { aha_clock_stamps[4 * bid + 0] = __builtin_amdgcn_s_memtime(); aha_clock_stamps[4 * bid + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    const int total = my_tiles * nk;
    const int full_rounds = T / G;
    // j-th tile of this workgroup -> position in the locality-ordered sequence.  In a full round the workgroups of one XCD
    // (blockIdx equal mod 8 under round-robin placement: speed only) take G/8 consecutive positions; the ragged last round
    // keeps the natural order so that the map stays a bijection onto [0, T).
    auto seq_of = [&](int j) {
        if ((G & 7) == 0 && j < full_rounds) return (j * 8 + (bid & 7)) * (G >> 3) + (bid >> 3);
        return j * G + bid;
    };

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    // ---- DMA side: per-piece source offsets (elements from g.A / g.W, without k) of the tile the prefetch stream is in
    // pieces 0,1: W rows 16 (wave + 8 i) ...; pieces 2,3: A rows 16 (wave + 8 i) ...; piece 4: A rows 256 + 4 wave .. + 3, 4 B per lane
    const int prow = lane >> 2, pslot = lane & 3;                    // 16-B pieces: 16 rows x 4 slots
    const int qrow = lane >> 4, qslot = (lane >> 2) & 3, qbyte = (lane & 3) * 4;   // 4-B quarter piece: 4 rows x 4 slots x 4 dwords
    unsigned poff[PPIECES];
    // W row-major [N][ldw]: a row's 64-byte piece of k-step s sits 32 s elements into the row.  W k-blocked [K/32][N][32] (g.Wkb, the twin
    // registered at load): rows are 32 elements apart and a k-step N * 32, so the 16 rows of a piece are ONE contiguous KiB = 8 whole cache
    // lines instead of 16 half lines - the mid-M LM kernel's operands travel the same way (gemm_wl.hip, -16 % on down_proj).
    const unsigned wrs = g.Wkb ? 32u : (unsigned)g.ldw, wks = g.Wkb ? (unsigned)g.N * 32u : 32u;
    // the same for A when its producer wrote it k-blocked [K/32][M][32] (g.akb = M: the tower's LayerNorm and fc1 epilogue at these shapes)
    const unsigned ars = g.akb ? 32u : (unsigned)g.lda, aks = g.akb ? (unsigned)g.akb * 32u : 32u;
    int d_j = 0, d_k = 0;                                            // prefetch stream position: tile index, k-step
    auto set_tile_offsets = [&](int j) {
        int bm, bn;
        p288_tile_coords(seq_of(min(j, my_tiles - 1)), tiles_m, tiles_n, &bm, &bn);
        const int m0 = bm * PBM, n0 = bn * PBN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int wrow = (wave + 8 * i) * 16 + prow;             // swizzle key: (row >> 2) & 3 with the row index inside its operand
            poff[i] = (unsigned)min(n0 + wrow, g.N - 1) * wrs + ((pslot ^ ((wrow >> 2) & 3)) << 3);
            const int arow = (wave + 8 * i) * 16 + prow;
            poff[2 + i] = (unsigned)min(m0 + arow, g.M - 1) * ars + ((pslot ^ ((arow >> 2) & 3)) << 3);
        }
        const int arow = 256 + 4 * wave + qrow;
        poff[4] = (unsigned)min(m0 + arow, g.M - 1) * ars + ((qslot ^ ((arow >> 2) & 3)) << 3);
    };
    const char* Ab = reinterpret_cast<const char*>(g.A);
    const char* Wb = reinterpret_cast<const char*>(g.Wkb ? g.Wkb : g.W);
    // issue piece i of the prefetch stream's current k-step into stage `st`
    auto dma_piece = [&](int i, int st, int k0) {
        bf16* sb = lds + st * PSTAGE;
        if (i < 2)
            __builtin_amdgcn_global_load_lds((gptr_t)(Wb + ((size_t)poff[i] + (size_t)(k0 >> 5) * wks) * 2), (lptr_t)(sb + (PBM + (wave + 8 * i) * 16) * PBK), 16, 0, 0);
        else if (i < 4)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ab + ((size_t)poff[i] + (size_t)(k0 >> 5) * aks) * 2), (lptr_t)(sb + ((wave + 8 * (i - 2)) * 16) * PBK), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(Ab + ((size_t)poff[4] + (size_t)(k0 >> 5) * aks) * 2 + qbyte), (lptr_t)(sb + (256 + 4 * wave) * PBK), 4, 0, 0);
    };
    // advance the prefetch stream by one k-step (past the last real step it keeps re-reading the last one into dead stages:
    // the per-wave vmcnt arithmetic stays the same to the end)
    auto dma_advance = [&]() {
        if (++d_k == nk) {
            d_k = 0;
            ++d_j;
            set_tile_offsets(d_j);
        }
    };

    f32x4 acc[PWI][PWJ];
#pragma unroll
    for (int i = 0; i < PWI; ++i)
#pragma unroll
        for (int j = 0; j < PWJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The 64 bias values of this wave's columns for tile j -> its LDS line, one 4-byte LDS-DMA per lane (the upper half of the
    // 256 bytes is unused padding).  Issued a whole tile before the epilogue that reads it, so the per-step vmcnt waits have
    // long retired it; always issued (from any valid address when there is no bias) so that the vmcnt arithmetic is fixed.
    auto bias_dma = [&](int j) {
        int bm, bn;
        p288_tile_coords(seq_of(min(j, my_tiles - 1)), tiles_m, tiles_n, &bm, &bn);
        const int col = min(bn * PBN + wn * (16 * PWJ) + lane * 2, g.N - 2);
        const char* src = g.bias ? reinterpret_cast<const char*>(g.bias) + col * 2 : Wb;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)bias_lds, 4, 0, 0);
    };
    // ---- prologue: the first tile's bias line, then k-steps 0, 1, 2 of the stream into stages 0, 1, 2
    bias_dma(0);
    set_tile_offsets(0);
#pragma unroll
    for (int s = 0; s < PSTAGES - 1; ++s) {
        const int k0 = d_k * PBK;
#pragma unroll
        for (int i = 0; i < PPIECES; ++i) dma_piece(i, s, k0);
        dma_advance();
    }

    // Software-pipelined k-steps.
    // A k-step s is split in two halves around ONE barrier:
    //   TOP(s):  read A fragments of row tiles 4..8 from stage s | MFMAs of row tiles 0..3 (their fragments and the W fragments
    //            were read during MID(s-1)) | DMA pieces 0,1 of k-step s+3
    //   MID(s):  lgkmcnt(0) (every LDS read of stage s by this wave has returned) -> vmcnt (this wave's pieces of k-step s+1
    //            have landed) -> s_barrier -> MFMAs of row tiles 4..8 | read the W fragments and the A fragments of row tiles
    //            0..3 of stage s+1 | DMA pieces 2,3,4 of k-step s+3
    // so the matrix pipe never waits for a fragment read at the top of a k-step (both waves of a SIMD used to stand there
    // together after the barrier).  Hazards: k-step s+3 refills the stage of k-step s-1, whose last reads (TOP(s-1)) every
    // wave has completed (lgkmcnt(0)) before the barrier of MID(s-1), and its first piece is issued in TOP(s), after that
    // barrier; stage s+1 is read only after the barrier of MID(s), before which every wave waited for its own pieces of it.
    // vmcnt at MID(s): younger than k-step s+1's pieces are the 5 of s+2 and pieces 0,1 of s+3 -> 7 (+ the epilogue's
    // stores and bias line during the two MIDs after an epilogue, see `relax`).
    bf16x8 af[PWI], wf[PWJ];
    // fragment addresses: row = (multiple of 16) + r16 in both operands, so the swizzle key (row >> 2) & 3 is (r16 >> 2) for every
    // fragment: ONE per-lane byte offset, everything else is a scalar base plus an immediate
    const unsigned frag_off = (unsigned)r16 * (PBK * 2) + ((q ^ (r16 >> 2)) << 4);
    const char* lds_b = reinterpret_cast<const char*>(lds);
    const unsigned lds_u32 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)dsm_raw;   // LDS byte offset of the ring
    auto read_w1 = [&](int j, int st) {
        wf[j] = *reinterpret_cast<const bf16x8*>(lds_b + st * (PSTAGE * 2) + (PBM + wn * (16 * PWJ) + j * 16) * (PBK * 2) + frag_off);
    };
    auto read_a = [&](int i0, int i1, int st) {
#pragma unroll
        for (int i = 0; i < PWI; ++i)
            if (i >= i0 && i < i1)
                af[i] = *reinterpret_cast<const bf16x8*>(lds_b + st * (PSTAGE * 2) + (wm * (16 * PWI) + i * 16) * (PBK * 2) + frag_off);
    };
    constexpr int SPLIT = 4;                                        // row tiles 0..3 in TOP, 4..8 in MID
    int relax = 0, relax_n = 0;
    int c_j = 0, c_k = 0;
    int st_cur = 0, st_new = PSTAGES - 1;

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PSTAGES - 2) * PPIECES) : "memory");       // k-step 0 landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < PWJ; ++j) read_w1(j, 0);
    read_a(0, SPLIT, 0);

    // Both halves walk the 36 accumulator tiles COLUMN-major (j outer): a W fragment is dead after its five MID MFMAs and is
    // reloaded from the next stage into the same registers right there - one W register set, no copies.
    for (int step = 0; step < total; ++step) {
        const int st_next = st_cur == PSTAGES - 1 ? 0 : st_cur + 1;
        const int k0n = d_k * PBK;
        // ---- TOP.  The A fragments of row tiles 4..8 are first used in MID: they are read with inline-asm ds_read_b128, which
        // hipcc's waitcnt insertion does not see (with ordinary loads it put lgkmcnt(0) - the LDS counter is in order and its
        // loop-carried bookkeeping is conservative - in front of the first TOP MFMA, i.e. the matrix pipe waited for reads
        // whose data it needs half a k-step later).  They are issued right AFTER the first MFMA, whose compiler-inserted wait has
        // retired every older LDS read, so no later compiler wait can catch them; MID's explicit lgkmcnt(0) + sched_barrier
        // orders them before their first use (cdna_hip_programming.md 5.4 rule 18, 5.7 item 1).
        acc[0][0] = mfma16(wf[0], af[0], acc[0][0]);
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned va = lds_u32 + st_cur * (PSTAGE * 2) + wm * (16 * PWI * PBK * 2) + frag_off;
#pragma unroll
            for (int i = SPLIT; i < PWI; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[i]) : "v"(va), "n"(i * 16 * PBK * 2) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < PWJ; ++j)
#pragma unroll
            for (int i = 0; i < SPLIT; ++i) {
                const int n = j * SPLIT + i + 1;
                if (n > 1) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
                if (n == 6) dma_piece(0, st_new, k0n);
                if (n == 12) dma_piece(1, st_new, k0n);
            }
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- MID
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (relax > 0) {
            --relax;
            if (relax_n == 19) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPIECES + 2 + 19) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPIECES + 2 + 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPIECES + 2) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_a(0, SPLIT, st_next);                                  // row tiles 0..3 of the next k-step (their registers are free)
#pragma unroll
        for (int j = 0; j < PWJ; ++j) {
#pragma unroll
            for (int i = SPLIT; i < PWI; ++i) {
                acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
                const int n = j * (PWI - SPLIT) + (i - SPLIT) + 1;
                if (n == 4) dma_piece(2, st_new, k0n);
                if (n == 9) dma_piece(3, st_new, k0n);
                if (n == 14) dma_piece(4, st_new, k0n);
            }
            read_w1(j, st_next);                                    // W fragment j of the next k-step
        }
        __builtin_amdgcn_sched_group_barrier(0x100, SPLIT, 0);                  // next A fragments of row tiles 0..3
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // W fragment 0
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // W fragment 1
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // W fragment 2
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // W fragment 3
        __builtin_amdgcn_sched_barrier(0);
        dma_advance();
        st_cur = st_next;
        st_new = st_new == PSTAGES - 1 ? 0 : st_new + 1;

        if (++c_k < nk) continue;
        // ---- the tile is complete: epilogue (the next k-steps' pieces are in flight, the next k-step's first fragments in registers)
        c_k = 0;
        int bm, bn;
        p288_tile_coords(seq_of(c_j), tiles_m, tiles_n, &bm, &bn);
        ++c_j;
        const int mw = bm * PBM + wm * (16 * PWI), nw = bn * PBN + wn * (16 * PWJ);
        // (Measured and dropped: issuing the next k-step's five pieces here, in front of the tile's stores - the stage they refill is
        // already free - so that data issued behind the stores is first needed four k-steps later instead of three: same-job A/B
        // 106.3 -> 105.1 us QKV, 134.9 -> 134.0 fc1, out-proj and fc2 unchanged; same bits.  profiles/r03_tile_clock_ablation.txt.)
        const int combo = g.act * 4 + (g.residual ? 2 : 0) + (g.rowadd ? 1 : 0);
        switch (combo) {
            case ACT_NONE * 4 + 0: p288_epilogue<ACT_NONE, false, false>(g, acc, stg, bias_lds, mw, nw, lane); break;
            case ACT_NONE * 4 + 1: p288_epilogue<ACT_NONE, false, true>(g, acc, stg, bias_lds, mw, nw, lane); break;
            case ACT_NONE * 4 + 2: p288_epilogue<ACT_NONE, true, false>(g, acc, stg, bias_lds, mw, nw, lane); break;
            case ACT_GELU_TANH * 4 + 0: p288_epilogue<ACT_GELU_TANH, false, false>(g, acc, stg, bias_lds, mw, nw, lane); break;
            case ACT_GELU_ERF * 4 + 0: p288_epilogue<ACT_GELU_ERF, false, false>(g, acc, stg, bias_lds, mw, nw, lane); break;
            case ACT_QUICK_GELU * 4 + 0: p288_epilogue<ACT_QUICK_GELU, false, false>(g, acc, stg, bias_lds, mw, nw, lane); break;
            default: p288_epilogue<-1, true, true>(g, acc, stg, bias_lds, mw, nw, lane); break;
        }
        bias_dma(c_j);
        relax = 2;
        relax_n = (bm * PBM + PBM <= g.M && bn * PBN + PBN <= g.N) ? 19 : 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // nothing may still target LDS when the block retires
#ifdef AHA_CLOCK_STAMP
    if (tid == 0) { aha_clock_stamps[4 * bid + 2] = __builtin_amdgcn_s_memtime(); aha_clock_stamps[4 * bid + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

static int g_p288_cus = 0;
// k-blocked twins of tile-GEMM weights, keyed by the row-major pointer the callers pass (registered by the weight loader, forgotten
// when their context goes): the kernel's contract stays "W row-major"; the twin only changes where its DMA finds the same bytes.
struct WkbTwin { const bf16* kb; int N, K; };
static std::unordered_map<const void*, WkbTwin> g_wkb_map;
static std::mutex g_wkb_mu;        // contexts on different threads register / forget / look up twins (the table is per process)
static int g_wkb_on = 1;           // tuning "tile_wkb"
extern "C" void aha_gemm_tile_kb_register(const void* w, const void* kb, int N, int K) {
    std::lock_guard<std::mutex> lk(g_wkb_mu);
    if (kb) g_wkb_map[w] = WkbTwin{reinterpret_cast<const bf16*>(kb), N, K};
    else g_wkb_map.erase(w);
}
extern "C" void aha_gemm_tile_set_wkb(int on) { g_wkb_on = on; }
extern "C" hipError_t aha_gemm_tile_p288(const GemmTileArgs* g_, hipStream_t st) {
    GemmTileArgs gg = *g_;
    gg.Wkb = nullptr;
    if (g_wkb_on && gg.K % 32 == 0) {
        std::lock_guard<std::mutex> lk(g_wkb_mu);
        auto it = g_wkb_map.find(gg.W);
        if (it != g_wkb_map.end() && it->second.N == gg.N && it->second.K == gg.K) gg.Wkb = it->second.kb;      // the whole matrix, as registered
    }
    const GemmTileArgs* g = &gg;
    if ((gg.akb && gg.akb != gg.M) || (gg.ckb && (gg.ckb != gg.M || (gg.N & 31)))) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tile_p288s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PLDS_BYTES);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        g_p288_cus = prop.multiProcessorCount;
        attr_set = true;
    }
    const int tiles_m = ceil_div(g->M, PBM), tiles_n = ceil_div(g->N, PBN), T = tiles_m * tiles_n;
    int grid = T < g_p288_cus ? T : (g_p288_cus & ~7);            // several rounds: whole XCD groups (the locality order needs gridDim % 8 == 0)
    hipLaunchKernelGGL(gemm_tile_p288s_kernel, dim3(grid), dim3(512), PLDS_BYTES, st, *g, tiles_m, tiles_n);
    return hipGetLastError();
}

// what the shape must satisfy (the caller falls back to the other tile kernels otherwise).  K >= 4 k-steps: the next tile's bias
// line is issued by LDS-DMA right behind an epilogue and only the THIRD k-step's strict vmcnt wait guarantees it has landed
// before the next epilogue reads bias_lds (the first two waits are relaxed by the epilogue's younger stores).
extern "C" int aha_gemm_tile_p288_ok(const GemmTileArgs* g) {
    return (g->K % PBK) == 0 && g->K >= 4 * PBK && !(g->N & 7) && !(g->ldc & 7) && (!g->residual || !(g->ldr & 7)) &&
           (!g->rowadd || !(g->ldra & 7)) && !(g->lda & 7) && !(g->ldw & 7);
}
// share of the chip's MFMA time the 288 x 256 decomposition of this shape uses: tile padding x round quantisation
extern "C" float aha_gemm_tile_p288_efficiency(const GemmTileArgs* g, int n_cus) {
    const int tiles_m = ceil_div(g->M, PBM), tiles_n = ceil_div(g->N, PBN), T = tiles_m * tiles_n;
    const int grid = T < n_cus ? T : (n_cus & ~7);
    const int rounds = ceil_div(T, grid);
    return ((float)g->M * (float)g->N) / ((float)rounds * (float)n_cus * (float)(PBM * PBN));
}
