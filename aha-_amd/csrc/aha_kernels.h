// Argument records and launcher prototypes of the gfx950 kernels (internal, not the C ABI).
#pragma once
#include "aha_common.h"

enum { EPI_PARTIAL = 0, EPI_BF16 = 1, EPI_SWIGLU = 2, EPI_F32_RBF = 3 };

struct GemmWsArgs {
    const bf16* X; int ldx; int M;
    const bf16x8* Wp; int KS;        // k-steps of 32 in the packed weight (padded to a multiple of 8, zero-filled)
    int Kx;                          // valid columns of X (= the weight's real K)
    int n_tiles;                     // 16-row tiles of the packed weight (N_pad/16)
    int S;                           // split-K factor (gridDim.y)
    float* partial; int ldp; long slab_stride;  // EPI_PARTIAL: slab s at partial + s*slab_stride, rows [M][ldp]
    bf16* out; int ldo;              // EPI_BF16 / EPI_SWIGLU
    float* outf; int ldof;           // EPI_F32_RBF
    const bf16* bias;                // EPI_BF16 optional
    int N;                           // valid output columns (for store guards)
    int xkb;                         // mid-M kernel only: 0 = X row-major [M][ldx]; else X is k-blocked [K/32][xkb rows][32] (contiguous k-step panels)
    int okb;                         // mid-M kernel, EPI_SWIGLU only: 0 = out row-major [M][ldo]; else out is k-blocked [N/32][okb rows][32]
};

enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_QUICK_GELU = 3 };   // 3: CLIP, x * sigmoid(1.702 x), three bf16 roundings

struct GemmTileArgs {
    const bf16* A; int lda; int M;
    const bf16* W; int ldw; int N;
    int K;                      // K % 8 == 0
    bf16* C; int ldc;
    const bf16* bias;           // [N] or null
    int act;
    const bf16* residual; int ldr;   // out = bf16(residual + bf16(lin)); may alias C
    const bf16* rowadd; int rowadd_period, ldra;  // out = bf16(bf16(lin) + rowadd[m % period]) (position embedding)
    int wide_epi;               // set by aha_gemm_tile (tuning "tile_epi"): LDS-transposed 16-byte epilogue of the LDS-DMA kernels
    const bf16* Wkb;            // set by aha_gemm_tile_p288 from the registry below: the same weight k-blocked [K/32][N][32], or null
    int akb;                    // persistent tile kernel only: 0 = A row-major [M][lda]; else A is k-blocked [K/32][akb rows][32] (akb == M)
    int ckb;                    // persistent tile kernel only: 0 = C row-major [M][ldc]; else C is written k-blocked [N/32][ckb rows][32] (the next GEMM's A)
};

// Weight prefetch riders (latency path of the vision tower): extra workgroups of a launch that do nothing but read byte ranges -
// the weights of GEMMs a few launches ahead - so that those bytes sit in the Infinity Cache when their GEMM starts.
struct WeightPrefetch {
    const void* p[4]; long bytes[4];      // byte ranges to pull through the caches (16-byte aligned starts; bytes may be 0)
    int n_riders;                         // rider workgroups of 256 threads in the launch (0: none)
};
#ifdef __HIPCC__
static __device__ __forceinline__ void prefetch_rider(const WeightPrefetch& pf, const int rider) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    unsigned acc = 0;
#pragma unroll
    for (int sgm = 0; sgm < 4; ++sgm) {
        const long last = pf.bytes[sgm] - 16;                 // clamp instead of branching: every load of a round is issued before any is waited for
        if (last < 0) continue;
        const char* base = reinterpret_cast<const char*>(pf.p[sgm]);
        const long n_pieces = (pf.bytes[sgm] + 4095) >> 12;   // 4-KiB pieces, one 16-byte load per thread, dealt round-robin to the riders
        for (long pc = rider; pc < n_pieces; pc += (long)pf.n_riders * 16) {
            u32x4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                long off = ((pc + (long)j * pf.n_riders) << 12) + (threadIdx.x & 255) * 16;
                off = off < last ? off : last;
                v[j] = *reinterpret_cast<const u32x4*>(base + off);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc ^= v[j][0];
        }
    }
    asm volatile("" ::"v"(acc));                              // the loads stay; nothing is stored
}
#endif

struct AttnArgs {
    const bf16* q; long q_bs; int ldq;
    const bf16* k; const bf16* v; long kv_bs; int ldk;     // dense mode only
    bf16* out; long o_bs; int ldo;
    float* part_o; float* part_ml;
    int T, G, Hkv, Lk;                                     // Lk: dense mode only
    int split_len, n_splits;
    float scale;
    int layer;                                             // LM mode: cache layer index
    int hd;                                                // real head dim (set by aha_attention)
    // LM, frozen-static steps only: Q from the QKV GEMM's split-K slabs (null -> read a.q)
    const float* q_partial; int q_S; long q_slab_stride; int q_ldp;
    const bf16* q_bias; const bf16* rope_cos; const bf16* rope_sin; int n_pos;
    int okb;                                               // LM mode: 0 = out row-major [B*T][ldo]; else k-blocked [ldo/32][okb rows][32] (consumer: the mid-M o_proj GEMM)
};

struct ResidNormArgs {
    const float* partial; int S; long slab_stride; int ldp;
    const bf16* lin_bf16; int ldl;
    bf16* h; int ldh;
    const bf16* w; bf16* xn; int ldx;
    int H; float eps;
    int xkb;                         // 0: xn row-major [M][ldx]; else k-blocked [H/32][xkb rows][32] (consumer: the mid-M GEMM)
};

// ---- the persistent layer engine (lm_engine.hip): GEMM phases over an LDS-DMA weight ring, hand-offs through counters
struct EngGemm {
    const bf16x8* Wp; int KS;        // packed weight (gemm_ws.hip layout) and its k-steps per n-tile
    const bf16* Xkb;                 // input panels [KS][48][32] (k-blocked, 48-row panels; rows >= M and k-steps >= K/32 are zero)
    int epi;                         // EPI_SWIGLU: gate/up pairs -> out_kb (write-through) + per-slice "pairs done" counters; EPI_PARTIAL: split-K slabs
    bf16* out_kb; int out_cols;      // EPI_SWIGLU: activation panels [out_cols/32][48][32]
    float* partial; int ldp; long slab_stride;   // EPI_PARTIAL: slab s at partial + s*slab_stride, rows [M][ldp]
};
struct EngAssign {                   // what ONE workgroup does in ONE GEMM phase (host-built table, [phase][workgroup])
    int tile0, nt;                   // its n-tiles [tile0, tile0 + nt), nt <= 10
    int ks0, nk;                     // its k-steps [ks0, ks0 + nk), nk even
    int slice;                       // EPI_PARTIAL: the slab it writes
    int ready_idx, ready_target;     // its X panels are published when sync[ready_idx*32] >= ready_target
    int sig0_idx, sig0_cnt, sig1_idx, sig1_cnt;   // EPI_SWIGLU: counters it adds to when its outputs are stored
    int pad_;
};
struct EngArgs {
    ResidNormArgs rn;                // row phase: workgroup r < M reduces row r (rn.xn = 48-row panels, written write-through); rn.H = 0: none
    int rows_idx;                    // counter the row phase adds to (one per finished row)
    int M, grid, n_gemm;
    EngGemm gemm[4];
    const EngAssign* asg;            // [n_gemm][grid]
    unsigned* sync;                  // this launch's counters, one per 128-byte line ([32 * 15]: workgroups finished - the last one zeroes the block again)
    int* err;                        // sticky device error word (heads_kernel poisons the scores when set)
    unsigned long long* stamps;      // diagnostic: [grid][16] wall-clock stamps (null in the product)
    int exp;                         // experiment bits (tuning "engine_exp")
};

// the register-streaming MLP launch (lm_stream.hip): gate/up + SwiGLU -> down_proj, gemm_ws_kernel's work split
struct MlpStreamArgs {
    GemmWsArgs gu, dn;               // gate/up (S = 1, out = act row-major, written through) and down_proj (S slices, partial slabs; X = gu.out)
    int M;
    int gu_blocks, gu_wpb;           // gate/up: workgroups and waves per workgroup that own a pair (237 x 5 at Qwen2-7B)
    int dn_bx, dn_wpb;               // down_proj: workgroups per K slice and tiles (= waves with a tile) per workgroup; the last wave of a workgroup never owns tiles: it polls
    unsigned* sync;                  // [32 * (1 + s)]: gate/up pairs done of down_proj slice s; [32 * 15]: workgroups finished - the last one zeroes the block again
    int* err;                        // sticky device error word
    unsigned long long* stamps;      // diagnostic: [grid][16] wall-clock stamps (null in the product)
};

// frame ingest (ingest.hip): one source frame -> one [3,S,S] canvas
struct IngestArgs {
    const uint8_t* src; int h, w, src_bgr;       // uint8 [h][w][3]; src_bgr: channels arrive B,G,R
    uint8_t* out; int S;                          // uint8 [3][S][S] RGB
    int new_w, new_h, left, top;                  // resized size and its offset inside the canvas
    int need_h, need_v;                           // width / height actually change
    const int* xb; const int* xk; int xks;        // Pillow: bounds [new_w][2], coeffs [new_w][xks];  OpenCV: xb = [new_w][4]
    const int* yb; const int* yk; int yks;        //         bounds [new_h][2], coeffs [new_h][yks];          yb = [new_h][4]
};

struct QkvFinishArgs {
    const float* partial; int S; long slab_stride; int ldp;
    const bf16* qkv_bf16; int ldq_in;            // alternative input (bias already added)
    const bf16* bias;                            // [ (Hq+2Hkv)*D ] (partial path)
    const bf16* rope_cos; const bf16* rope_sin;  // [n_pos][D]
    int n_pos;
    bf16* q_rot; int ldq;
    int Hq, Hkv, D, layer;
};

extern "C" {
int aha_gemm_ws_max_m(int epi);
hipError_t aha_gemm_ws(const GemmWsArgs* a, int epi, int wpb, hipStream_t st);
int aha_gemm_wl_supports(const GemmWsArgs* a, int epi);
hipError_t aha_gemm_wl(const GemmWsArgs* a, int epi, hipStream_t st);
hipError_t aha_pack_w(const bf16* W, int N, int K, int ldw, bf16x8* Wp, int KS, int tile_stride, int tile_off, hipStream_t st);
hipError_t aha_gemm_tile(const GemmTileArgs* g, hipStream_t st);
void aha_gemm_tile_set_dma(int on);
void aha_gemm_tile_set_epi(int on);
void aha_gemm_tile_set_p288(int on);
hipError_t aha_gemm_tile_p288(const GemmTileArgs* g, hipStream_t st);
int aha_gemm_tile_will_use_p288(const GemmTileArgs* g);     // 1: aha_gemm_tile would run this shape on the persistent 288x256 kernel (the only one that takes akb / ckb)
void aha_gemm_tile_kb_register(const void* w_rowmajor, const void* w_kblocked, int N, int K);   // k-blocked twin of a tile-GEMM weight (gemm_tile_p.hip); null twin: forget
void aha_gemm_tile_set_wkb(int on);
hipError_t aha_rows_to_kblocked(const bf16* in, int rows, int K, int ld, bf16* out, hipStream_t st);
int aha_gemm_tile_p288_ok(const GemmTileArgs* g);
float aha_gemm_tile_p288_efficiency(const GemmTileArgs* g, int n_cus);
void aha_gemm_ws_set_kc_small(int v);
void aha_attention_set_dense_tpw(int v);
void aha_attention_set_lm_kernel(int v);
void aha_attention_set_head_kernel(int v);
void aha_attention_set_d96(int v);
extern "C" void aha_gemm_wl_set_balanced(int on);
hipError_t aha_attention(const AttnArgs* a, const StepDesc* sd_dev, int B, int head_dim, hipStream_t st);   // sd_dev: DEVICE pointer or null (dense)
hipError_t aha_lm_engine(const EngArgs* a, hipStream_t st);
hipError_t aha_lm_mlp_stream(const MlpStreamArgs* p, int grid, hipStream_t st);
int aha_lm_mlp_stream_ok(int gu_KS, int dn_KS, int dn_S);
int aha_lm_engine_lds_bytes();
int aha_lm_engine_rows();
int aha_lm_engine_ntmax();
hipError_t aha_rmsnorm(const bf16* x, int ldx, const bf16* w, bf16* out, int ldo, int M, int H, float eps, hipStream_t st);
hipError_t aha_resid_norm(const ResidNormArgs* a, int M, hipStream_t st);
hipError_t aha_qkv_finish(const QkvFinishArgs* a, const StepDesc* sd_dev, int M, hipStream_t st);   // sd_dev: DEVICE pointer
hipError_t aha_qkv_finish_attn_static(const QkvFinishArgs* a, const StepDesc* sd_dev, int M, int T, bf16* out, int ldo, float scale, hipStream_t st);
void aha_sink_rerotate_set_pg(int v);
hipError_t aha_sink_rerotate(const StepDesc* sd_dev, unsigned stream_mask, int n_streams, int nmax, const bf16* rcos, const bf16* rsin, const bf16* cosb, const bf16* sinb, int layers, int Hkv, int D, hipStream_t st);
hipError_t aha_cache_update_layer(const StreamStep* ss, int layer, int Hkv, int D, int T, const bf16* knew, const bf16* vnew, const bf16* rcos, const bf16* rsin, const bf16* cosb, const bf16* sinb, hipStream_t st);
hipError_t aha_repetition_penalty(float* logits, int V, const long* hist, const int* n_hist, float penalty, float* tmp, hipStream_t st);
hipError_t aha_generation_bookkeep(const long* tok, long eos, long* hist, int* n_hist, int cap, int use_hist, long* out_ids, int i, hipStream_t st);
hipError_t aha_heads(const bf16* xn, int ldx, int row_first, int row_step, int count, const bf16* heads_w, int H, float* scores, float* raw, const int* poison, hipStream_t st);
hipError_t aha_im2col_norm(const uint8_t* frames, int N, int S, int P, int Kp, const float* mean3, const float* std3, bf16* out, hipStream_t st);
hipError_t aha_clip_assemble(const bf16* patches, const bf16* cls, const bf16* pos, bf16* x, int n, int Np, int Dv, hipStream_t st);
hipError_t aha_layernorm(const bf16* x, int ldx, const bf16* w, const bf16* b, bf16* out, int ldo, int M, int D, float eps, hipStream_t st);
hipError_t aha_layernorm_pf(const bf16* x, int ldx, const bf16* w, const bf16* b, bf16* out, int ldo, int M, int D, float eps,
                            const WeightPrefetch* pf, hipStream_t st);   // + weight prefetch riders (latency path); pf may be null
hipError_t aha_layernorm_kb(const bf16* x, int ldx, const bf16* w, const bf16* b, bf16* out_kb, int M, int D, float eps, hipStream_t st);   // out k-blocked [D/32][M][32]
hipError_t aha_pool(const bf16* in, bf16* out, int N, int g, int go, int H, int stride, int mode, int frame_rows, hipStream_t st);
hipError_t aha_kblocked_to_rows(const bf16* in, int M, int K, bf16* out, int ldo, hipStream_t st);
hipError_t aha_kblocked_to_rows_n(const bf16* in, int M, int rows, int K, bf16* out, int ldo, hipStream_t st);
hipError_t aha_gather_pool_rows(const bf16* in, bf16* out, int N, int g, int go, int s, int Dv, int frame_rows, hipStream_t st);
hipError_t aha_embed_gather(const long* ids, int n, const bf16* table, int H, int vocab, bf16* out, int ldo, hipStream_t st);
hipError_t aha_argmax(const float* logits, int ld, int V, int rows, long* out, hipStream_t st);
hipError_t aha_ingest_launch(const IngestArgs* a, int method, hipStream_t st);
}
