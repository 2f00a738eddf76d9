// api_internal.h -- what the translation units of the C ABI share: the context and stream objects, error plumbing, and the
// declarations of the helpers that cross files (api_ctx.hip: set-up, weights, streams; api_vision.hip: tower + encode entry
// points; api_lm.hip: the LM step and its graph replay; api_ops.hip: operator-level entry points; api_generate.hip).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/aha_amd.h"
#include "aha_kernels.h"

#define AHA_E_INVAL (-22)
#define AHA_E_NOMEM (-12)
#define AHA_E_HIP (-5)
#define AHA_E_RANGE (-34)
#define AHA_E_NOENT (-2)

struct PackedW {
    bf16x8* p = nullptr;
    int n_tiles = 0, KS = 0, N = 0, K = 0;
    double bytes() const { return (double)n_tiles * KS * 1024.0; }
};
struct LayerW {
    PackedW qkv, o, gateup, down;
    bf16 *qkv_bias = nullptr, *ln1 = nullptr, *ln2 = nullptr;
};
struct VLayerW {
    bf16 *ln1w, *ln1b, *wqkv, *bqkv, *wo, *bo, *ln2w, *ln2b, *w1, *b1, *w2, *b2;
};
enum { GK_QKV = 0, GK_O = 1, GK_GATEUP = 2, GK_DOWN = 3, GK_GEMMS = 4,      // the four weight-streaming GEMM kinds (split / wpb knobs)
       GK_ATTN = 4, GK_REROT = 5, GK_COUNT = 6 };                         // timed kinds also cover cache attention and sink re-rotation

struct aha_ctx {
    aha_model_desc d;
    int device = 0;
    std::string err;
    int grid = 0, Np = 0, Tt = 0, Kp = 0, Fp = 0, go = 0, Tf = 0;   // Tt: tower tokens per frame (Np, or Np + 1 with CLIP's class token)
    bf16 *cls_emb = nullptr, *pre_ln_w = nullptr, *pre_ln_b = nullptr, *v_patch = nullptr;   // CLIP only
    float px_mean[3] = {0.5f, 0.5f, 0.5f}, px_std[3] = {0.5f, 0.5f, 0.5f};   // Kp / Fp: patch-vector / MLP width padded to whole 64-wide k-tiles
    bool weights_loaded = false;
    // LM weights
    std::vector<LayerW> L;
    bf16 *final_norm = nullptr, *heads_w = nullptr, *embed = nullptr;
    PackedW lm_head;
    // vision weights
    bf16 *patch_w = nullptr, *patch_b = nullptr, *pos_emb = nullptr;
    std::vector<VLayerW> V;
    bf16 *p0w = nullptr, *p0b = nullptr, *p2w = nullptr, *p2b = nullptr;
    bf16 *post_ln_w = nullptr, *post_ln_b = nullptr;      // optional: only the vision_live.py contract uses the tower's post_layernorm
    // optional: SigLIP attention-pooling head (pooler_output; models/vision_live.py:26-31, frame_token_cls)
    bf16 *hd_probe = nullptr, *hd_in_w = nullptr, *hd_in_b = nullptr, *hd_out_w = nullptr, *hd_out_b = nullptr, *hd_ln_w = nullptr,
         *hd_ln_b = nullptr, *hd_w1 = nullptr, *hd_b1 = nullptr, *hd_w2 = nullptr, *hd_b2 = nullptr, *hd_q = nullptr;
    bool hd_q_ready = false;
    // tables
    bf16 *rope_cos = nullptr, *rope_sin = nullptr;
    int n_pos = 0;
    std::map<std::tuple<int, int, int>, std::pair<bf16*, bf16*>> rerot;
    // HIP-graph replay of frozen TrulyStaticCache steps (tuning "use_graph"): cached executables keyed by the exact step
    // device-resident step descriptor: written to a pinned ring slot and uploaded once per step (1 KB), so kernels take a
    // constant pointer and a captured graph does not bake the per-step stream state in
    StepDesc* sd_pin = nullptr; StepDesc* sd_dev = nullptr; int sd_slot = 0;
    static constexpr int SD_SLOTS = 256;
    hipEvent_t sd_ev[SD_SLOTS] = {nullptr};                 // recorded behind each slot's upload; waited on before the slot is reused
    // the LM / vision workspaces belong to the context: work submitted on a different HIP stream than the previous call's is
    // ordered behind it with an event (correct, merely serialised) instead of racing on them
    hipStream_t last_lm_stream = nullptr, last_vit_stream = nullptr; bool lm_stream_set = false, vit_stream_set = false;
    hipEvent_t lm_done = nullptr, vit_done = nullptr;
    struct GraphEntry {
        int B = 0, T = 0, epoch = 0, n_splits = 0, split_len = 0, flags = 0, seen = 0; hipGraphExec_t exec = nullptr; bool failed = false;
        double wb = 0, fl = 0; int ev_used[8] = {0}; double gk_bytes[8] = {0};      // bookkeeping of the captured step
    };
    std::vector<GraphEntry> graphs;
    std::vector<hipGraphExec_t> retired_graphs;
    hipStream_t cap_stream = nullptr;
    float* graph_scores = nullptr;
    int use_graph = 1, tune_epoch = 0;
    int* bar_err = nullptr;                                  // device error flag the heads kernel checks (poisons the scores with NaN when set)
    int n_cus = 0;
    struct IngestTab { int *xb = nullptr, *xk = nullptr, *yb = nullptr, *yk = nullptr; int xks = 0, yks = 0;
                       hipStream_t up_stream = nullptr; hipEvent_t ready = nullptr; };   // tables are uploaded on up_stream; other streams wait on `ready`
    std::vector<void*> pinned;                               // host staging of coefficient tables (kept: async uploads read them)
    std::map<std::tuple<int, int, int>, IngestTab> ingest_tabs;      // (method, h, w) -> device coefficient tables
    // LM workspaces
    bf16 *h = nullptr, *xn = nullptr, *q_rot = nullptr, *attn_out = nullptr, *act = nullptr;
    float *partial = nullptr, *part_o = nullptr, *part_ml = nullptr, *logits = nullptr, *heads_tmp = nullptr;
    size_t partial_floats = 0, part_o_floats = 0, attn_rows_pad = 0;   // attn_rows_pad: partial rows the attention buffers hold per (kv head, 1/16 of them)
    int last_B = 0, last_T = 0;
    // ViT workspaces
    bf16 *v_a0 = nullptr, *v_x = nullptr, *v_h = nullptr, *v_qkv = nullptr, *v_attn = nullptr, *v_f = nullptr,
         *v_p1 = nullptr, *v_p2 = nullptr;
    // tuning
    int split[GK_GEMMS] = {0, 0, 0, 0};
    // waves per workgroup per GEMM kind (measured: tools/tune_lm.py).  gate/up: 1184 wave-tasks as 237 five-wave workgroups
    // (one per CU on 237 CUs) instead of 148 eight-wave ones: each CU then ingests less than its ~43 GB/s ceiling.
    int wpb[GK_GEMMS] = {4, 4, 5, 8};
    int attn_split_len = 0;
    int time_gemm = 0;
    int act_kb = 3;                                         // tuning: k-blocked operands between the mid-M kernels: >= 1 the SwiGLU activation (down_proj), >= 2 gate/up's normed input, >= 3 the QKV and o_proj inputs
    int attn_kb_rows = 0;                                   // rows of the k-blocked attention output the last step left in c->attn_out (0: row-major)
    int act_kb_rows = 0;                                    // rows of the k-blocked activation the last step left in c->act (0: row-major)
    int dev_xkb = 0;                                        // experiment: aha_linear_forward reads X k-blocked ([K/32][ldx rows][32])
    int use_wl = 1;                                         // tuning: mid-M GEMM kernel (gemm_wl.hip) for row chunks above 128 (0: gemm_ws everywhere)
    int layer_first = 0, layer_count = 0;                   // tuning: run only decoder layers [first, first+count) (0 = all); parity taps
    // generation scratch (aha_generate_greedy): next-token id, embedding row, penalty temporaries, device history count, host poll slot
    long* gen_tok = nullptr; bf16* gen_emb = nullptr; float* gen_tmp = nullptr; int* gen_nhist = nullptr; long* gen_out = nullptr;
    long* gen_pin = nullptr; hipEvent_t gen_ev = nullptr; int gen_cap = 0;
    // operator-level attention (aha_attention_forward): its own descriptor slot ring is the step's (sd_pin / sd_dev)
    int pool_subset = 1;                 // projector only on the patch rows bilinear pooling samples (tuning "pool_subset"; bit-identical)
    int vit_prefetch_rows = 2400;        // tuning "vit_prefetch": tower encodes of up to this many token rows (4 frames of 576) prefetch weights from their LayerNorm launches
    int vit_akb = 1;                     // tuning "vit_akb": k-blocked activations between the tower's LayerNorm / fc1 and the persistent tile GEMMs
    int vit_riders = 256;                // tuning "vit_riders": rider workgroups per prefetching launch
    int vit_alias = 0;                   // diagnostic: encoder layer l runs layer l % vit_alias's weights (0 = its own): a tower whose weights stay cache-resident
    int static_attn = 1;                 // frozen-static steps with a prefix <= 64 keys: qkv_finish + attention in one launch (tuning "static_attn")
    int fuse_static = 0;                 // frozen-static steps: skip K/V projection + Q built inside attention (tuning key
                                         // "fuse_static"; bit-identical, measured 0 % gain: the chain is latency-bound)
    // single-launch forms of a layer's MLP half on single-stream steps (M <= 48 rows); tuning "engine": 0 off (default), 1 = lm_engine.hip
    // (LDS-DMA loader ring: resid_norm + gate/up + SwiGLU + down_proj; measured 5 % slower on the whole step), 2 = lm_stream.hip (register
    // streaming: gate/up + SwiGLU -> down_proj; measures what the two launches measure).  Both bit-identical (tests/test_gpu_layers.py);
    // experiments, not the product path (profiles/r06_engine_mlp_stamps.txt).
    int engine = 0;
    bf16 *eng_xn = nullptr, *eng_act = nullptr;             // hand-off panels [K/32][48][32]
    unsigned* eng_sync = nullptr;                           // [layers][16 counters, one per 128-byte line], zero between launches (the kernels reset them on the way out)
    EngAssign* eng_asg = nullptr;                           // device copy of eng_host
    std::vector<EngAssign> eng_host;                        // [phase][workgroup]
    int eng_epoch = -1, eng_grid = 0, eng_G = 0; bool eng_ok = false;
    unsigned long long* eng_stamps = nullptr;               // diagnostic (aha_lm_engine_stamps)
    int eng_exp = 0;                                        // tuning "engine_exp": experiment bits handed to the kernel
    int eng_ran = 0;                                        // rows of the last step if the engine produced c->eng_act (parity tap 4), else 0
    // accounting of the last step
    double last_weight_bytes = 0, last_kv_bytes = 0, last_flops = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[GK_COUNT];
    int ev_used[GK_COUNT] = {0};
    double gk_bytes[GK_COUNT] = {0};
    std::vector<void*> allocs;
    std::vector<const void*> kb_keys;                        // row-major tile-GEMM weights that have a k-blocked twin registered (forgotten on destroy)
};

struct aha_stream {
    aha_ctx* ctx;
    int device = 0;                      // aha_stream_destroy must not dereference ctx (it may already be gone)
    int policy, W, sink, cap;
    bf16 *k = nullptr, *v = nullptr;
    int len = 0, head = 0, seen = 0;
    bool poisoned = false;               // a step failed after destructive device work was enqueued: refuse steps until aha_stream_reset
    int semantics = AHA_ATTN_TRAILING;
    int pos_off = 0;                     // added to the RoPE position of new token 0 (aha_stream_set_position_offset)
    // operator-level aha_cache_update: the step planned by layer 0's call, followed by the other layers of that step
    StreamStep op_ss; int op_T = 0, op_next_layer = 0; bool op_valid = false;
};

static inline int fail(aha_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
#define HIPCHK(c, expr)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((c), AHA_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
static inline int dalloc(aha_ctx* c, T** out, size_t count) {
    void* p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) return fail(c, AHA_E_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    c->allocs.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return 0;
}

// Order work submitted on `st` behind everything the previous call of the same family (LM / vision) submitted on another
// stream: the workspaces belong to the context, so two streams must not run on them concurrently.
static inline int order_behind(aha_ctx* c, hipStream_t st, hipStream_t* last, bool* set, hipEvent_t* ev) {
    if (*set && *last != st) {
        if (!*ev) HIPCHK(c, hipEventCreateWithFlags(ev, hipEventDisableTiming));
        if (hipEventRecord(*ev, *last) != hipSuccess || hipStreamWaitEvent(st, *ev, 0) != hipSuccess) {
            (void)hipGetLastError();                     // the previous stream is gone: its work is ordered by a full sync
            HIPCHK(c, hipDeviceSynchronize());
        }
    }
    *last = st;
    *set = true;
    return 0;
}
#define ORDER_LM(c, st) do { if (int rc_ = order_behind((c), (st), &(c)->last_lm_stream, &(c)->lm_stream_set, &(c)->lm_done)) return rc_; } while (0)
#define ORDER_VIT(c, st) do { if (int rc_ = order_behind((c), (st), &(c)->last_vit_stream, &(c)->vit_stream_set, &(c)->vit_done)) return rc_; } while (0)

// ---- helpers defined in one file and used by others
int plan_stream(aha_ctx* c, aha_stream* s, int T, StreamStep* o);                                  // api_ctx.hip
hipError_t tile_gemm(const bf16* A, int lda, int M, const bf16* W, int ldw, int N, int K, bf16* C, int ldc, const bf16* bias, int act,
                     const bf16* residual, int ldr, const bf16* rowadd, int period, int ldra, hipStream_t st);      // api_vision.hip
int vit_layers(aha_ctx* c, int n, int l0, int l1, hipStream_t st);                                 // api_vision.hip
void attn_geometry(const aha_ctx* c, int B, int T, int max_lk, int split_override, int* split_len_out, int* n_splits_out);   // api_lm.hip
int pick_split(aha_ctx* c, int kind, const PackedW& w, int M, int nt_per_wave);                    // api_lm.hip
int ws_gemm(aha_ctx* c, int kind, const bf16* X, int ldx, int M, const PackedW& w, int epi, int S, float* partial, int ldp, bf16* out, int ldo,
            float* outf, int ldof, hipStream_t st, int kb = 0);                                    // api_lm.hip
bool ws_all_wl(const aha_ctx* c, int epi, int M, int K);                                           // api_lm.hip
int ws_row_chunk(const aha_ctx* c, int epi, int M, int K);                                         // api_lm.hip
hipError_t ws_or_wl(const aha_ctx* c, const GemmWsArgs* a, int epi, int wpb, hipStream_t st);      // api_lm.hip
GemmWsArgs ws_args(const bf16* X, int ldx, int M, int m0, int mrows, const PackedW& w, int S, float* partial, int ldp, bf16* out, int ldo,
                   float* outf, int ldof);                                                        // api_lm.hip
int alloc_packed(aha_ctx* c, PackedW* w, int n_tiles, int K);                                      // api_ctx.hip
