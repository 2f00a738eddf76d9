// C-ABI of the streaming path (include/aha_amd.h): context, weight repacking, per-stream KV
// state machines (the reference's cache policies as ring bookkeeping), and the orchestration of
// the gfx950 kernels for aha_vit_encode / aha_lm_step.
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
    // fused MLP block (lm_fused.hip): device arrival counter + its host-side base, error flag, switch
    unsigned long long* bar_ctr = nullptr; unsigned long long bar_base = 0; int* bar_err = nullptr; int fuse_mlp = 0, n_cus = 0;
    struct IngestTab { int *xb = nullptr, *xk = nullptr, *yb = nullptr, *yk = nullptr; int xks = 0, yks = 0;
                       hipStream_t up_stream = nullptr; hipEvent_t ready = nullptr; };   // tables are uploaded on up_stream; other streams wait on `ready`
    std::vector<void*> pinned;                               // host staging of coefficient tables (kept: async uploads read them)
    std::map<std::tuple<int, int, int>, IngestTab> ingest_tabs;      // (method, h, w) -> device coefficient tables
    // LM workspaces
    bf16 *h = nullptr, *xn = nullptr, *q_rot = nullptr, *attn_out = nullptr, *act = nullptr;
    float *partial = nullptr, *part_o = nullptr, *part_ml = nullptr, *logits = nullptr, *heads_tmp = nullptr;
    size_t partial_floats = 0, part_o_floats = 0;
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
    int act_kb = 2;                                         // tuning: k-blocked SwiGLU activation between the mid-M gate/up and down GEMMs
    int act_kb_rows = 0;                                    // rows of the k-blocked activation the last step left in c->act (0: row-major)
    int dev_xkb = 0;                                        // experiment: aha_linear_forward reads X k-blocked ([K/32][ldx rows][32])
    int use_wl = 1;                                         // tuning: mid-M GEMM kernel (gemm_wl.hip) for row chunks above 128 (0: gemm_ws everywhere)
    int layer_first = 0, layer_count = 0;                   // tuning: run only decoder layers [first, first+count) (0 = all); parity taps
    // generation scratch (aha_generate_greedy): next-token id, embedding row, penalty temporaries, device history count, host poll slot
    long* gen_tok = nullptr; bf16* gen_emb = nullptr; float* gen_tmp = nullptr; int* gen_nhist = nullptr; long* gen_out = nullptr;
    long* gen_pin = nullptr; hipEvent_t gen_ev = nullptr; int gen_cap = 0;
    // operator-level attention (aha_attention_forward): its own descriptor slot ring is the step's (sd_pin / sd_dev)
    int pool_subset = 1;                 // projector only on the patch rows bilinear pooling samples (tuning "pool_subset"; bit-identical)
    int static_attn = 1;                 // frozen-static steps with a prefix <= 64 keys: qkv_finish + attention in one launch (tuning "static_attn")
    int fuse_static = 0;                 // frozen-static steps: skip K/V projection + Q built inside attention (tuning key
                                         // "fuse_static"; bit-identical, measured 0 % gain: the chain is latency-bound)
    // accounting of the last step
    double last_weight_bytes = 0, last_kv_bytes = 0, last_flops = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[GK_COUNT];
    int ev_used[GK_COUNT] = {0};
    double gk_bytes[GK_COUNT] = {0};
    std::vector<void*> allocs;
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

static int fail(aha_ctx* c, int code, const std::string& msg) {
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
static int dalloc(aha_ctx* c, T** out, size_t count) {
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
static int order_behind(aha_ctx* c, hipStream_t st, hipStream_t* last, bool* set, hipEvent_t* ev) {
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

// --------------------------------------------------------------------------------------------
extern "C" const char* aha_version(void) { return "aha_amd 0.1 (gfx950)"; }

extern "C" const char* aha_last_error(aha_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" int aha_ctx_create(const aha_model_desc* d, int device, aha_ctx** out) {
    if (!d || !out) return AHA_E_INVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return AHA_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return AHA_E_HIP;
    aha_ctx* c = new aha_ctx();
    c->d = *d;
    c->device = device;
    *out = c;
    const int Dh = d->head_dim;
    if (Dh != 64 && Dh != 128) return fail(c, AHA_E_INVAL, "LM head_dim must be 64 or 128");
    if (d->hidden % 32 || d->inter % 32 || (d->heads * Dh) % 32) return fail(c, AHA_E_INVAL, "LM dims must be multiples of 32");
    if (d->heads % d->kv_heads) return fail(c, AHA_E_INVAL, "heads %% kv_heads != 0");
    const int vhd = d->v_hidden / d->v_heads;
    if (vhd < 8 || vhd > 128 || (vhd & 7) || vhd * d->v_heads != d->v_hidden) return fail(c, AHA_E_INVAL, "vision head_dim must be a multiple of 8, <= 128");
    if (d->v_hidden % 8 || d->v_inter % 8 || d->v_hidden > 4096) return fail(c, AHA_E_INVAL, "vision dims must be multiples of 8, width <= 4096");
    if (d->hidden > 8192) return fail(c, AHA_E_INVAL, "hidden > 8192 unsupported");
    c->grid = d->image_size / d->patch_size;
    c->Np = c->grid * c->grid;
    if (d->v_kind != AHA_VISION_SIGLIP && d->v_kind != AHA_VISION_CLIP) return fail(c, AHA_E_INVAL, "unknown v_kind");
    c->Tt = c->Np + (d->v_kind == AHA_VISION_CLIP ? 1 : 0);
    if (d->v_kind == AHA_VISION_CLIP) {                  // transformers.utils.constants OPENAI_CLIP_MEAN / OPENAI_CLIP_STD
        const float m[3] = {0.48145466f, 0.4578275f, 0.40821073f}, sd[3] = {0.26862954f, 0.26130258f, 0.27577711f};
        for (int i = 0; i < 3; ++i) { c->px_mean[i] = m[i]; c->px_std[i] = sd[i]; }
    }
    // K of the patch embedding (3*P*P = 588) and of fc2 (so400m: 4304) padded with zeros to whole k-tiles so that
    // every tower GEMM is eligible for the LDS-DMA kernels (gemm_tile.hip); zero columns add exact zeros.
    c->Kp = round_up(3 * d->patch_size * d->patch_size, 64);
    c->Fp = round_up(d->v_inter, 64);
    c->go = d->pool_mode == 0 ? ceil_div(c->grid, d->pool_stride) : c->grid / d->pool_stride;
    c->Tf = c->go * c->go;

    // ---- LM workspaces
    const size_t M = (size_t)d->max_step_tokens, H = d->hidden, QD = (size_t)d->heads * Dh, I = d->inter;
    const int G = d->heads / d->kv_heads;
    int rc;
    if ((rc = dalloc(c, &c->h, M * H))) return rc;
    if ((rc = dalloc(c, &c->xn, M * H))) return rc;
    if ((rc = dalloc(c, &c->q_rot, M * QD))) return rc;
    if ((rc = dalloc(c, &c->attn_out, M * QD))) return rc;
    if ((rc = dalloc(c, &c->act, M * I))) return rc;
    const size_t nqkv = round_up((d->heads + 2 * d->kv_heads) * Dh, 16);
    c->partial_floats = 16 * M * (nqkv > H ? nqkv : H);       // up to 16 split-K slabs
    if ((rc = dalloc(c, &c->partial, c->partial_floats))) return rc;
    const size_t rows_pad = (size_t)G * M + 16 * AHA_MAX_B;
    c->part_o_floats = (size_t)d->kv_heads * 16 * rows_pad * Dh;   // up to 16 key splits
    if ((rc = dalloc(c, &c->part_o, c->part_o_floats))) return rc;
    if ((rc = dalloc(c, &c->part_ml, (size_t)d->kv_heads * 16 * rows_pad * 2))) return rc;
    if ((rc = dalloc(c, &c->logits, (size_t)AHA_MAX_B * d->vocab))) return rc;
    if ((rc = dalloc(c, &c->heads_tmp, M * 4))) return rc;

    // ---- device-resident step descriptor
    if ((rc = dalloc(c, &c->sd_dev, 1))) return rc;
    if (hipHostMalloc((void**)&c->sd_pin, sizeof(StepDesc) * aha_ctx::SD_SLOTS, hipHostMallocDefault) != hipSuccess)
        return fail(c, AHA_E_NOMEM, "hipHostMalloc failed");
    // ---- graph replay state
    if ((rc = dalloc(c, &c->graph_scores, (size_t)AHA_MAX_B * 3))) return rc;
    if (hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking) != hipSuccess) return fail(c, AHA_E_HIP, "hipStreamCreate failed");
    // ---- fused-kernel barrier state
    if ((rc = dalloc(c, &c->bar_ctr, 16 * 17)) || (rc = dalloc(c, &c->bar_err, 1))) return rc;
    if (hipMemset(c->bar_ctr, 0, 16 * 17 * sizeof(unsigned long long)) != hipSuccess || hipMemset(c->bar_err, 0, sizeof(int)) != hipSuccess)
        return fail(c, AHA_E_NOMEM, "hipMemset failed");
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(c, AHA_E_HIP, "hipGetDeviceProperties failed");
        c->n_cus = prop.multiProcessorCount;
    }

    // ---- ViT workspaces
    const size_t R = (size_t)d->max_vit_frames * c->Tt, Dv = d->v_hidden;
    if ((rc = dalloc(c, &c->v_a0, R * c->Kp))) return rc;
    if (d->v_kind == AHA_VISION_CLIP && (rc = dalloc(c, &c->v_patch, R * Dv))) return rc;      // patch embeddings before the class token is appended
    if ((rc = dalloc(c, &c->v_x, R * Dv))) return rc;
    if ((rc = dalloc(c, &c->v_h, R * Dv))) return rc;
    if ((rc = dalloc(c, &c->v_qkv, R * 3 * Dv))) return rc;
    if ((rc = dalloc(c, &c->v_attn, R * Dv))) return rc;
    if ((rc = dalloc(c, &c->v_f, R * c->Fp))) return rc;
    if (hipMemset(c->v_f, 0, R * c->Fp * sizeof(bf16)) != hipSuccess)       // pad columns stay zero: fc1 writes n < v_inter only
        return fail(c, AHA_E_NOMEM, "hipMemset failed");
    if ((rc = dalloc(c, &c->v_p1, R * H))) return rc;
    if ((rc = dalloc(c, &c->v_p2, R * H))) return rc;
    return 0;
}

extern "C" void aha_ctx_destroy(aha_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (auto& g : c->graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    for (auto e : c->retired_graphs) hipGraphExecDestroy(e);
    if (c->cap_stream) hipStreamDestroy(c->cap_stream);
    if (c->sd_pin) hipHostFree(c->sd_pin);
    for (auto e : c->sd_ev) if (e) hipEventDestroy(e);
    if (c->lm_done) hipEventDestroy(c->lm_done);
    if (c->vit_done) hipEventDestroy(c->vit_done);
    for (void* p : c->pinned) hipHostFree(p);
    if (c->gen_pin) hipHostFree(c->gen_pin);
    if (c->gen_out) { hipFree(c->gen_out); hipFree(c->gen_tmp); }
    if (c->gen_ev) hipEventDestroy(c->gen_ev);
    for (auto& kv : c->ingest_tabs) if (kv.second.ready) hipEventDestroy(kv.second.ready);
    for (void* p : c->allocs) hipFree(p);
    for (int k = 0; k < GK_COUNT; ++k)
        for (auto& pr : c->ev[k]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    delete c;
}

extern "C" int aha_ctx_set_tuning(aha_ctx* c, const char* key, int value) {
    if (!c || !key) return AHA_E_INVAL;
    std::string k(key);
    c->tune_epoch++;                                     // captured graphs bake the launch configuration in
    if (k == "split_qkv") c->split[GK_QKV] = value;
    else if (k == "split_o") c->split[GK_O] = value;
    else if (k == "split_gateup") c->split[GK_GATEUP] = value;   // ignored by the fused SwiGLU epilogue (always 1)
    else if (k == "split_down") c->split[GK_DOWN] = value;
    else if (k == "wpb_qkv") c->wpb[GK_QKV] = value;
    else if (k == "wpb_o") c->wpb[GK_O] = value;
    else if (k == "wpb_gateup") c->wpb[GK_GATEUP] = value;
    else if (k == "wpb_down") c->wpb[GK_DOWN] = value;
    else if (k == "attn_split_len") c->attn_split_len = value;
    else if (k == "time_gemm") c->time_gemm = value;
    else if (k == "use_wl") c->use_wl = value;
    else if (k == "dev_xkb") c->dev_xkb = value;
    else if (k == "act_kb") c->act_kb = value;
    else if (k == "wl_bal") aha_gemm_wl_set_balanced(value);
    else if (k == "layer_first") c->layer_first = value;       // with layer_count: run decoder layers [first, first+count) only (parity taps)
    else if (k == "layer_count") c->layer_count = value;
    else if (k == "rerot_pg") aha_sink_rerotate_set_pg(value);   // plane groups of the sink re-rotation kernel (0 = heuristic)
    else if (k == "fuse_static") c->fuse_static = value;
    else if (k == "static_attn") c->static_attn = value;
    else if (k == "pool_subset") c->pool_subset = value;
    else if (k == "use_graph") c->use_graph = value;              // 1 (default): replay frozen-static steps from a captured HIP graph
    else if (k == "fuse_mlp") c->fuse_mlp = value;               // 1: resid_norm + gate/up + down in one launch (M <= 64); 2: sc1 hand-offs
    else if (k == "kc_small") aha_gemm_ws_set_kc_small(value);
    else if (k == "attn_lm") aha_attention_set_lm_kernel(value);   // 1 (default): frame-sized LM steps use attn_lm_kernel (LDS-DMA, all row tiles per workgroup)
    else if (k == "attn_head") aha_attention_set_head_kernel(value);   // whole-head-in-LDS dense (ViT) attention: 0 off, 1 auto, 2 always when eligible
    else if (k == "attn_tpw") aha_attention_set_dense_tpw(value);   // dense attention: query tiles per wave (0 auto)
    else if (k == "tile_dma") aha_gemm_tile_set_dma(value);
    else if (k == "tile_p288s") aha_gemm_tile_p288_set_pipelined(value);   // 1 (default): software-pipelined fragment reads in the persistent tile kernel
    else if (k == "tile_p288") aha_gemm_tile_set_p288(value);    // 1 (default): persistent 288x256 tile kernel on the throughput shapes
    else if (k == "tile_epi") aha_gemm_tile_set_epi(value);      // 1 (default): LDS-transposed wide epilogue of the LDS-DMA tile kernels      // 0 off, 1 auto (default), 2 force
    else return fail(c, AHA_E_NOENT, "unknown tuning key " + k);
    return 0;
}

// --------------------------------------------------------------------------------------------
// weights
// --------------------------------------------------------------------------------------------
typedef std::unordered_map<std::string, const aha_tensor_view*> TMap;

static const aha_tensor_view* need(aha_ctx* c, const TMap& m, const std::string& name, int ndim, int64_t d0, int64_t d1) {
    auto it = m.find(name);
    if (it == m.end()) { c->err = "missing tensor " + name; return nullptr; }
    const aha_tensor_view* t = it->second;
    int64_t numel = 1;
    for (int i = 0; i < t->ndim; ++i) numel *= t->shape[i];
    int64_t want = d0 * (ndim > 1 ? d1 : 1);
    if (numel != want) { c->err = "bad shape for " + name; return nullptr; }
    return t;
}

static int copy_vec(aha_ctx* c, const TMap& m, const std::string& name, int64_t n, bf16** dst, hipStream_t st) {
    const aha_tensor_view* t = need(c, m, name, 1, n, 1);
    if (!t) return AHA_E_NOENT;
    int rc = dalloc(c, dst, (size_t)n);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(*dst, t->data, n * 2, hipMemcpyDeviceToDevice, st));
    return 0;
}

static int alloc_packed(aha_ctx* c, PackedW* w, int n_tiles, int K) {
    w->n_tiles = n_tiles;
    w->K = K;
    w->KS = round_up(ceil_div(K, 32), 8);        // whole chunks for every KC in {1,2,4,8}; pack_w zero-fills k >= K
    return dalloc(c, &w->p, (size_t)n_tiles * w->KS * 64);
}

static int pack_into(aha_ctx* c, const TMap& m, const std::string& name, int N, int K, PackedW* w, int tile_stride,
                     int tile_off, hipStream_t st) {
    const aha_tensor_view* t = need(c, m, name, 2, N, K);
    if (!t) return AHA_E_NOENT;
    HIPCHK(c, aha_pack_w((const bf16*)t->data, N, K, K, w->p, w->KS, tile_stride, tile_off, st));
    return 0;
}

extern "C" int aha_ctx_load_weights(aha_ctx* c, const aha_tensor_view* tensors, size_t n, aha_hip_stream st_) {
    if (c) c->tune_epoch++;                                  // captured graphs hold pointers into the old tables

    if (!c || !tensors) return AHA_E_INVAL;
    if (c->weights_loaded || !c->L.empty())
        return fail(c, AHA_E_INVAL, "weights were already loaded into this context (create a new context to load another checkpoint)");
    hipStream_t st = (hipStream_t)st_;
    HIPCHK(c, hipSetDevice(c->device));
    TMap m;
    for (size_t i = 0; i < n; ++i) m[tensors[i].name] = &tensors[i];
    const aha_model_desc& d = c->d;
    const int H = d.hidden, Dh = d.head_dim, QD = d.heads * Dh, KD = d.kv_heads * Dh, I = d.inter;
    int rc;
    // ---- LM
    c->L.resize(d.layers);
    for (int l = 0; l < d.layers; ++l) {
        LayerW& w = c->L[l];
        const std::string p = "model.layers." + std::to_string(l) + ".";
        if ((rc = copy_vec(c, m, p + "input_layernorm.weight", H, &w.ln1, st))) return rc;
        if ((rc = copy_vec(c, m, p + "post_attention_layernorm.weight", H, &w.ln2, st))) return rc;
        // fused q|k|v
        const int nq = QD / 16, nk = KD / 16;
        if ((rc = alloc_packed(c, &w.qkv, nq + 2 * nk, H))) return rc;
        w.qkv.N = QD + 2 * KD;
        if ((rc = pack_into(c, m, p + "self_attn.q_proj.weight", QD, H, &w.qkv, 1, 0, st))) return rc;
        if ((rc = pack_into(c, m, p + "self_attn.k_proj.weight", KD, H, &w.qkv, 1, nq, st))) return rc;
        if ((rc = pack_into(c, m, p + "self_attn.v_proj.weight", KD, H, &w.qkv, 1, nq + nk, st))) return rc;
        if ((rc = dalloc(c, &w.qkv_bias, (size_t)QD + 2 * KD))) return rc;
        const aha_tensor_view *bq = need(c, m, p + "self_attn.q_proj.bias", 1, QD, 1), *bk = need(c, m, p + "self_attn.k_proj.bias", 1, KD, 1),
                              *bv = need(c, m, p + "self_attn.v_proj.bias", 1, KD, 1);
        if (!bq || !bk || !bv) return AHA_E_NOENT;
        HIPCHK(c, hipMemcpyAsync(w.qkv_bias, bq->data, QD * 2, hipMemcpyDeviceToDevice, st));
        HIPCHK(c, hipMemcpyAsync(w.qkv_bias + QD, bk->data, KD * 2, hipMemcpyDeviceToDevice, st));
        HIPCHK(c, hipMemcpyAsync(w.qkv_bias + QD + KD, bv->data, KD * 2, hipMemcpyDeviceToDevice, st));
        if ((rc = alloc_packed(c, &w.o, H / 16, QD))) return rc;
        w.o.N = H;
        if ((rc = pack_into(c, m, p + "self_attn.o_proj.weight", H, QD, &w.o, 1, 0, st))) return rc;
        // gate/up interleaved by 16-row tiles: tile 2t = gate tile t, tile 2t+1 = up tile t
        if ((rc = alloc_packed(c, &w.gateup, 2 * (I / 16), H))) return rc;
        w.gateup.N = I;
        if ((rc = pack_into(c, m, p + "mlp.gate_proj.weight", I, H, &w.gateup, 2, 0, st))) return rc;
        if ((rc = pack_into(c, m, p + "mlp.up_proj.weight", I, H, &w.gateup, 2, 1, st))) return rc;
        if ((rc = alloc_packed(c, &w.down, H / 16, I))) return rc;
        w.down.N = H;
        if ((rc = pack_into(c, m, p + "mlp.down_proj.weight", H, I, &w.down, 1, 0, st))) return rc;
    }
    if ((rc = copy_vec(c, m, "model.norm.weight", H, &c->final_norm, st))) return rc;
    if ((rc = dalloc(c, &c->heads_w, (size_t)4 * H))) return rc;
    {
        const aha_tensor_view *ti = need(c, m, "informative_head.weight", 2, 2, H), *tr = need(c, m, "relevance_head.weight", 2, 1, H),
                              *tu = need(c, m, "uncertainty_head.weight", 2, 1, H);
        if (!ti || !tr || !tu) return AHA_E_NOENT;
        HIPCHK(c, hipMemcpyAsync(c->heads_w, ti->data, 2 * H * 2, hipMemcpyDeviceToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->heads_w + 2 * H, tr->data, H * 2, hipMemcpyDeviceToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->heads_w + 3 * H, tu->data, H * 2, hipMemcpyDeviceToDevice, st));
    }
    if (m.count("model.embed_tokens.weight")) {
        if ((rc = copy_vec(c, m, "model.embed_tokens.weight", (int64_t)d.vocab * H, &c->embed, st))) return rc;
    }
    if (m.count("lm_head.weight")) {
        if ((rc = alloc_packed(c, &c->lm_head, ceil_div(d.vocab, 16), H))) return rc;
        c->lm_head.N = d.vocab;
        if ((rc = pack_into(c, m, "lm_head.weight", d.vocab, H, &c->lm_head, 1, 0, st))) return rc;
    }
    // ---- vision
    const int Dv = d.v_hidden, PP3 = 3 * d.patch_size * d.patch_size;
    {
        const aha_tensor_view* t = need(c, m, "vision.embeddings.patch_embedding.weight", 2, Dv, PP3);
        if (!t) return AHA_E_NOENT;
        if ((rc = dalloc(c, &c->patch_w, (size_t)Dv * c->Kp))) return rc;
        HIPCHK(c, hipMemsetAsync(c->patch_w, 0, (size_t)Dv * c->Kp * 2, st));
        HIPCHK(c, hipMemcpy2DAsync(c->patch_w, (size_t)c->Kp * 2, t->data, (size_t)PP3 * 2, (size_t)PP3 * 2, Dv, hipMemcpyDeviceToDevice, st));
    }
    if (d.v_kind == AHA_VISION_CLIP) {
        // CLIPVisionEmbeddings: no patch bias; class_embedding; Np + 1 positions with the class token's row FIRST in the
        // checkpoint - stored here patches first, class token last (the tower's internal token order, see vit_tower)
        c->patch_b = nullptr;
        if ((rc = copy_vec(c, m, "vision.embeddings.class_embedding", Dv, &c->cls_emb, st))) return rc;
        const aha_tensor_view* tp = need(c, m, "vision.embeddings.position_embedding.weight", 2, c->Tt, Dv);
        if (!tp) return AHA_E_NOENT;
        if ((rc = dalloc(c, &c->pos_emb, (size_t)c->Tt * Dv))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->pos_emb, (const bf16*)tp->data + Dv, (size_t)c->Np * Dv * 2, hipMemcpyDeviceToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->pos_emb + (size_t)c->Np * Dv, tp->data, (size_t)Dv * 2, hipMemcpyDeviceToDevice, st));
        if ((rc = copy_vec(c, m, "vision.pre_layrnorm.weight", Dv, &c->pre_ln_w, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.pre_layrnorm.bias", Dv, &c->pre_ln_b, st))) return rc;
    } else {
        if ((rc = copy_vec(c, m, "vision.embeddings.patch_embedding.bias", Dv, &c->patch_b, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.embeddings.position_embedding.weight", (int64_t)c->Np * Dv, &c->pos_emb, st))) return rc;
    }
    c->V.resize(d.v_layers);
    for (int l = 0; l < d.v_layers; ++l) {
        VLayerW& w = c->V[l];
        const std::string p = "vision.encoder.layers." + std::to_string(l) + ".";
        if ((rc = copy_vec(c, m, p + "layer_norm1.weight", Dv, &w.ln1w, st))) return rc;
        if ((rc = copy_vec(c, m, p + "layer_norm1.bias", Dv, &w.ln1b, st))) return rc;
        if ((rc = copy_vec(c, m, p + "layer_norm2.weight", Dv, &w.ln2w, st))) return rc;
        if ((rc = copy_vec(c, m, p + "layer_norm2.bias", Dv, &w.ln2b, st))) return rc;
        if ((rc = dalloc(c, &w.wqkv, (size_t)3 * Dv * Dv))) return rc;
        if ((rc = dalloc(c, &w.bqkv, (size_t)3 * Dv))) return rc;
        const char* names[3] = {"q_proj", "k_proj", "v_proj"};
        for (int j = 0; j < 3; ++j) {
            const aha_tensor_view *tw = need(c, m, p + "self_attn." + names[j] + ".weight", 2, Dv, Dv),
                                  *tb = need(c, m, p + "self_attn." + names[j] + ".bias", 1, Dv, 1);
            if (!tw || !tb) return AHA_E_NOENT;
            HIPCHK(c, hipMemcpyAsync(w.wqkv + (size_t)j * Dv * Dv, tw->data, (size_t)Dv * Dv * 2, hipMemcpyDeviceToDevice, st));
            HIPCHK(c, hipMemcpyAsync(w.bqkv + (size_t)j * Dv, tb->data, (size_t)Dv * 2, hipMemcpyDeviceToDevice, st));
        }
        if ((rc = copy_vec(c, m, p + "self_attn.out_proj.weight", (int64_t)Dv * Dv, &w.wo, st))) return rc;
        if ((rc = copy_vec(c, m, p + "self_attn.out_proj.bias", Dv, &w.bo, st))) return rc;
        if ((rc = copy_vec(c, m, p + "mlp.fc1.weight", (int64_t)d.v_inter * Dv, &w.w1, st))) return rc;
        if ((rc = copy_vec(c, m, p + "mlp.fc1.bias", d.v_inter, &w.b1, st))) return rc;
        {
            const aha_tensor_view* t = need(c, m, p + "mlp.fc2.weight", 2, Dv, d.v_inter);
            if (!t) return AHA_E_NOENT;
            if ((rc = dalloc(c, &w.w2, (size_t)Dv * c->Fp))) return rc;
            HIPCHK(c, hipMemsetAsync(w.w2, 0, (size_t)Dv * c->Fp * 2, st));
            HIPCHK(c, hipMemcpy2DAsync(w.w2, (size_t)c->Fp * 2, t->data, (size_t)d.v_inter * 2, (size_t)d.v_inter * 2, Dv,
                                       hipMemcpyDeviceToDevice, st));
        }
        if ((rc = copy_vec(c, m, p + "mlp.fc2.bias", Dv, &w.b2, st))) return rc;
    }
    if (m.count("vision.post_layernorm.weight") && m.count("vision.post_layernorm.bias")) {
        if ((rc = copy_vec(c, m, "vision.post_layernorm.weight", Dv, &c->post_ln_w, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.post_layernorm.bias", Dv, &c->post_ln_b, st))) return rc;
    }
    if (m.count("vision.head.probe")) {                     // all or nothing: need() reports the first missing tensor
        const int64_t F = d.v_inter;
        if ((rc = copy_vec(c, m, "vision.head.probe", Dv, &c->hd_probe, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.head.attention.in_proj_weight", (int64_t)3 * Dv * Dv, &c->hd_in_w, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.head.attention.in_proj_bias", 3 * Dv, &c->hd_in_b, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.head.attention.out_proj.weight", (int64_t)Dv * Dv, &c->hd_out_w, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.head.attention.out_proj.bias", Dv, &c->hd_out_b, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.head.layernorm.weight", Dv, &c->hd_ln_w, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.head.layernorm.bias", Dv, &c->hd_ln_b, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.head.mlp.fc1.weight", F * Dv, &c->hd_w1, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.head.mlp.fc1.bias", F, &c->hd_b1, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.head.mlp.fc2.weight", (int64_t)Dv * F, &c->hd_w2, st))) return rc;
        if ((rc = copy_vec(c, m, "vision.head.mlp.fc2.bias", Dv, &c->hd_b2, st))) return rc;
        if ((rc = dalloc(c, &c->hd_q, (size_t)Dv))) return rc;
        c->hd_q_ready = false;
    }
    if ((rc = copy_vec(c, m, "mm_projector.0.weight", (int64_t)H * Dv, &c->p0w, st))) return rc;
    if ((rc = copy_vec(c, m, "mm_projector.0.bias", H, &c->p0b, st))) return rc;
    if ((rc = copy_vec(c, m, "mm_projector.2.weight", (int64_t)H * H, &c->p2w, st))) return rc;
    if ((rc = copy_vec(c, m, "mm_projector.2.bias", H, &c->p2b, st))) return rc;
    HIPCHK(c, hipStreamSynchronize(st));      // sources may be freed by the caller after return
    c->weights_loaded = true;
    return 0;
}

extern "C" int aha_ctx_set_rope_table(aha_ctx* c, const void* cosb, const void* sinb, int n_pos, aha_hip_stream st_) {
    if (c) c->tune_epoch++;                                  // captured graphs hold pointers into the old tables

    if (!c || !cosb || !sinb || n_pos <= 0) return AHA_E_INVAL;
    hipStream_t st = (hipStream_t)st_;
    int rc;
    const size_t n = (size_t)n_pos * c->d.head_dim;
    if ((rc = dalloc(c, &c->rope_cos, n))) return rc;
    if ((rc = dalloc(c, &c->rope_sin, n))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->rope_cos, cosb, n * 2, hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->rope_sin, sinb, n * 2, hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    c->n_pos = n_pos;
    return 0;
}

extern "C" int aha_ctx_set_rerotation_table(aha_ctx* c, int window, int n_sink, int T, const void* cosb, const void* sinb,
                                            aha_hip_stream st_) {
    if (!c || !cosb || !sinb) return AHA_E_INVAL;
    const int rows = window - n_sink - T;
    if (rows <= 0) return fail(c, AHA_E_RANGE, "rerotation table needs window - n_sink - T > 0");
    hipStream_t st = (hipStream_t)st_;
    auto key = std::make_tuple(window, n_sink, T);
    if (c->rerot.count(key)) return 0;
    bf16 *pc, *ps;
    int rc;
    const size_t n = (size_t)rows * c->d.head_dim;
    if ((rc = dalloc(c, &pc, n))) return rc;
    if ((rc = dalloc(c, &ps, n))) return rc;
    HIPCHK(c, hipMemcpyAsync(pc, cosb, n * 2, hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipMemcpyAsync(ps, sinb, n * 2, hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    c->rerot[key] = {pc, ps};
    return 0;
}

extern "C" int aha_ctx_has_rerotation_table(aha_ctx* c, int window, int n_sink, int T) {
    return c && c->rerot.count(std::make_tuple(window, n_sink, T)) ? 1 : 0;
}

// --------------------------------------------------------------------------------------------
// streams
// --------------------------------------------------------------------------------------------
extern "C" int aha_stream_open(aha_ctx* c, int policy, int window, int n_sink, int capacity, aha_stream** out) {
    if (!c || !out) return AHA_E_INVAL;
    if (policy < AHA_CACHE_NONE || policy > AHA_CACHE_STATIC) return fail(c, AHA_E_INVAL, "bad cache policy");
    if (policy != AHA_CACHE_NONE && window <= 0) return fail(c, AHA_E_INVAL, "window must be > 0");
    if (policy == AHA_CACHE_SINK && (n_sink < 0 || n_sink >= window)) return fail(c, AHA_E_INVAL, "bad n_sink");
    if (policy == AHA_CACHE_NONE && capacity <= 0) return fail(c, AHA_E_INVAL, "capacity must be > 0");
    aha_stream* s = new aha_stream();
    s->ctx = c;
    s->device = c->device;
    s->policy = policy;
    s->W = window;
    s->sink = policy == AHA_CACHE_SINK ? n_sink : 0;
    s->cap = policy == AHA_CACHE_NONE ? capacity : window;
    const size_t n = (size_t)c->d.layers * c->d.kv_heads * s->cap * c->d.head_dim;
    if (hipSetDevice(c->device) != hipSuccess || hipMalloc((void**)&s->k, n * 2) != hipSuccess ||
        hipMalloc((void**)&s->v, n * 2) != hipSuccess) {
        if (s->k) hipFree(s->k);
        delete s;
        return fail(c, AHA_E_NOMEM, "KV cache allocation failed");
    }
    hipMemset(s->k, 0, n * 2);
    hipMemset(s->v, 0, n * 2);
    *out = s;
    return 0;
}
extern "C" int aha_stream_reset(aha_stream* s) {
    if (!s) return AHA_E_INVAL;
    s->len = s->head = s->seen = 0;
    s->op_valid = false;
    s->poisoned = false;
    return 0;
}
extern "C" int aha_stream_seq_length(const aha_stream* s) { return s ? s->len : AHA_E_INVAL; }
extern "C" int aha_stream_seen_tokens(const aha_stream* s) { return s ? s->seen : AHA_E_INVAL; }
extern "C" int aha_stream_set_attn_semantics(aha_stream* s, int sem) {
    if (!s || (sem != AHA_ATTN_TRAILING && sem != AHA_ATTN_HF449_SDPA && sem != AHA_ATTN_FA2)) return AHA_E_INVAL;
    s->semantics = sem;
    return 0;
}
extern "C" int aha_stream_set_position_offset(aha_stream* s, int offset) {
    if (!s || offset < 0) return AHA_E_INVAL;
    s->pos_off = offset;
    return 0;
}
extern "C" void aha_stream_destroy(aha_stream* s) {
    if (!s) return;
    hipSetDevice(s->device);
    hipDeviceSynchronize();
    hipFree(s->k);
    hipFree(s->v);
    delete s;
}

// Advance one stream's bookkeeping by T new tokens and describe the step for the kernels.
// Follows SinkCache.update (test/sink_cache.py:123-162), SlidingWindowCache.update
// (test/sliding_window_cache.py:28-44), TrulyStaticCache.update (test/static_cache.py:26-36) and
// DynamicCache; position rule: positions = get_seq_length() + arange(T).
static int plan_stream(aha_ctx* c, aha_stream* s, int T, StreamStep* o) {
    const int L = s->len, W = s->W;
    memset(o, 0, sizeof(*o));
    o->k_base = s->k;
    o->v_base = s->v;
    o->cap = s->cap;
    o->pos_base = L + s->pos_off;
    o->ring_cap = 1;
    o->write_count = T;
    int new_len = L, new_head = s->head;
    bool shifted = false;
    switch (s->policy) {
        case AHA_CACHE_NONE:
            if (L + T > s->cap) return fail(c, AHA_E_RANGE, "stream capacity exceeded (AHA_CACHE_NONE)");
            o->n_fixed = s->cap;
            o->write_base = L;
            new_len = L + T;
            break;
        case AHA_CACHE_STATIC:
            o->n_fixed = s->cap;
            if (L == 0) {
                o->write_base = 0;
                o->write_count = T < W ? T : W;
                new_len = o->write_count;
            } else {
                o->write_base = -1;
                o->write_count = 0;
            }
            break;
        case AHA_CACHE_SLIDING:
            if (T > W) return fail(c, AHA_E_RANGE, "T > window unsupported (SlidingWindowCache)");
            o->n_fixed = 0;
            o->ring_cap = W;
            if (L + T <= W) {
                o->write_base = L;
                new_len = L + T;
            } else {
                new_head = (s->head + (L + T - W)) % W;
                new_len = W;
                o->write_base = W - T;
                shifted = true;
            }
            break;
        case AHA_CACHE_SINK: {
            o->n_fixed = s->sink;
            o->ring_cap = W - s->sink;
            if (L == 0 ? T < W : L + T < W) {
                o->write_base = L;
                new_len = L + T;
            } else {
                if (L == 0) return fail(c, AHA_E_RANGE, "first chunk >= window unsupported (SinkCache)");
                const int keep = W - s->sink - T;
                if (keep <= 0 || L < s->sink) return fail(c, AHA_E_RANGE, "T too large for window - n_sink (SinkCache)");
                new_head = (s->head + (L + T - W)) % o->ring_cap;
                new_len = W;
                o->write_base = W - T;
                o->n_rerot = keep;
                o->rerot_row0 = 0;
                shifted = true;
            }
            break;
        }
    }
    o->ring_head = new_head;
    o->len_after = new_len;
    if (s->policy == AHA_CACHE_STATIC)
        // first call: plain causal.  Frozen: the cache returns the prefix only; sdpa-style masks make all of it visible,
        // flash-attn-2 (the reference's default attn_implementation, models/arguments_live.py:30) aligns its causal mask bottom-right:
        // key j visible to new token t iff j <= t + (L - T)
        o->causal_off = (L == 0) ? 0 : (s->semantics == AHA_ATTN_FA2 ? L - T : (1 << 29));
    else if (s->semantics == AHA_ATTN_HF449_SDPA)
        o->causal_off = L;                                         // key j visible iff j <= L_before + t
    else
        o->causal_off = new_len - T;                               // trailing T x T block causal
    (void)shifted;
    s->len = new_len;
    s->head = new_head;
    s->seen += T;
    return 0;
}

__global__ void export_kv_kernel(StreamStep ss, int layer, int Hkv, int D, int want_v, int len, bf16* out) {
    const int j = blockIdx.x, hk = blockIdx.y;
    const int slot = phys_slot(ss, j);
    const bf16* src = (want_v ? ss.v_base : ss.k_base) + (((long)layer * Hkv + hk) * ss.cap + slot) * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) out[((long)hk * len + j) * D + d] = src[d];
}

extern "C" int aha_stream_export_kv(aha_ctx* c, const aha_stream* s, int layer, int want_v, void* out, aha_hip_stream st) {
    if (!c || !s || !out) return AHA_E_INVAL;
    if (s->len == 0) return 0;
    StreamStep ss;
    memset(&ss, 0, sizeof(ss));
    ss.k_base = s->k; ss.v_base = s->v; ss.cap = s->cap; ss.ring_head = s->head;
    if (s->policy == AHA_CACHE_NONE || s->policy == AHA_CACHE_STATIC) { ss.n_fixed = s->cap; ss.ring_cap = 1; }
    else if (s->policy == AHA_CACHE_SLIDING) { ss.n_fixed = 0; ss.ring_cap = s->W; }
    else { ss.n_fixed = s->sink; ss.ring_cap = s->W - s->sink; }
    hipLaunchKernelGGL(export_kv_kernel, dim3(s->len, c->d.kv_heads), dim3(64), 0, (hipStream_t)st, ss, layer, c->d.kv_heads,
                       c->d.head_dim, want_v, s->len, (bf16*)out);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------------------------
// vision
// --------------------------------------------------------------------------------------------
static hipError_t tile_gemm(const bf16* A, int lda, int M, const bf16* W, int ldw, int N, int K, bf16* C, int ldc, const bf16* bias,
                            int act, const bf16* residual, int ldr, const bf16* rowadd, int period, int ldra, hipStream_t st) {
    GemmTileArgs g;
    g.A = A; g.lda = lda; g.M = M; g.W = W; g.ldw = ldw; g.N = N; g.K = K; g.C = C; g.ldc = ldc; g.bias = bias; g.act = act;
    g.residual = residual; g.ldr = ldr; g.rowadd = rowadd; g.rowadd_period = period > 0 ? period : 1; g.ldra = ldra;
    return aha_gemm_tile(&g, st);
}

static int vit_layers(aha_ctx* c, int n, int l0, int l1, hipStream_t st);
static int vit_tower(aha_ctx* c, const uint8_t* frames, int n, hipStream_t st) {
    const aha_model_desc& d = c->d;
    const bool clip = d.v_kind == AHA_VISION_CLIP;
    const int Dv = d.v_hidden, T = c->Tt, rows = n * T, vhd = Dv / d.v_heads;
    HIPCHK(c, aha_im2col_norm(frames, n, d.image_size, d.patch_size, c->Kp, c->px_mean, c->px_std, c->v_a0, st));
    if (!clip) {
        HIPCHK(c, tile_gemm(c->v_a0, c->Kp, n * c->Np, c->patch_w, c->Kp, Dv, c->Kp, c->v_x, Dv, c->patch_b, ACT_NONE, nullptr, 0,
                            c->pos_emb, c->Np, Dv, st));
    } else {
        // CLIP (transformers modeling_clip.py, CLIPVisionEmbeddings + pre_layrnorm): bias-free patch conv, class token, positions,
        // then a LayerNorm over every token before layer 0.  The class token is row Np of each frame's T = Np + 1 rows.
        HIPCHK(c, tile_gemm(c->v_a0, c->Kp, n * c->Np, c->patch_w, c->Kp, Dv, c->Kp, c->v_patch, Dv, nullptr, ACT_NONE, nullptr, 0, nullptr, 0, 0, st));
        HIPCHK(c, aha_clip_assemble(c->v_patch, c->cls_emb, c->pos_emb, c->v_h, n, c->Np, Dv, st));
        HIPCHK(c, aha_layernorm(c->v_h, Dv, c->pre_ln_w, c->pre_ln_b, c->v_x, Dv, rows, Dv, d.v_ln_eps, st));
    }
    return vit_layers(c, n, 0, d.v_layers, st);
}

// Encoder layers [l0, l1) of the tower on the hidden state in c->v_x ([n * Tt][Dv]), in place
// (SiglipEncoderLayer / CLIPEncoderLayer: pre-LN attention block + pre-LN MLP block, each with its residual).
static int vit_layers(aha_ctx* c, int n, int l0, int l1, hipStream_t st) {
    const aha_model_desc& d = c->d;
    const bool clip = d.v_kind == AHA_VISION_CLIP;
    const int Dv = d.v_hidden, T = c->Tt, rows = n * T, vhd = Dv / d.v_heads;
    const int act = clip ? ACT_QUICK_GELU : ACT_GELU_TANH;
    for (int l = l0; l < l1; ++l) {
        const VLayerW& w = c->V[l];
        HIPCHK(c, aha_layernorm(c->v_x, Dv, w.ln1w, w.ln1b, c->v_h, Dv, rows, Dv, d.v_ln_eps, st));
        HIPCHK(c, tile_gemm(c->v_h, Dv, rows, w.wqkv, Dv, 3 * Dv, Dv, c->v_qkv, 3 * Dv, w.bqkv, ACT_NONE, nullptr, 0, nullptr, 0, 0, st));
        AttnArgs a;
        memset(&a, 0, sizeof(a));
        a.q = c->v_qkv; a.q_bs = (long)T * 3 * Dv; a.ldq = 3 * Dv;
        a.k = c->v_qkv + Dv; a.v = c->v_qkv + 2 * Dv; a.kv_bs = (long)T * 3 * Dv; a.ldk = 3 * Dv;
        a.out = c->v_attn; a.o_bs = (long)T * Dv; a.ldo = Dv;
        a.T = T; a.G = 1; a.Hkv = d.v_heads; a.Lk = T;
        a.split_len = round_up(T, 64); a.n_splits = 1;
        a.scale = 1.0f / sqrtf((float)vhd);
        HIPCHK(c, aha_attention(&a, nullptr, n, vhd, st));
        HIPCHK(c, tile_gemm(c->v_attn, Dv, rows, w.wo, Dv, Dv, Dv, c->v_x, Dv, w.bo, ACT_NONE, c->v_x, Dv, nullptr, 0, 0, st));
        HIPCHK(c, aha_layernorm(c->v_x, Dv, w.ln2w, w.ln2b, c->v_h, Dv, rows, Dv, d.v_ln_eps, st));
        HIPCHK(c, tile_gemm(c->v_h, Dv, rows, w.w1, Dv, d.v_inter, Dv, c->v_f, c->Fp, w.b1, act, nullptr, 0, nullptr, 0, 0, st));
        HIPCHK(c, tile_gemm(c->v_f, c->Fp, rows, w.w2, c->Fp, Dv, c->Fp, c->v_x, Dv, w.b2, ACT_NONE, c->v_x, Dv, nullptr, 0, 0, st));
    }
    return 0;
}

static int vit_check(aha_ctx* c, const void* frames, const void* out, int n) {
    if (!c || !frames || !out) return AHA_E_INVAL;
    if (!c->weights_loaded) return fail(c, AHA_E_INVAL, "weights not loaded");
    if (n > c->d.max_vit_frames) return fail(c, AHA_E_RANGE, "n_frames > max_vit_frames");
    return 0;
}

// ---- frame ingest (ingest.hip) -------------------------------------------------------------------------------
void aha_ingest_pil_tables(int in_size, int out_size, int* ksize_out, std::vector<int>* bounds, std::vector<int>* kk);
void aha_ingest_cv_tables(int src_size, int dst_size, bool horizontal, std::vector<int>* tab);

// once per geometry: staged in pinned host memory that lives as long as the context, copied asynchronously on the caller's
// stream (the ingest kernel that reads the table is enqueued behind it) - no host synchronisation on the frame path
static int upload_ints(aha_ctx* c, const std::vector<int>& v, int** dst, hipStream_t st) {
    int rc = dalloc(c, dst, v.size());
    if (rc) return rc;
    void* pin = nullptr;
    const size_t bytes = (v.empty() ? 1 : v.size()) * sizeof(int);
    if (hipHostMalloc(&pin, bytes, hipHostMallocDefault) != hipSuccess) return fail(c, AHA_E_NOMEM, "hipHostMalloc failed");
    c->pinned.push_back(pin);
    memcpy(pin, v.data(), v.size() * sizeof(int));
    HIPCHK(c, hipMemcpyAsync(*dst, pin, v.size() * sizeof(int), hipMemcpyHostToDevice, st));
    return 0;
}

extern "C" int aha_frame_ingest(aha_ctx* c, const uint8_t* src, int height, int width, int src_is_bgr, int method,
                                uint8_t* out, aha_hip_stream st_) {
    if (!c) return AHA_E_INVAL;
    if (!src || !out) return fail(c, AHA_E_INVAL, "null frame pointer");
    if (method != AHA_RESIZE_PIL_BICUBIC && method != AHA_RESIZE_CV2_LINEAR) return fail(c, AHA_E_INVAL, "unknown resize method");
    if (height <= 0 || width <= 0 || height > 16384 || width > 16384) return fail(c, AHA_E_RANGE, "frame size out of range");
    const int S = c->d.image_size;
    // test/live_infer_for_video.py:108-119: the long side becomes S, the short side int((short / long) * S) in double
    int new_w, new_h;
    if (width > height) { new_w = S; new_h = (int)(((double)height / (double)width) * S); }
    else { new_h = S; new_w = (int)(((double)width / (double)height) * S); }
    if (new_w < 1 || new_h < 1) return fail(c, AHA_E_RANGE, "aspect ratio leaves an empty resized frame");
    auto key = std::make_tuple(method, height, width);
    auto it = c->ingest_tabs.find(key);
    if (it == c->ingest_tabs.end()) {
        aha_ctx::IngestTab t;
        int rc;
        if (method == AHA_RESIZE_PIL_BICUBIC) {
            std::vector<int> b, k;
            if (new_w != width) {
                aha_ingest_pil_tables(width, new_w, &t.xks, &b, &k);
                if ((rc = upload_ints(c, b, &t.xb, (hipStream_t)st_)) || (rc = upload_ints(c, k, &t.xk, (hipStream_t)st_))) return rc;
            }
            if (new_h != height) {
                aha_ingest_pil_tables(height, new_h, &t.yks, &b, &k);
                if ((rc = upload_ints(c, b, &t.yb, (hipStream_t)st_)) || (rc = upload_ints(c, k, &t.yk, (hipStream_t)st_))) return rc;
            }
        } else if (new_w != width || new_h != height) {
            std::vector<int> tab;
            aha_ingest_cv_tables(width, new_w, true, &tab);
            if ((rc = upload_ints(c, tab, &t.xb, (hipStream_t)st_))) return rc;
            aha_ingest_cv_tables(height, new_h, false, &tab);
            if ((rc = upload_ints(c, tab, &t.yb, (hipStream_t)st_))) return rc;
        }
        t.up_stream = (hipStream_t)st_;
        HIPCHK(c, hipEventCreateWithFlags(&t.ready, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(t.ready, t.up_stream));
        it = c->ingest_tabs.emplace(key, t).first;
    }
    const aha_ctx::IngestTab& t = it->second;
    if (t.up_stream != (hipStream_t)st_) HIPCHK(c, hipStreamWaitEvent((hipStream_t)st_, t.ready, 0));
    IngestArgs a{};
    a.src = src; a.h = height; a.w = width; a.src_bgr = src_is_bgr ? 1 : 0;
    a.out = out; a.S = S;
    a.new_w = new_w; a.new_h = new_h; a.left = (S - new_w) / 2; a.top = (S - new_h) / 2;
    a.need_h = new_w != width; a.need_v = new_h != height;
    a.xb = t.xb; a.xk = t.xk; a.xks = t.xks; a.yb = t.yb; a.yk = t.yk; a.yks = t.yks;
    HIPCHK(c, aha_ingest_launch(&a, method, (hipStream_t)st_));
    return 0;
}

extern "C" int aha_vit_encode(aha_ctx* c, const uint8_t* frames, int n, void* out_embeds, aha_hip_stream st_) {
    int rc = vit_check(c, frames, out_embeds, n);
    if (rc || n <= 0) return rc;
    hipStream_t st = (hipStream_t)st_;
    ORDER_VIT(c, st);
    const aha_model_desc& d = c->d;
    // With a CLIP tower (LLaVA's select_feature = 'patch') the projector also runs over the class-token rows (1 in Np + 1,
    // cheaper than compacting) and the pooling reads the Np patch rows of each frame's Tt.
    const int Dv = d.v_hidden, rows = n * c->Tt, H = d.hidden;
    if ((rc = vit_tower(c, frames, n, st))) return rc;
    // Bilinear pooling with an even integer stride samples only 4 go^2 of the g^2 patch rows, with weights 1/2: run the
    // projector on those rows only and pool the compact (2 go)^2 grid - same values, same arithmetic, bit-identical embeddings
    // (elementwise.hip: gather_pool_rows_kernel; tuning "pool_subset").  24 -> 6: 144 of 576 rows, 75 % of the projector saved.
    const int s = c->go > 0 ? c->grid / c->go : 0;
    if (c->pool_subset && d.pool_mode == 0 && c->go > 0 && c->grid % c->go == 0 && s >= 4 && s % 2 == 0) {
        const int gc = 2 * c->go, crow = n * gc * gc;
        HIPCHK(c, aha_gather_pool_rows(c->v_x, c->v_h, n, c->grid, c->go, s, Dv, c->Tt, st));       // v_h: free after the tower
        HIPCHK(c, tile_gemm(c->v_h, Dv, crow, c->p0w, Dv, H, Dv, c->v_p1, H, c->p0b, ACT_GELU_ERF, nullptr, 0, nullptr, 0, 0, st));
        HIPCHK(c, tile_gemm(c->v_p1, H, crow, c->p2w, H, H, H, c->v_p2, H, c->p2b, ACT_NONE, nullptr, 0, nullptr, 0, 0, st));
        HIPCHK(c, aha_pool(c->v_p2, (bf16*)out_embeds, n, gc, c->go, H, 2, 0, gc * gc, st));
        return 0;
    }
    HIPCHK(c, tile_gemm(c->v_x, Dv, rows, c->p0w, Dv, H, Dv, c->v_p1, H, c->p0b, ACT_GELU_ERF, nullptr, 0, nullptr, 0, 0, st));
    HIPCHK(c, tile_gemm(c->v_p1, H, rows, c->p2w, H, H, H, c->v_p2, H, c->p2b, ACT_NONE, nullptr, 0, nullptr, 0, 0, st));
    HIPCHK(c, aha_pool(c->v_p2, (bf16*)out_embeds, n, c->grid, c->go, H, d.pool_stride, d.pool_mode, c->Tt, st));
    return 0;
}

// The encode contract of models/vision_live.py:11-31 (_siglip_vision_encode) and :34-54 (_clip_vision_encode):
// tower -> last_hidden_state (SigLIP: + post_layernorm) -> adaptive_avg_pool2d to pooled x pooled (frame_token_pooled) and, with
// frame_token_cls, the class token in front of it - SigLIP: pooler_output = the attention-pooling head on the post-layernormed
// tokens (a learned probe attends over them, then x + mlp(layernorm(x))); CLIP: last_hidden_state[:, 0], returned only WITHOUT
// pooling (the reference's torch.cat of [N, D] and [N, P, D] at vision_live.py:54 raises: refused here as well) -> connector.
// Pooling happens BEFORE the projector, so the projector runs on cls + pooled^2 rows per frame.
extern "C" int aha_vit_encode_live(aha_ctx* c, const uint8_t* frames, int n, int pooled, int cls, void* out_embeds, aha_hip_stream st_) {
    int rc = vit_check(c, frames, out_embeds, n);
    if (rc || n <= 0) return rc;
    const bool clip = c->d.v_kind == AHA_VISION_CLIP;
    if (!clip && !c->post_ln_w) return fail(c, AHA_E_NOENT, "vision.post_layernorm.{weight,bias} were not loaded");
    if (pooled < 0 || pooled > c->grid || (pooled == 0 && !cls)) return fail(c, AHA_E_RANGE, "pooled grid must be in 1..patch grid (0: class token only)");
    if (cls && clip && pooled) return fail(c, AHA_E_INVAL, "_clip_vision_encode cannot return the class token together with pooled tokens (models/vision_live.py:54 raises)");
    if (cls && !clip && !c->hd_probe) return fail(c, AHA_E_NOENT, "vision.head.* (attention-pooling head) was not loaded");
    // the token assembly and the connector write n * (cls + pooled^2) rows into workspaces sized max_vit_frames * Tt rows
    if ((long)n * ((cls ? 1 : 0) + pooled * pooled) > (long)c->d.max_vit_frames * c->Tt)
        return fail(c, AHA_E_RANGE, "n_frames * (class token + pooled^2) exceeds the vision workspace (max_vit_frames * tokens per frame)");
    hipStream_t st = (hipStream_t)st_;
    ORDER_VIT(c, st);
    const aha_model_desc& d = c->d;
    const int Dv = d.v_hidden, rows = n * c->Tt, H = d.hidden, P = pooled * pooled, tok = (cls ? 1 : 0) + P, prow = n * tok;
    if ((rc = vit_tower(c, frames, n, st))) return rc;
    const bf16* tokens = c->v_attn;                          // [n][tok][Dv] rows handed to the connector
    if (clip) {
        // last_hidden_state is the encoder output (transformers applies post_layernorm to the pooled class token only); the class
        // token is the last row of each frame here.  Pooling runs over the first Np rows of each frame's Tt.
        if (cls) HIPCHK(c, hipMemcpy2DAsync(c->v_attn, (size_t)Dv * 2, c->v_x + (size_t)(c->Tt - 1) * Dv, (size_t)c->Tt * Dv * 2, (size_t)Dv * 2, n,
                                            hipMemcpyDeviceToDevice, st));
        else HIPCHK(c, aha_pool(c->v_x, c->v_attn, n, c->grid, pooled, Dv, 0, 3, c->Tt, st));
    } else {
        HIPCHK(c, aha_layernorm(c->v_x, Dv, c->post_ln_w, c->post_ln_b, c->v_h, Dv, rows, Dv, d.v_ln_eps, st));
        if (P) HIPCHK(c, aha_pool(c->v_h, c->v_attn, n, c->grid, pooled, Dv, 0, 3, 0, st));
        if (cls) {
            const int F = d.v_inter, vhd = Dv / d.v_heads;
            if (!c->hd_q_ready) {                            // the probe is a parameter: its query projection is computed once
                HIPCHK(c, tile_gemm(c->hd_probe, Dv, 1, c->hd_in_w, Dv, Dv, Dv, c->hd_q, Dv, c->hd_in_b, ACT_NONE, nullptr, 0, nullptr, 0, 0, st));
                c->hd_q_ready = true;
            }
            // K | V of every token (nn.MultiheadAttention's packed in_proj rows D..3D) into the tower's qkv buffer
            HIPCHK(c, tile_gemm(c->v_h, Dv, rows, c->hd_in_w + (size_t)Dv * Dv, Dv, 2 * Dv, Dv, c->v_qkv + Dv, 3 * Dv, c->hd_in_b + Dv, ACT_NONE,
                                nullptr, 0, nullptr, 0, 0, st));
            bf16 *hb0 = c->v_p1, *hb1 = hb0 + (size_t)n * Dv, *hb2 = hb1 + (size_t)n * Dv, *cl = hb2 + (size_t)n * Dv;   // [n][Dv] each; v_p1 is idle until the connector
            AttnArgs a;
            memset(&a, 0, sizeof(a));
            a.q = c->hd_q; a.q_bs = 0; a.ldq = Dv;          // one query row, shared by every frame
            a.k = c->v_qkv + Dv; a.v = c->v_qkv + 2 * Dv; a.kv_bs = (long)c->Tt * 3 * Dv; a.ldk = 3 * Dv;
            a.out = hb0; a.o_bs = Dv; a.ldo = Dv;
            a.T = 1; a.G = 1; a.Hkv = d.v_heads; a.Lk = c->Tt;
            a.split_len = round_up(c->Tt, 64); a.n_splits = 1;
            a.scale = 1.0f / sqrtf((float)vhd);
            HIPCHK(c, aha_attention(&a, nullptr, n, vhd, st));
            HIPCHK(c, tile_gemm(hb0, Dv, n, c->hd_out_w, Dv, Dv, Dv, hb1, Dv, c->hd_out_b, ACT_NONE, nullptr, 0, nullptr, 0, 0, st));
            HIPCHK(c, aha_layernorm(hb1, Dv, c->hd_ln_w, c->hd_ln_b, hb2, Dv, n, Dv, d.v_ln_eps, st));
            HIPCHK(c, tile_gemm(hb2, Dv, n, c->hd_w1, Dv, F, Dv, c->v_f, c->Fp, c->hd_b1, ACT_GELU_TANH, nullptr, 0, nullptr, 0, 0, st));
            HIPCHK(c, tile_gemm(c->v_f, c->Fp, n, c->hd_w2, F, Dv, F, cl, Dv, c->hd_b2, ACT_NONE, hb1, Dv, nullptr, 0, 0, st));
            if (P) {                                         // [class token | pooled grid] per frame, assembled in the (now idle) tower output buffer
                HIPCHK(c, hipMemcpy2DAsync(c->v_x, (size_t)tok * Dv * 2, cl, (size_t)Dv * 2, (size_t)Dv * 2, n, hipMemcpyDeviceToDevice, st));
                HIPCHK(c, hipMemcpy2DAsync(c->v_x + Dv, (size_t)tok * Dv * 2, c->v_attn, (size_t)P * Dv * 2, (size_t)P * Dv * 2, n,
                                           hipMemcpyDeviceToDevice, st));
                tokens = c->v_x;
            } else {
                HIPCHK(c, hipMemcpyAsync(c->v_x, cl, (size_t)n * Dv * 2, hipMemcpyDeviceToDevice, st));
                tokens = c->v_x;
            }
        }
    }
    HIPCHK(c, tile_gemm(tokens, Dv, prow, c->p0w, Dv, H, Dv, c->v_p1, H, c->p0b, ACT_GELU_ERF, nullptr, 0, nullptr, 0, 0, st));
    HIPCHK(c, tile_gemm(c->v_p1, H, prow, c->p2w, H, H, H, (bf16*)out_embeds, H, c->p2b, ACT_NONE, nullptr, 0, nullptr, 0, 0, st));
    return 0;
}
extern "C" int aha_vit_encode_pooled_first(aha_ctx* c, const uint8_t* frames, int n, int pooled, void* out_embeds, aha_hip_stream st) {
    if (pooled <= 0) return c ? fail(c, AHA_E_RANGE, "pooled grid must be in 1..patch grid") : AHA_E_INVAL;
    return aha_vit_encode_live(c, frames, n, pooled, 0, out_embeds, st);
}

extern "C" int aha_vit_last_tower_output(aha_ctx* c, int n_frames, void* out, aha_hip_stream st) {
    if (!c || !out || n_frames <= 0 || n_frames > c->d.max_vit_frames) return AHA_E_INVAL;
    // rows per frame: Np (SigLIP) or Np + 1 with the class token as the LAST row (CLIP)
    ORDER_VIT(c, (hipStream_t)st);
    HIPCHK(c, hipMemcpyAsync(out, c->v_x, (size_t)n_frames * c->Tt * c->d.v_hidden * 2, hipMemcpyDeviceToDevice, (hipStream_t)st));
    return 0;
}

extern "C" int aha_embed_tokens(aha_ctx* c, const int64_t* ids, int n, void* out, aha_hip_stream st) {
    if (!c || !ids || !out) return AHA_E_INVAL;
    if (!c->embed) return fail(c, AHA_E_NOENT, "model.embed_tokens.weight was not loaded");
    HIPCHK(c, aha_embed_gather((const long*)ids, n, c->embed, c->d.hidden, c->d.vocab, (bf16*)out, c->d.hidden, (hipStream_t)st));
    return 0;
}

// --------------------------------------------------------------------------------------------
// LM step
// --------------------------------------------------------------------------------------------
static int pick_split(aha_ctx* c, int kind, const PackedW& w, int M, int nt_per_wave) {
    (void)M;                                  // S must NOT depend on M: a batched step stays bit-identical to solo steps
    const int nc = w.KS / 8;                  // slices are placed in units of 8 k-steps (gemm_ws.hip), independent of KC
    int S = c->split[kind];
    if (S <= 0) {
        // ~2 four-wave workgroups per CU.  (One per CU - O 8 -> 4, QKV 7 -> 3, fewer slabs for the reducing kernels - measured
        // ~1 % faster on the single-stream step but 3-5 % slower on the batched shapes, which run 8-wave workgroups and were left
        // under-filled; S may not depend on M, so the batched-friendly value stays.)
        const int nblk = ceil_div(w.n_tiles, c->wpb[kind] * nt_per_wave);
        S = 512 / (nblk > 0 ? nblk : 1);
        if (S > 8) S = 8;
    }
    if (S > nc) S = nc;
    if (S > 16) S = 16;
    if (S < 1) S = 1;
    return S;
}

static GemmWsArgs ws_args(const bf16* X, int ldx, int M, int m0, int mrows, const PackedW& w, int S, float* partial, int ldp, bf16* out,
                          int ldo, float* outf, int ldof) {
    GemmWsArgs a;
    memset(&a, 0, sizeof(a));
    a.X = X + (long)m0 * ldx; a.ldx = ldx; a.M = mrows;
    a.Wp = w.p; a.KS = w.KS; a.Kx = w.K; a.n_tiles = w.n_tiles; a.S = S;
    a.partial = partial ? partial + (long)m0 * ldp : nullptr; a.ldp = ldp; a.slab_stride = (long)M * ldp;
    a.out = out ? out + (long)m0 * ldo : nullptr; a.ldo = ldo;
    a.outf = outf ? outf + (long)m0 * ldof : nullptr; a.ldof = ldof;
    a.bias = nullptr; a.N = w.N;
    return a;
}

// Row chunking and kernel choice of the weight-streaming GEMMs.  M <= 128: gemm_ws (weights in registers).  Above that the
// mid-M kernel (gemm_wl.hip: both operands through LDS-DMA stages) takes chunks of up to 320 rows; both kernels sum every
// output element's k-steps in the same order with the same split-K slices, so the choice never changes a bit.
static int ws_row_chunk(const aha_ctx* c, int epi, int M, int K) {
    const bool wl_ok = c->use_wl && M > 128 && (epi == EPI_PARTIAL || epi == EPI_SWIGLU) && K % 32 == 0;
    if (!wl_ok) return aha_gemm_ws_max_m(epi);
    // even chunks of whole row tiles, so that every chunk of an M > 128 step stays in the mid-M kernel's range (129..320)
    const int n = ceil_div(M, 320);
    return round_up(ceil_div(M, n), 16);
}
// Every row chunk of this GEMM runs gemm_wl (what a k-blocked operand layout needs: gemm_ws reads row-major X only).
static bool ws_all_wl(const aha_ctx* c, int epi, int M, int K) {
    if (!(c->use_wl && M > 128 && (epi == EPI_PARTIAL || epi == EPI_SWIGLU) && K % 32 == 0)) return false;
    const int mmax = ws_row_chunk(c, epi, M, K);
    for (int m0 = 0; m0 < M; m0 += mmax) {
        const int rows = (M - m0 < mmax) ? M - m0 : mmax;
        if (rows <= 128 || rows > 320) return false;
    }
    return true;
}
static hipError_t ws_or_wl(const aha_ctx* c, const GemmWsArgs* a, int epi, int wpb, hipStream_t st) {
    if (c->use_wl && aha_gemm_wl_supports(a, epi)) return aha_gemm_wl(a, epi, st);
    return aha_gemm_ws(a, epi, wpb, st);
}

// HIP-event bracket of one timed launch group (tuning "time_gemm": bit k = kind k), on the launch stream
static int timed_begin(aha_ctx* c, int kind, hipStream_t st) {
    if ((int)c->ev[kind].size() <= c->ev_used[kind]) {
        hipEvent_t a, b;
        HIPCHK(c, hipEventCreate(&a));
        HIPCHK(c, hipEventCreate(&b));
        c->ev[kind].push_back({a, b});
    }
    HIPCHK(c, hipEventRecord(c->ev[kind][c->ev_used[kind]].first, st));
    return 0;
}
static int timed_end(aha_ctx* c, int kind, double bytes, hipStream_t st) {
    HIPCHK(c, hipEventRecord(c->ev[kind][c->ev_used[kind]].second, st));
    c->ev_used[kind]++;
    c->gk_bytes[kind] += bytes;
    return 0;
}

static int ws_gemm(aha_ctx* c, int kind, const bf16* X, int ldx, int M, const PackedW& w, int epi, int S, float* partial, int ldp,
                   bf16* out, int ldo, float* outf, int ldof, hipStream_t st, int kb = 0) {
    // kb bit 0: X is k-blocked ([K/32][M][32], gemm_wl.hip); bit 1: the SwiGLU output is written k-blocked.  Callers set them
    // only when ws_all_wl() holds for the GEMMs on both sides of the buffer.
    const int mmax = ws_row_chunk(c, epi, M, w.K);
    const bool timed = kind >= 0 && ((c->time_gemm >> kind) & 1);      // time_gemm: bit k = GEMM kind k
    if (timed) { if (int rc = timed_begin(c, kind, st)) return rc; }
    for (int m0 = 0; m0 < M; m0 += mmax) {
        GemmWsArgs a = ws_args(X, ldx, M, m0, (M - m0 < mmax) ? M - m0 : mmax, w, S, partial, ldp, out, ldo, outf, ldof);
        if (kb & 1) { a.X = X + (long)m0 * 32; a.xkb = M; }
        if (kb & 2) { a.out = out + (long)m0 * 32; a.okb = M; }
        HIPCHK(c, ws_or_wl(c, &a, epi, kind >= 0 ? c->wpb[kind] : 4, st));
    }
    if (timed) { if (int rc = timed_end(c, kind, w.bytes() * ceil_div(M, mmax), st)) return rc; }
    c->last_weight_bytes += w.bytes();
    c->last_flops += 2.0 * (double)w.n_tiles * 16.0 * (double)w.K * (double)M;
    return 0;
}

extern "C" int aha_lm_step(aha_ctx* c, aha_stream* const* streams, int B, const void* embeds, int T, float* out_scores,
                           float* out_raw, void* out_last_hidden, aha_hip_stream st_) {
    if (!c || !streams || !embeds) return AHA_E_INVAL;
    if (!c->weights_loaded) return fail(c, AHA_E_INVAL, "weights not loaded");
    if (!c->rope_cos) return fail(c, AHA_E_INVAL, "rope table not set");
    if (B <= 0 || B > AHA_MAX_B) return fail(c, AHA_E_RANGE, "B out of range (1..16)");
    if (T <= 0 || B * T > c->d.max_step_tokens) return fail(c, AHA_E_RANGE, "B*T > max_step_tokens");
    hipStream_t st = (hipStream_t)st_;
    const aha_model_desc& d = c->d;
    const int H = d.hidden, Dh = d.head_dim, QD = d.heads * Dh, I = d.inter, M = B * T, G = d.heads / d.kv_heads;

    // ---- plan (host bookkeeping only; validate everything before mutating any stream)
    for (int b = 0; b < B; ++b) {
        aha_stream* s = streams[b];
        if (!s || s->ctx != c) return fail(c, AHA_E_INVAL, "bad stream handle");
        if (s->poisoned) return fail(c, AHA_E_INVAL, "stream state is undefined after a failed step (keys were re-rotated / slots overwritten): call aha_stream_reset");
        for (int b2 = 0; b2 < b; ++b2)
            if (streams[b2] == s && !(s->policy == AHA_CACHE_STATIC && s->len > 0))
                return fail(c, AHA_E_INVAL, "a stream may appear only once per step (except a frozen TrulyStaticCache stream, whose "
                                            "step neither reads nor writes per-step state: its frames are independent)");
    }
    StepDesc sd;
    memset(&sd, 0, sizeof(sd));
    sd.B = B;
    sd.T = T;
    // Host bookkeeping is advanced by plan_stream BEFORE any device work is enqueued; this guard puts every stream back
    // if anything fails before the first destructive launch (planning, descriptor upload), so a caller may retry.  Once
    // device work that changes the caches has been enqueued (in-place re-rotation of kept keys, ring slots overwritten by
    // the K/V append) a retry would rotate the kept keys a second time: the streams are then marked poisoned instead of
    // rolled back and refuse further steps until aha_stream_reset.
    struct Rollback {
        aha_stream* const* streams; int n = 0; int saved[AHA_MAX_B][3]; bool armed = true, destructive = false;
        ~Rollback() {
            if (!armed) return;
            for (int b = n - 1; b >= 0; --b) {
                if (destructive) { streams[b]->poisoned = true; continue; }
                streams[b]->len = saved[b][0]; streams[b]->head = saved[b][1]; streams[b]->seen = saved[b][2];
            }
        }
    } guard{streams};
    for (int b = 0; b < B; ++b) {
        aha_stream* s = streams[b];
        guard.saved[b][0] = s->len; guard.saved[b][1] = s->head; guard.saved[b][2] = s->seen;
        guard.n = b + 1;
        int rc = plan_stream(c, s, T, &sd.s[b]);
        if (!rc && sd.s[b].pos_base + T > c->n_pos) rc = fail(c, AHA_E_RANGE, "position exceeds the RoPE table");
        if (rc) return rc;
    }
    ORDER_LM(c, st);
    // SinkCache re-rotation (test/sink_cache.py:35-55): sink_rerotate_kernel computes the coefficients from the RoPE table on the
    // fly unless the caller registered a (window, n_sink, T) table (aha_ctx_set_rerotation_table) - nothing is allocated or built
    // inside a per-frame call.  The rows it reads are RoPE positions sink .. window - 1.
    for (int b = 0; b < B; ++b)
        if (sd.s[b].n_rerot > 0 && streams[b]->W > c->n_pos) return fail(c, AHA_E_RANGE, "SinkCache window exceeds the RoPE table");
    c->last_weight_bytes = c->last_kv_bytes = c->last_flops = 0;
    for (int k = 0; k < GK_COUNT; ++k) { c->ev_used[k] = 0; c->gk_bytes[k] = 0; }

    // ---- SinkCache re-rotation of kept keys (all layers, one launch per distinct table)
    int max_lk = 0;
    for (int b = 0; b < B; ++b) {
        max_lk = sd.s[b].len_after > max_lk ? sd.s[b].len_after : max_lk;
        c->last_kv_bytes += (double)sd.s[b].len_after * d.layers * d.kv_heads * Dh * 2 * 2;
    }
    {
        // Upload this step's descriptor (pinned ring slot -> the one device copy; stream order keeps the previous step's
        // kernels ahead of the overwrite).  A slot is rewritten only after the upload that last read it has completed:
        // an event recorded behind each upload is waited on before reuse - free when the caller synchronises every step,
        // and a real wait only for a caller that runs more than SD_SLOTS steps ahead of the GPU.
        const int si = c->sd_slot;
        StepDesc* slot = c->sd_pin + si;
        c->sd_slot = (si + 1) % aha_ctx::SD_SLOTS;
        if (c->sd_ev[si]) HIPCHK(c, hipEventSynchronize(c->sd_ev[si]));
        else HIPCHK(c, hipEventCreateWithFlags(&c->sd_ev[si], hipEventDisableTiming));
        *slot = sd;
        HIPCHK(c, hipMemcpyAsync(c->sd_dev, slot, sizeof(StepDesc), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipEventRecord(c->sd_ev[si], st));
        // streams sharing (W, sink) share the table; one launch per group, selected by a stream mask
        bool done[AHA_MAX_B] = {false};
        for (int b = 0; b < B; ++b) {
            if (done[b] || sd.s[b].n_rerot == 0) continue;
            unsigned mask = 0;
            int nmax = 0;
            for (int b2 = 0; b2 < B; ++b2) {
                const bool same = sd.s[b2].n_rerot > 0 && streams[b2]->W == streams[b]->W && streams[b2]->sink == streams[b]->sink;
                if (same) {
                    done[b2] = true;
                    mask |= 1u << b2;
                    nmax = sd.s[b2].n_rerot > nmax ? sd.s[b2].n_rerot : nmax;
                }
            }
            std::pair<bf16*, bf16*> tb{nullptr, nullptr};
            if (auto it = c->rerot.find(std::make_tuple(streams[b]->W, streams[b]->sink, T)); it != c->rerot.end()) tb = it->second;
            const bool timed = (c->time_gemm >> GK_REROT) & 1;
            if (timed) { if (int rc = timed_begin(c, GK_REROT, st)) return rc; }
            guard.destructive = true;
            HIPCHK(c, aha_sink_rerotate(c->sd_dev, mask, B, nmax, tb.first, tb.second, c->rope_cos, c->rope_sin, d.layers, d.kv_heads, Dh, st));
            if (timed) {
                double by = 0;                               // algorithmic: every kept key of every layer read and written once
                for (int b2 = 0; b2 < B; ++b2)
                    if ((mask >> b2) & 1u) by += (double)sd.s[b2].n_rerot * d.layers * d.kv_heads * Dh * 2.0 * 2.0;
                if (int rc = timed_end(c, GK_REROT, by, st)) return rc;
            }
        }
    }

    // attention geometry (launch shape only: the kernels read the key counts from the device descriptor)
    int split_len = c->attn_split_len > 0 ? round_up(c->attn_split_len, 64) : 256;
    int n_splits = ceil_div(max_lk, split_len);
    if (n_splits > 16) { split_len = round_up(ceil_div(max_lk, 16), 64); n_splits = ceil_div(max_lk, split_len); }
    if (n_splits < 1) n_splits = 1;

    // Every stream of the step a frozen TrulyStaticCache: the new tokens' K/V are neither stored nor read
    // (test/static_cache.py:33-36).  By default the full q|k|v projection still runs, as in the reference (its K/V columns
    // are dead work, but the headline streams what the reference streams).  Experiments: fuse_static = 2 projects only the
    // q tiles (they come first in the packed weight; same split, bit-identical q; -0.7 % step time); fuse_static = 1 also
    // builds Q inside the attention kernel instead of launching qkv_finish (measured slower).
    bool all_static_frozen = true;
    for (int b = 0; b < B; ++b) all_static_frozen = all_static_frozen && sd.s[b].write_base < 0;
    const bool q_only = all_static_frozen && c->fuse_static != 0;
    const bool frozen_all = all_static_frozen && c->fuse_static == 1;
    // ... and with a short frozen prefix (configs[1]: the 20-token query turn) qkv_finish and the attention are one launch on the
    // vector ALUs (elementwise.hip: qkv_finish_attn_static_kernel; tuning "static_attn", on by default)
    const bool static_attn = all_static_frozen && !frozen_all && c->static_attn && max_lk <= 64 && G <= 8;

    // Everything from the first RMSNorm to the heads, on stream `st`, scores to `scores_out`: run directly, or recorded
    // into a HIP graph (below).
    int l_first = 0, l_end = d.layers;
    if (c->layer_count > 0) {
        l_first = c->layer_first < 0 ? 0 : (c->layer_first >= d.layers ? d.layers - 1 : c->layer_first);
        l_end = l_first + c->layer_count > d.layers ? d.layers : l_first + c->layer_count;
    }
    // layout of the SwiGLU activation this step leaves in c->act (a function of M and the tunings only, so a replayed graph
    // agrees with it): k-blocked when both MLP GEMMs run the mid-M kernel on every row chunk (the fused MLP block needs M <= 64)
    c->act_kb_rows = (c->act_kb && I % 32 == 0 && ws_all_wl(c, EPI_SWIGLU, M, H) && ws_all_wl(c, EPI_PARTIAL, M, I)) ? M : 0;
    auto layers_and_heads = [&](hipStream_t st, float* scores_out) -> int {
        // ---- first RMSNorm (the residual stream c->h already holds the embeddings)
        HIPCHK(c, aha_rmsnorm(c->h, H, c->L[l_first].ln1, c->xn, H, M, H, d.rms_eps, st));

        int rc;
        for (int l = l_first; l < l_end; ++l) {
            const LayerW& w = c->L[l];
            // QKV projection -> split-K slabs (q tiles only for an all-frozen step, see above)
            PackedW wq = w.qkv;
            if (q_only) { wq.n_tiles = QD / 16; wq.N = QD; }
            const int nq_ld = w.qkv.n_tiles * 16;
            const int Sq = pick_split(c, GK_QKV, w.qkv, M, 1);      // same split as the full projection: bit-identical q
            if ((rc = ws_gemm(c, GK_QKV, c->xn, H, M, wq, EPI_PARTIAL, Sq, c->partial, nq_ld, nullptr, 0, nullptr, 0, st))) return rc;
            AttnArgs a;
            memset(&a, 0, sizeof(a));
            if (static_attn) {
                QkvFinishArgs qa;
                memset(&qa, 0, sizeof(qa));
                qa.partial = c->partial; qa.S = Sq; qa.slab_stride = (long)M * nq_ld; qa.ldp = nq_ld; qa.bias = w.qkv_bias;
                qa.rope_cos = c->rope_cos; qa.rope_sin = c->rope_sin; qa.n_pos = c->n_pos;
                qa.q_rot = c->q_rot; qa.ldq = QD; qa.Hq = d.heads; qa.Hkv = d.kv_heads; qa.D = Dh; qa.layer = l;
                const bool t_attn = (c->time_gemm >> GK_ATTN) & 1;
                if (t_attn) { if ((rc = timed_begin(c, GK_ATTN, st))) return rc; }
                HIPCHK(c, aha_qkv_finish_attn_static(&qa, c->sd_dev, M, T, c->attn_out, QD, 1.0f / sqrtf((float)Dh), st));
                if (t_attn) {
                    double by = 0;
                    for (int b = 0; b < B; ++b) by += (double)sd.s[b].len_after * d.kv_heads * Dh * 2.0 * 2.0;
                    if ((rc = timed_end(c, GK_ATTN, by, st))) return rc;
                }
            } else {
            if (!frozen_all) {
                QkvFinishArgs qa;
                memset(&qa, 0, sizeof(qa));
                qa.partial = c->partial; qa.S = Sq; qa.slab_stride = (long)M * nq_ld; qa.ldp = nq_ld; qa.bias = w.qkv_bias;
                qa.rope_cos = c->rope_cos; qa.rope_sin = c->rope_sin; qa.n_pos = c->n_pos;
                qa.q_rot = c->q_rot; qa.ldq = QD; qa.Hq = d.heads; qa.Hkv = d.kv_heads; qa.D = Dh; qa.layer = l;
                HIPCHK(c, aha_qkv_finish(&qa, c->sd_dev, M, st));
            } else {
                a.q_partial = c->partial; a.q_S = Sq; a.q_slab_stride = (long)M * nq_ld; a.q_ldp = nq_ld; a.q_bias = w.qkv_bias;
                a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.n_pos = c->n_pos;
            }
            // attention over the stream caches
            a.q = c->q_rot; a.q_bs = (long)T * QD; a.ldq = QD;
            a.out = c->attn_out; a.o_bs = (long)T * QD; a.ldo = QD;
            a.part_o = c->part_o; a.part_ml = c->part_ml;
            a.T = T; a.G = G; a.Hkv = d.kv_heads; a.split_len = split_len; a.n_splits = n_splits;
            a.scale = 1.0f / sqrtf((float)Dh); a.layer = l;
            const bool t_attn = (c->time_gemm >> GK_ATTN) & 1;
            if (t_attn) { if ((rc = timed_begin(c, GK_ATTN, st))) return rc; }
            HIPCHK(c, aha_attention(&a, c->sd_dev, B, Dh, st));
            if (t_attn) {                                    // algorithmic: this layer's K and V of every stream read once
                double by = 0;
                for (int b = 0; b < B; ++b) by += (double)sd.s[b].len_after * d.kv_heads * Dh * 2.0 * 2.0;
                if ((rc = timed_end(c, GK_ATTN, by, st))) return rc;
            }
            }
            // o_proj -> slabs ; reduce + residual + post-attention RMSNorm
            const int So = pick_split(c, GK_O, w.o, M, 1);
            if ((rc = ws_gemm(c, GK_O, c->attn_out, QD, M, w.o, EPI_PARTIAL, So, c->partial, H, nullptr, 0, nullptr, 0, st))) return rc;
            ResidNormArgs ra;
            memset(&ra, 0, sizeof(ra));
            ra.partial = c->partial; ra.S = So; ra.slab_stride = (long)M * H; ra.ldp = H;
            ra.h = c->h; ra.ldh = H; ra.w = w.ln2; ra.xn = c->xn; ra.ldx = H; ra.H = H; ra.eps = d.rms_eps;
            const int Sd = pick_split(c, GK_DOWN, w.down, M, 1);
            // One launch for resid_norm + gate/up + down (lm_fused.hip) when the step is a single small row block and both GEMM
            // phases fit one workgroup per CU; otherwise (and while a GEMM kind is being timed) three launches.  c->partial is
            // shared safely: o_proj's slabs are read in phase A, down's are written in phase C, two grid barriers later.
            const int gu_blocks = ceil_div(w.gateup.n_tiles, 16), dn_blocks = ceil_div(w.down.n_tiles, 8) * Sd;
            const bool fuse = c->fuse_mlp && M <= 64 && !c->time_gemm && c->wpb[GK_GATEUP] == 8 && c->wpb[GK_DOWN] == 8 &&
                              gu_blocks <= c->n_cus && dn_blocks <= c->n_cus && M <= c->n_cus;
            if (fuse) {
                MlpBlockArgs mb;
                memset(&mb, 0, sizeof(mb));
                mb.rn = ra; mb.M = M;
                mb.gu = ws_args(c->xn, H, M, 0, M, w.gateup, 1, nullptr, 0, c->act, I, nullptr, 0);
                mb.dn = ws_args(c->act, I, M, 0, M, w.down, Sd, c->partial, H, nullptr, 0, nullptr, 0);
                mb.ctr = c->bar_ctr; mb.base = c->bar_base; mb.err = c->bar_err; mb.sc1 = c->fuse_mlp >= 2;
                const int grid = c->n_cus < 256 ? c->n_cus : 256;
                HIPCHK(c, aha_lm_mlp_block(&mb, grid, st));
                c->bar_base += (unsigned long long)aha_lm_mlp_block_counter_step(grid);
                c->last_weight_bytes += w.gateup.bytes() + w.down.bytes();
                c->last_flops += 2.0 * 16.0 * M * ((double)w.gateup.n_tiles * w.gateup.K + (double)w.down.n_tiles * w.down.K);
            } else {
                // Between mid-M kernels the operands travel k-blocked ([K/32][M][32]): the consumer's LDS-DMA then pulls
                // contiguous 1-KiB panels instead of 16 half cache lines per instruction (-16 % on down at M = 288; same bits).
                // Here: the normed input of gate/up (xkb) and the SwiGLU activation for down_proj (akb).
                const bool akb = c->act_kb_rows != 0, xkb = akb && c->act_kb >= 2 && H % 32 == 0;
                ra.xkb = xkb ? M : 0;
                HIPCHK(c, aha_resid_norm(&ra, M, st));
                ra.xkb = 0;
                // gate/up with fused SwiGLU epilogue
                if ((rc = ws_gemm(c, GK_GATEUP, c->xn, H, M, w.gateup, EPI_SWIGLU, 1, nullptr, 0, c->act, I, nullptr, 0, st, (akb ? 2 : 0) | (xkb ? 1 : 0)))) return rc;
                // down_proj -> slabs ; reduce + residual + next RMSNorm (next layer's input norm or model.norm)
                if ((rc = ws_gemm(c, GK_DOWN, c->act, I, M, w.down, EPI_PARTIAL, Sd, c->partial, H, nullptr, 0, nullptr, 0, st, akb ? 1 : 0))) return rc;
            }
            ra.S = Sd;
            ra.w = (l + 1 < d.layers) ? c->L[l + 1].ln1 : c->final_norm;
            HIPCHK(c, aha_resid_norm(&ra, M, st));
        }
        // ---- heads on the last token of every stream
        if (scores_out || out_raw) HIPCHK(c, aha_heads(c->xn, H, T - 1, T, B, c->heads_w, H, scores_out, out_raw, c->bar_err, st));
        if (out_last_hidden)
            HIPCHK(c, hipMemcpy2DAsync(out_last_hidden, (size_t)H * 2, c->xn + (size_t)(T - 1) * H, (size_t)T * H * 2, (size_t)H * 2, B,
                                       hipMemcpyDeviceToDevice, st));

        return 0;
    };

    // ---- residual stream <- embeds
    HIPCHK(c, hipMemcpyAsync(c->h, embeds, (size_t)M * H * 2, hipMemcpyDeviceToDevice, st));

    // ---- HIP-graph replay.  A step is ~230 launches; the host needs ~9 us per launch (2.8 ms per step, 88 % of the GPU time
    // of a static step) and falls behind the GPU in the run of short kernels.  The per-step stream state lives in the device
    // descriptor, so the recorded launches depend only on the launch geometry: batch, tokens, key-split shape, the
    // frozen-static flags and the tuning epoch.  One captured graph therefore serves every step of that shape - any cache
    // policy, any stream - and is replayed; it is captured on a private stream the second time a shape is seen (every lazily
    // set kernel attribute has been set by then); any failure falls back to direct launches for that shape.  The descriptor
    // upload, the sink re-rotation and the input / score copies stay outside the graph.
    // from here on the launches overwrite ring slots; where a stream evicts this step, a failure can no longer be rolled back
    for (int b = 0; b < B; ++b)
        if (sd.s[b].write_base >= 0 && guard.saved[b][0] + sd.s[b].write_count > sd.s[b].len_after) guard.destructive = true;
    const double attn_flops = 4.0 * T * (double)max_lk * QD * B * d.layers;
    const int gflags = (q_only ? 1 : 0) | (frozen_all ? 2 : 0) | (static_attn ? 4 : 0);
    // (While GEMM launches are being timed the step is launched directly: a plain hipEventRecord issued during stream capture
    // does not become a graph node, so a replay would leave the events holding stale timestamps.)
    if (c->use_graph && out_scores && !out_raw && !out_last_hidden && !c->fuse_mlp && !c->time_gemm) {
        aha_ctx::GraphEntry* ge = nullptr;
        for (auto& g : c->graphs)
            if (g.B == B && g.T == T && g.epoch == c->tune_epoch && g.n_splits == n_splits && g.split_len == split_len && g.flags == gflags) {
                ge = &g;
                break;
            }
        if (!ge) {
            if (c->graphs.size() >= 64) {                    // more shapes than a 32k-token growing cache sweeps through (~44)
                // drop the oldest shape; its executable may in principle still be queued, so it is only retired here and
                // destroyed behind a device synchronisation once a few have piled up (rare), or with the context
                if (c->graphs.front().exec) c->retired_graphs.push_back(c->graphs.front().exec);
                c->graphs.erase(c->graphs.begin());
                if (c->retired_graphs.size() >= 16) {
                    HIPCHK(c, hipDeviceSynchronize());
                    for (auto e : c->retired_graphs) hipGraphExecDestroy(e);
                    c->retired_graphs.clear();
                }
            }
            c->graphs.emplace_back();
            ge = &c->graphs.back();
            ge->B = B; ge->T = T; ge->epoch = c->tune_epoch; ge->n_splits = n_splits; ge->split_len = split_len; ge->flags = gflags;
        }
        if (!ge->exec && !ge->failed && ge->seen >= 1) {
            hipGraph_t graph = nullptr;
            bool ok = hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (ok) {
                const int brc = layers_and_heads(c->cap_stream, c->graph_scores);
                const hipError_t e = hipStreamEndCapture(c->cap_stream, &graph);
                ok = brc == 0 && e == hipSuccess && graph != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&ge->exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) hipGraphDestroy(graph);
            if (!ok) { ge->exec = nullptr; ge->failed = true; (void)hipGetLastError(); }
            ge->wb = c->last_weight_bytes; ge->fl = c->last_flops;       // what the recorded launches stream / compute
            for (int k = 0; k < GK_COUNT; ++k) { ge->ev_used[k] = c->ev_used[k]; ge->gk_bytes[k] = c->gk_bytes[k]; }
            c->last_weight_bytes = c->last_flops = 0;                    // the capture executed nothing
            for (int k = 0; k < GK_COUNT; ++k) { c->ev_used[k] = 0; c->gk_bytes[k] = 0; }
        }
        ge->seen++;
        if (ge->exec) {
            HIPCHK(c, hipGraphLaunch(ge->exec, st));
            HIPCHK(c, hipMemcpyAsync(out_scores, c->graph_scores, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
            c->last_weight_bytes = ge->wb; c->last_flops = ge->fl + attn_flops;
            for (int k = 0; k < GK_COUNT; ++k) { c->ev_used[k] = 0; c->gk_bytes[k] = 0; }   // a replay records no GEMM events (time_gemm steps are launched directly)
            c->last_B = B;
            c->last_T = T;
            guard.armed = false;
            return 0;
        }
    }
    if (const int brc = layers_and_heads(st, out_scores)) return brc;
    c->last_flops += attn_flops;
    c->last_B = B;
    c->last_T = T;
    guard.armed = false;
    return 0;
}

extern "C" int aha_lm_heads_all(aha_ctx* c, float* out_raw, aha_hip_stream st) {
    if (!c || !out_raw || c->last_B == 0) return AHA_E_INVAL;
    ORDER_LM(c, (hipStream_t)st);
    HIPCHK(c, aha_heads(c->xn, c->d.hidden, 0, 1, c->last_B * c->last_T, c->heads_w, c->d.hidden, nullptr, out_raw, c->bar_err, (hipStream_t)st));
    return 0;
}

extern "C" int aha_lm_last_hidden_all(aha_ctx* c, void* out, aha_hip_stream st) {
    if (!c || !out || c->last_B == 0) return AHA_E_INVAL;
    ORDER_LM(c, (hipStream_t)st);
    HIPCHK(c, hipMemcpyAsync(out, c->xn, (size_t)c->last_B * c->last_T * c->d.hidden * 2, hipMemcpyDeviceToDevice, (hipStream_t)st));
    return 0;
}

extern "C" int aha_lm_logits_last(aha_ctx* c, float* logits, int64_t* argmax, aha_hip_stream st_) {
    if (!c || c->last_B == 0) return AHA_E_INVAL;
    if (!c->lm_head.p) return fail(c, AHA_E_NOENT, "lm_head.weight was not loaded");
    hipStream_t st = (hipStream_t)st_;
    ORDER_LM(c, st);
    const int B = c->last_B, T = c->last_T, H = c->d.hidden, V = c->d.vocab;
    float* lg = logits ? logits : c->logits;
    // last-token rows are strided by T*H in xn: ldx = T*H makes them the M = B rows of the GEMM
    int rc = ws_gemm(c, -1, c->xn + (size_t)(T - 1) * H, T * H, B, c->lm_head, EPI_F32_RBF, 1, nullptr, 0, nullptr, 0, lg, V, st);
    if (rc) return rc;
    if (argmax) HIPCHK(c, aha_argmax(lg, V, V, B, (long*)argmax, st));
    return 0;
}

// all-position lm_head of the last step: outputs.logits [B,T,V] of the reference forward
// (video_head_live_llava_qwen.py:175), fp32, row-chunked through the weight-streaming GEMM
extern "C" int aha_lm_logits_all(aha_ctx* c, float* logits, aha_hip_stream st_) {
    if (!c || !logits || c->last_B == 0) return AHA_E_INVAL;
    if (!c->lm_head.p) return fail(c, AHA_E_NOENT, "lm_head.weight was not loaded");
    hipStream_t st = (hipStream_t)st_;
    ORDER_LM(c, st);
    const int M = c->last_B * c->last_T, H = c->d.hidden, V = c->d.vocab;
    return ws_gemm(c, -1, c->xn, H, M, c->lm_head, EPI_F32_RBF, 1, nullptr, 0, nullptr, 0, logits, V, st);
}

// parity tap: copy a workspace of the last aha_lm_step.  which: 0 residual stream h [M][hidden] (= the hidden state after the
// last executed decoder layer), 1 xn [M][hidden] (h normalised for the next layer / by model.norm), 2 rotated queries
// [M][heads*head_dim], 3 attention output [M][heads*head_dim], 4 SwiGLU activation [M][inter]; the last three hold the LAST
// executed layer's values (tuning layer_first / layer_count select it).
extern "C" int aha_lm_debug_tap(aha_ctx* c, int which, void* out, aha_hip_stream st_) {
    if (!c || !out || c->last_B == 0) return AHA_E_INVAL;
    hipStream_t st = (hipStream_t)st_;
    ORDER_LM(c, st);
    const size_t M = (size_t)c->last_B * c->last_T;
    const void* src; size_t cols;
    switch (which) {
        case 0: src = c->h; cols = c->d.hidden; break;
        case 1: src = c->xn; cols = c->d.hidden; break;
        case 2: src = c->q_rot; cols = (size_t)c->d.heads * c->d.head_dim; break;
        case 3: src = c->attn_out; cols = (size_t)c->d.heads * c->d.head_dim; break;
        case 4:
            if (c->act_kb_rows) {                                  // the mid-M path leaves the activation k-blocked
                HIPCHK(c, aha_kblocked_to_rows(c->act, c->act_kb_rows, c->d.inter, (bf16*)out, c->d.inter, st));
                return 0;
            }
            src = c->act; cols = c->d.inter; break;
        default: return fail(c, AHA_E_INVAL, "unknown tap");
    }
    HIPCHK(c, hipMemcpyAsync(out, src, M * cols * 2, hipMemcpyDeviceToDevice, st));
    return 0;
}

// --------------------------------------------------------------------------------------------
// operator level
// --------------------------------------------------------------------------------------------
struct aha_linear { aha_ctx* ctx; int device; PackedW w; bool pairs; };

extern "C" int aha_linear_create(aha_ctx* c, const void* w, const void* w_up, int N, int K, aha_linear** out, aha_hip_stream st_) {
    if (!c || !w || !out || N <= 0 || K <= 0) return AHA_E_INVAL;
    if (K % 8) return fail(c, AHA_E_INVAL, "K must be a multiple of 8");
    hipStream_t st = (hipStream_t)st_;
    aha_linear* L = new aha_linear();
    L->ctx = c; L->device = c->device; L->pairs = w_up != nullptr;
    const int nt = ceil_div(N, 16);
    PackedW& pw = L->w;
    pw.n_tiles = L->pairs ? 2 * nt : nt; pw.K = K; pw.N = N;
    pw.KS = round_up(ceil_div(K, 32), 8);
    if (hipMalloc((void**)&pw.p, (size_t)pw.n_tiles * pw.KS * 1024) != hipSuccess) { delete L; return fail(c, AHA_E_NOMEM, "hipMalloc failed"); }
    hipError_t e = aha_pack_w((const bf16*)w, N, K, K, pw.p, pw.KS, L->pairs ? 2 : 1, 0, st);
    if (e == hipSuccess && L->pairs) e = aha_pack_w((const bf16*)w_up, N, K, K, pw.p, pw.KS, 2, 1, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);        // set-up call: the sources may be freed on return
    if (e != hipSuccess) { hipFree(pw.p); delete L; return fail(c, AHA_E_HIP, std::string("pack: ") + hipGetErrorString(e)); }
    *out = L;
    return 0;
}
extern "C" void aha_linear_destroy(aha_linear* L) {
    if (!L) return;
    hipSetDevice(L->device);
    hipDeviceSynchronize();
    hipFree(L->w.p);
    delete L;
}
extern "C" int aha_linear_split_k(aha_ctx* c, const aha_linear* L, int requested) {
    if (!c || !L) return AHA_E_INVAL;
    const int nc = L->w.KS / 8;
    int S = requested < 1 ? 1 : requested;
    if (S > nc) S = nc;
    if (S > 16) S = 16;
    return S;
}
extern "C" int aha_linear_forward(aha_ctx* c, const aha_linear* L, const void* x, int ldx, int M, int epilogue, int split_k, const void* bias,
                                  void* out, int ldo, aha_hip_stream st_) {
    if (!c || !L || !x || !out || M <= 0) return AHA_E_INVAL;
    if (L->pairs != (epilogue == EPI_SWIGLU)) return fail(c, AHA_E_INVAL, "the SwiGLU epilogue needs a gate/up pair weight (and only it)");
    if (epilogue < EPI_PARTIAL || epilogue > EPI_F32_RBF) return fail(c, AHA_E_INVAL, "unknown epilogue");
    if (bias && epilogue != EPI_BF16) return fail(c, AHA_E_INVAL, "bias is supported by the bf16 epilogue only");
    hipStream_t st = (hipStream_t)st_;
    const int S = epilogue == EPI_PARTIAL ? aha_linear_split_k(c, L, split_k) : 1;
    const int mmax = ws_row_chunk(c, epilogue, M, L->w.K);
    for (int m0 = 0; m0 < M; m0 += mmax) {
        GemmWsArgs a = ws_args((const bf16*)x, ldx, M, m0, (M - m0 < mmax) ? M - m0 : mmax, L->w, S, epilogue == EPI_PARTIAL ? (float*)out : nullptr,
                               ldo, epilogue == EPI_BF16 || epilogue == EPI_SWIGLU ? (bf16*)out : nullptr, ldo,
                               epilogue == EPI_F32_RBF ? (float*)out : nullptr, ldo);
        a.bias = (const bf16*)bias;
        if (c->dev_xkb) { a.xkb = ldx; a.ldx = 32; }
        HIPCHK(c, ws_or_wl(c, &a, epilogue, 4, st));
    }
    return 0;
}
extern "C" int aha_linear_tile_forward(aha_ctx* c, const void* x, int ldx, int M, const void* w, int ldw, int N, int K, const void* bias, int act,
                                       const void* residual, int ldr, void* out, int ldo, aha_hip_stream st_) {
    if (!c || !x || !w || !out || M <= 0 || N <= 0 || K <= 0) return AHA_E_INVAL;
    if (K % 8 || ldx % 8 || ldw % 8) return fail(c, AHA_E_INVAL, "K and the leading dimensions must be multiples of 8");
    if (act < ACT_NONE || act > ACT_QUICK_GELU) return fail(c, AHA_E_INVAL, "unknown activation");
    HIPCHK(c, tile_gemm((const bf16*)x, ldx, M, (const bf16*)w, ldw, N, K, (bf16*)out, ldo, (const bf16*)bias, act, (const bf16*)residual, ldr,
                        nullptr, 0, 0, (hipStream_t)st_));
    return 0;
}
extern "C" int aha_rmsnorm_forward(aha_ctx* c, const void* x, int ldx, const void* w, void* out, int ldo, int M, int H, float eps, aha_hip_stream st) {
    if (!c || !x || !w || !out) return AHA_E_INVAL;
    HIPCHK(c, aha_rmsnorm((const bf16*)x, ldx, (const bf16*)w, (bf16*)out, ldo, M, H, eps, (hipStream_t)st));
    return 0;
}
extern "C" int aha_resid_rmsnorm_forward(aha_ctx* c, const float* partial, int S, void* h, const void* w, void* xn, int M, int H, float eps,
                                         aha_hip_stream st) {
    if (!c || !partial || !h || !w || !xn || S < 1 || S > 16) return AHA_E_INVAL;
    ResidNormArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.partial = partial; ra.S = S; ra.slab_stride = (long)M * H; ra.ldp = H;
    ra.h = (bf16*)h; ra.ldh = H; ra.w = (const bf16*)w; ra.xn = (bf16*)xn; ra.ldx = H; ra.H = H; ra.eps = eps;
    HIPCHK(c, aha_resid_norm(&ra, M, (hipStream_t)st));
    return 0;
}
extern "C" int aha_heads_forward(aha_ctx* c, const void* hidden, int ld, int rows, float* scores, float* raw, aha_hip_stream st) {
    if (!c || !hidden || (!scores && !raw) || !c->heads_w) return AHA_E_INVAL;
    HIPCHK(c, aha_heads((const bf16*)hidden, ld, 0, 1, rows, c->heads_w, c->d.hidden, scores, raw, c->bar_err, (hipStream_t)st));
    return 0;
}

// upload a descriptor built outside aha_lm_step through the same fenced pinned ring
static int upload_desc(aha_ctx* c, const StepDesc& sd, hipStream_t st) {
    const int si = c->sd_slot;
    c->sd_slot = (si + 1) % aha_ctx::SD_SLOTS;
    if (c->sd_ev[si]) HIPCHK(c, hipEventSynchronize(c->sd_ev[si]));
    else HIPCHK(c, hipEventCreateWithFlags(&c->sd_ev[si], hipEventDisableTiming));
    c->sd_pin[si] = sd;
    HIPCHK(c, hipMemcpyAsync(c->sd_dev, c->sd_pin + si, sizeof(StepDesc), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->sd_ev[si], st));
    return 0;
}
static void describe_current(const aha_stream* s, StreamStep* o) {
    memset(o, 0, sizeof(*o));
    o->k_base = s->k; o->v_base = s->v; o->cap = s->cap; o->ring_head = s->head; o->len_after = s->len; o->write_base = -1;
    if (s->policy == AHA_CACHE_NONE || s->policy == AHA_CACHE_STATIC) { o->n_fixed = s->cap; o->ring_cap = 1; }
    else if (s->policy == AHA_CACHE_SLIDING) { o->n_fixed = 0; o->ring_cap = s->W; }
    else { o->n_fixed = s->sink; o->ring_cap = s->W - s->sink; }
}

// Operator-level attention: T query rows per stream (bf16 [B][T][heads*head_dim], already rotated) against the streams' caches
// AS THEY ARE (the rows' own K/V must already be in the cache, e.g. through aha_cache_update), layer `layer`.
// causal_off[b]: key j is visible to row t iff j <= causal_off[b] + t; null = the trailing rule (seq_length - T).
extern "C" int aha_attention_forward(aha_ctx* c, aha_stream* const* streams, int B, const void* q, int T, int layer, const int* causal_off,
                                     int split_len, void* out, aha_hip_stream st_) {
    if (!c || !streams || !q || !out || B <= 0 || B > AHA_MAX_B || T <= 0) return AHA_E_INVAL;
    const aha_model_desc& d = c->d;
    if (layer < 0 || layer >= d.layers) return fail(c, AHA_E_RANGE, "layer out of range");
    if (B * T > d.max_step_tokens) return fail(c, AHA_E_RANGE, "B*T > max_step_tokens");
    hipStream_t st = (hipStream_t)st_;
    ORDER_LM(c, st);
    StepDesc sd;
    memset(&sd, 0, sizeof(sd));
    sd.B = B; sd.T = T;
    int max_lk = 0;
    for (int b = 0; b < B; ++b) {
        if (!streams[b] || streams[b]->ctx != c) return fail(c, AHA_E_INVAL, "bad stream handle");
        if (streams[b]->len <= 0) return fail(c, AHA_E_INVAL, "empty cache");
        describe_current(streams[b], &sd.s[b]);
        sd.s[b].causal_off = causal_off ? causal_off[b] : streams[b]->len - T;
        max_lk = streams[b]->len > max_lk ? streams[b]->len : max_lk;
    }
    int rc = upload_desc(c, sd, st);
    if (rc) return rc;
    int sl = split_len > 0 ? round_up(split_len, 64) : (c->attn_split_len > 0 ? round_up(c->attn_split_len, 64) : 256);
    int ns = ceil_div(max_lk, sl);
    if (ns > 16) { sl = round_up(ceil_div(max_lk, 16), 64); ns = ceil_div(max_lk, sl); }
    const int QD = d.heads * d.head_dim;
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const bf16*)q; a.q_bs = (long)T * QD; a.ldq = QD;
    a.out = (bf16*)out; a.o_bs = (long)T * QD; a.ldo = QD;
    a.part_o = c->part_o; a.part_ml = c->part_ml;
    a.T = T; a.G = d.heads / d.kv_heads; a.Hkv = d.kv_heads; a.split_len = sl; a.n_splits = ns < 1 ? 1 : ns;
    a.scale = 1.0f / sqrtf((float)d.head_dim); a.layer = layer;
    HIPCHK(c, aha_attention(&a, c->sd_dev, B, d.head_dim, st));
    return 0;
}

// ---- vision operators (the tower's non-GEMM kernels on caller tensors; the tiled GEMM is aha_linear_tile_forward) ----------
// SiglipAttention / CLIPAttention core (transformers modeling_siglip.py:116-170): softmax(q k^T / sqrt(head_dim)) v per frame
// and head, non-causal.  qkv: bf16 [n][T][3*heads*head_dim] (q | k | v concatenated per row, as the tower's fused projection
// writes them); out: bf16 [n][T][heads*head_dim].
extern "C" int aha_vit_attention_forward(aha_ctx* c, const void* qkv, int n, int T, int heads, int head_dim, void* out, aha_hip_stream st_) {
    if (!c || !qkv || !out || n <= 0 || T <= 0 || heads <= 0) return AHA_E_INVAL;
    if (head_dim < 8 || head_dim > 128 || (head_dim & 7)) return fail(c, AHA_E_INVAL, "head_dim must be a multiple of 8, <= 128");
    const int Dv = heads * head_dim;
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const bf16*)qkv; a.q_bs = (long)T * 3 * Dv; a.ldq = 3 * Dv;
    a.k = a.q + Dv; a.v = a.q + 2 * Dv; a.kv_bs = (long)T * 3 * Dv; a.ldk = 3 * Dv;
    a.out = (bf16*)out; a.o_bs = (long)T * Dv; a.ldo = Dv;
    a.T = T; a.G = 1; a.Hkv = heads; a.Lk = T;
    a.split_len = round_up(T, 64); a.n_splits = 1;
    a.scale = 1.0f / sqrtf((float)head_dim);
    HIPCHK(c, aha_attention(&a, nullptr, n, head_dim, (hipStream_t)st_));
    return 0;
}
// Encoder layers [layer_first, layer_first + layer_count) of the vision tower on a caller-supplied hidden state:
// x bf16 [n * tokens_per_frame][v_hidden] -> out (same shape); tokens_per_frame = patches (+ 1 class token LAST for CLIP).
// The per-layer parity tests teacher-force each layer from the oracle's input to it.
extern "C" int aha_vit_layers_forward(aha_ctx* c, const void* x, int n, int layer_first, int layer_count, void* out, aha_hip_stream st_) {
    if (!c || !x || !out || n <= 0) return AHA_E_INVAL;
    if (!c->weights_loaded) return fail(c, AHA_E_INVAL, "weights not loaded");
    if (n > c->d.max_vit_frames) return fail(c, AHA_E_RANGE, "n_frames > max_vit_frames");
    if (layer_first < 0 || layer_count <= 0 || layer_first + layer_count > c->d.v_layers) return fail(c, AHA_E_RANGE, "vision layer range");
    hipStream_t st = (hipStream_t)st_;
    ORDER_VIT(c, st);
    const size_t bytes = (size_t)n * c->Tt * c->d.v_hidden * 2;
    HIPCHK(c, hipMemcpyAsync(c->v_x, x, bytes, hipMemcpyDeviceToDevice, st));
    if (int rc = vit_layers(c, n, layer_first, layer_first + layer_count, st)) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->v_x, bytes, hipMemcpyDeviceToDevice, st));
    return 0;
}
// nn.LayerNorm over the last dimension (fp32 statistics, bf16 in / out): x bf16 [rows][ldx] -> out bf16 [rows][ldo].
extern "C" int aha_layernorm_forward(aha_ctx* c, const void* x, int ldx, const void* w, const void* b, void* out, int ldo, int rows, int cols,
                                     float eps, aha_hip_stream st_) {
    if (!c || !x || !w || !b || !out || rows <= 0 || cols <= 0 || (cols & 7) || cols > 4096) return AHA_E_INVAL;
    HIPCHK(c, aha_layernorm((const bf16*)x, ldx, (const bf16*)w, (const bf16*)b, (bf16*)out, ldo, rows, cols, eps, (hipStream_t)st_));
    return 0;
}
// image_processor.preprocess + the unfold of the patch-embedding Conv2d (test/inference.py:176; kernel = stride = patch):
// uint8 [n][3][S][S] -> bf16 [n * Np][Kp], row = patch (row-major over the patch grid), column = c*P*P + y*P + x of the
// normalised pixel ((x/255 - mean) / std, rounded to bf16 once), columns >= 3*P*P zero.  *out_cols reports Kp.
extern "C" int aha_vit_patchify_forward(aha_ctx* c, const uint8_t* frames, int n, void* out, int* out_cols, aha_hip_stream st_) {
    if (!c || !frames || !out || n <= 0) return AHA_E_INVAL;
    if (out_cols) *out_cols = c->Kp;
    HIPCHK(c, aha_im2col_norm(frames, n, c->d.image_size, c->d.patch_size, c->Kp, c->px_mean, c->px_std, (bf16*)out, (hipStream_t)st_));
    return 0;
}
// Spatial pooling of a token grid (video_head_live_llava_qwen.py:117-136; models/vision_live.py:21-24): in bf16 [n][frame_rows][C]
// whose first grid*grid rows are the row-major patch grid -> out bf16 [n][out_grid^2][C].  mode 0: F.interpolate(bilinear,
// align_corners=False) to out_grid; 1 / 2: avg / max pool with kernel = stride; 3: adaptive_avg_pool2d to out_grid.
extern "C" int aha_pool_forward(aha_ctx* c, const void* in, int n, int grid, int out_grid, int C_, int stride, int mode, int frame_rows, void* out,
                                aha_hip_stream st_) {
    if (!c || !in || !out || n <= 0 || grid <= 0 || out_grid <= 0 || C_ <= 0 || (C_ & 3) || mode < 0 || mode > 3) return AHA_E_INVAL;
    if (frame_rows < grid * grid) return fail(c, AHA_E_RANGE, "frame_rows < grid^2");
    HIPCHK(c, aha_pool((const bf16*)in, (bf16*)out, n, grid, out_grid, C_, stride, mode, frame_rows, (hipStream_t)st_));
    return 0;
}
// The rows of the patch grid that bilinear pooling with an even integer stride samples (2*out_grid per side), compacted:
// in bf16 [n][frame_rows][C] -> out bf16 [n][(2*out_grid)^2][C] (the projector then runs on these rows only; aha_vit_encode).
extern "C" int aha_pool_gather_rows_forward(aha_ctx* c, const void* in, int n, int grid, int out_grid, int C_, int frame_rows, void* out,
                                            aha_hip_stream st_) {
    if (!c || !in || !out || n <= 0 || grid <= 0 || out_grid <= 0 || C_ <= 0 || (C_ & 7)) return AHA_E_INVAL;
    const int s = grid / out_grid;
    if (grid % out_grid || s < 4 || (s & 1)) return fail(c, AHA_E_RANGE, "needs an even integer stride >= 4");
    HIPCHK(c, aha_gather_pool_rows((const bf16*)in, (bf16*)out, n, grid, out_grid, s, C_, frame_rows, (hipStream_t)st_));
    return 0;
}

// Operator-level Cache.update(key_states, value_states, layer_idx, cache_kwargs) of the reference's cache classes
// (test/sink_cache.py:74-164, test/sliding_window_cache.py:17-44, test/static_cache.py:18-36): new K/V bf16 [kv_heads][T][head_dim]
// (already rotated).  Like the reference, layer 0's call advances the bookkeeping (seen tokens, eviction, positions) and the
// other layers of the same step must follow in order; each call re-rotates that layer's kept keys (SinkCache) and appends.
// out_k / out_v (optional): the (K, V) the reference's update() returns, bf16 [kv_heads][seq_length][head_dim] in logical order
// - for a frozen TrulyStaticCache the stored prefix only.
extern "C" int aha_cache_update(aha_ctx* c, aha_stream* s, int layer, const void* k_new, const void* v_new, int T, void* out_k, void* out_v,
                                aha_hip_stream st_) {
    if (!c || !s || s->ctx != c || !k_new || !v_new || T <= 0) return AHA_E_INVAL;
    const aha_model_desc& d = c->d;
    if (layer < 0 || layer >= d.layers) return fail(c, AHA_E_RANGE, "layer out of range");
    if (s->poisoned) return fail(c, AHA_E_INVAL, "stream state is undefined after a failed step: call aha_stream_reset");
    hipStream_t st = (hipStream_t)st_;
    ORDER_LM(c, st);
    struct { StreamStep& ss; int& T; int& next_layer; bool& valid; } op{s->op_ss, s->op_T, s->op_next_layer, s->op_valid};
    if (layer == 0) {
        const int sv[3] = {s->len, s->head, s->seen};
        int rc = plan_stream(c, s, T, &op.ss);
        if (!rc && op.ss.n_rerot > 0 && !c->rope_cos) rc = fail(c, AHA_E_INVAL, "rope table not set");
        if (!rc && op.ss.n_rerot > 0 && s->W > c->n_pos) rc = fail(c, AHA_E_RANGE, "SinkCache window exceeds the RoPE table");
        if (rc) { s->len = sv[0]; s->head = sv[1]; s->seen = sv[2]; op.valid = false; return rc; }
        op.T = T; op.valid = true; op.next_layer = 0;
    }
    if (!op.valid || op.T != T || layer != op.next_layer)
        return fail(c, AHA_E_INVAL, "aha_cache_update: layers of a step must be updated in order 0..L-1 with the same T");
    op.next_layer = layer + 1;
    const bf16 *rc_ = nullptr, *rs_ = nullptr;
    if (op.ss.n_rerot > 0)                                  // a registered table if there is one, else coefficients on the fly
        if (auto it = c->rerot.find(std::make_tuple(s->W, s->sink, T)); it != c->rerot.end()) { rc_ = it->second.first; rs_ = it->second.second; }
    if (hipError_t e = aha_cache_update_layer(&op.ss, layer, d.kv_heads, d.head_dim, T, (const bf16*)k_new, (const bf16*)v_new, rc_, rs_, c->rope_cos,
                                              c->rope_sin, st); e != hipSuccess) {
        // the step cannot be completed: layers already updated hold the new state, the rest the old one
        op.valid = false;
        s->poisoned = true;
        return fail(c, AHA_E_HIP, std::string("aha_cache_update_layer: ") + hipGetErrorString(e));
    }
    if (out_k) { int rc = aha_stream_export_kv(c, s, layer, 0, out_k, st_); if (rc) return rc; }
    if (out_v) { int rc = aha_stream_export_kv(c, s, layer, 1, out_v, st_); if (rc) return rc; }
    return 0;
}

// fast_greedy_generate (models/modeling_live.py:64-90) + the token hand-back of _generate_response (test/inference.py:264-281):
// greedy decode from `first_ids` (the stream-generation prompt) against the stream's cache, at most max_new_tokens single-token
// steps, stopping after EOS.  argmax -> embedding row -> next step stay on the device (the id never travels through the host on
// the data path); the host only polls the 8-byte id behind each step to honour the early stop exactly - this call therefore
// blocks until the response is complete.  repetition_penalty > 0: RepetitionPenaltyLogitsProcessor over `history` (device
// int64 [history_cap], *history_len entries; generated non-EOS ids are appended, as the reference's generated_token_ids list).
// Chunked use: a response can be produced in several calls - pass max_new_tokens = the chunk size, and continue with
// first_ids = the last id of the previous chunk (n_first = 1, a device pointer) until an EOS arrives or the response limit is
// reached; between chunks the caller may step other streams (the generation state is the stream's cache plus `history`).
// on_token (optional): called with every new id as soon as it is host-visible (index = position in this call); a non-zero return
// stops the generation after that token (the cache then holds exactly the tokens fed so far, as after an EOS stop).  The
// callback may call aha_lm_step / aha_vit_encode for OTHER streams on the same HIP stream; it must not start another generation
// on this context.  *out_count always reports the ids written to out_ids_host, also when the call fails part-way.
extern "C" int aha_generate_greedy_cb(aha_ctx* c, aha_stream* s, const int64_t* first_ids, int n_first, int max_new_tokens, int64_t eos_token_id,
                                      float repetition_penalty, int64_t* history, int history_cap, int* history_len, int64_t* out_ids_host,
                                      int* out_count, aha_token_cb on_token, void* user, aha_hip_stream st_) {
    if (out_count) *out_count = 0;
    if (!c || !s || !first_ids || n_first <= 0 || max_new_tokens <= 0 || !out_ids_host || !out_count) return AHA_E_INVAL;
    if (!c->embed) return fail(c, AHA_E_NOENT, "model.embed_tokens.weight was not loaded");
    if (!c->lm_head.p) return fail(c, AHA_E_NOENT, "lm_head.weight was not loaded");
    const bool pen = repetition_penalty > 0.f;
    if (pen && (!history || !history_len || history_cap <= 0 || *history_len < 0 || *history_len > history_cap))
        return fail(c, AHA_E_INVAL, "repetition penalty needs a history buffer");
    hipStream_t st = (hipStream_t)st_;
    const aha_model_desc& d = c->d;
    const int H = d.hidden, V = d.vocab;
    if (n_first > d.max_step_tokens) return fail(c, AHA_E_RANGE, "prompt longer than max_step_tokens");
    int rc;
    if (!c->gen_tok) {
        if ((rc = dalloc(c, &c->gen_tok, 1)) || (rc = dalloc(c, &c->gen_nhist, 1)) || (rc = dalloc(c, &c->gen_emb, (size_t)d.max_step_tokens * H))) return rc;
        if (hipHostMalloc((void**)&c->gen_pin, sizeof(long), hipHostMallocDefault) != hipSuccess) return fail(c, AHA_E_NOMEM, "hipHostMalloc failed");
        HIPCHK(c, hipEventCreateWithFlags(&c->gen_ev, hipEventDisableTiming));
    }
    const int need = max_new_tokens > history_cap ? max_new_tokens : history_cap;
    if (c->gen_cap < need) {
        // geometric growth (a response history grows by one response per call), the replaced buffers are released: the work
        // that used them was synchronised by the token polls of the call that enqueued it
        int cap = c->gen_cap > 0 ? c->gen_cap : 1024;
        while (cap < need) cap *= 2;
        long* nout = nullptr; float* ntmp = nullptr;
        if (hipMalloc((void**)&nout, (size_t)cap * sizeof(long)) != hipSuccess || hipMalloc((void**)&ntmp, (size_t)cap * sizeof(float)) != hipSuccess) {
            if (nout) hipFree(nout);
            return fail(c, AHA_E_NOMEM, "generation scratch allocation failed");
        }
        if (c->gen_out) { HIPCHK(c, hipStreamSynchronize(st)); hipFree(c->gen_out); hipFree(c->gen_tmp); }
        c->gen_out = nout; c->gen_tmp = ntmp; c->gen_cap = cap;
    }
    if (pen) HIPCHK(c, hipMemcpyAsync(c->gen_nhist, history_len, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(c, aha_embed_gather((const long*)first_ids, n_first, c->embed, H, V, c->gen_emb, H, st));
    aha_stream* one[1] = {s};
    int T = n_first;
    int& n = *out_count;                                      // kept current: an error return reports the ids already delivered
#define GEN_CHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(c, AHA_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
    for (int i = 0; i < max_new_tokens; ++i) {
        if ((rc = aha_lm_step(c, one, 1, c->gen_emb, T, c->graph_scores + 3 * (AHA_MAX_B - 1), nullptr, nullptr, st_))) return rc;
        // lm_head on the last position -> fp32 logits (bf16-rounded, as a bf16 nn.Linear hands them on), penalty, argmax
        if ((rc = ws_gemm(c, -1, c->xn + (size_t)(T - 1) * H, T * H, 1, c->lm_head, EPI_F32_RBF, 1, nullptr, 0, nullptr, 0, c->logits, V, st))) return rc;
        if (pen) GEN_CHK(aha_repetition_penalty(c->logits, V, (const long*)history, c->gen_nhist, repetition_penalty, c->gen_tmp, st));
        GEN_CHK(aha_argmax(c->logits, V, V, 1, c->gen_tok, st));
        GEN_CHK(aha_generation_bookkeep(c->gen_tok, (long)eos_token_id, (long*)history, c->gen_nhist, history_cap, pen ? 1 : 0, c->gen_out, i, st));
        GEN_CHK(aha_embed_gather(c->gen_tok, 1, c->embed, H, V, c->gen_emb, H, st));         // next step's input, no host in between
        GEN_CHK(hipMemcpyAsync(c->gen_pin, c->gen_tok, sizeof(long), hipMemcpyDeviceToHost, st));
        GEN_CHK(hipEventRecord(c->gen_ev, st));
        GEN_CHK(hipEventSynchronize(c->gen_ev));
        const long tok = *c->gen_pin;
        out_ids_host[n++] = tok;
        if (pen && tok != eos_token_id && *history_len < history_cap) ++*history_len;
        T = 1;
        if (tok == eos_token_id) break;
        if (on_token && on_token(user, (int64_t)tok, i)) break;
    }
#undef GEN_CHK
    return 0;
}

extern "C" int aha_generate_greedy(aha_ctx* c, aha_stream* s, const int64_t* first_ids, int n_first, int max_new_tokens, int64_t eos_token_id,
                                   float repetition_penalty, int64_t* history, int history_cap, int* history_len, int64_t* out_ids_host,
                                   int* out_count, aha_hip_stream st_) {
    return aha_generate_greedy_cb(c, s, first_ids, n_first, max_new_tokens, eos_token_id, repetition_penalty, history, history_cap, history_len,
                                  out_ids_host, out_count, nullptr, nullptr, st_);
}

extern "C" int aha_lm_last_step_work(aha_ctx* c, double* wb, double* kvb, double* fl) {
    if (!c) return AHA_E_INVAL;
    if (wb) *wb = c->last_weight_bytes;
    if (kvb) *kvb = c->last_kv_bytes;
    if (fl) *fl = c->last_flops;
    return 0;
}

extern "C" int aha_lm_last_gemm_time(aha_ctx* c, int kind, float* ms, int* launches, double* bytes) {
    if (!c || kind < -1 || kind >= GK_COUNT) return AHA_E_INVAL;
    float total = 0.f;
    int n = 0;
    double by = 0;
    for (int k = 0; k < GK_COUNT; ++k) {
        if (kind == -1 ? k >= GK_GEMMS : kind != k) continue;          // -1: the four GEMM kinds together
        for (int i = 0; i < c->ev_used[k]; ++i) {
            float t = 0.f;
            HIPCHK(c, hipEventSynchronize(c->ev[k][i].second));
            HIPCHK(c, hipEventElapsedTime(&t, c->ev[k][i].first, c->ev[k][i].second));
            total += t;
            ++n;
        }
        by += c->gk_bytes[k];
    }
    if (ms) *ms = total;
    if (launches) *launches = n;
    if (bytes) *bytes = by;
    return 0;
}
