// C ABI, part 1: context, tuning, weight repacking, tables, per-stream KV state machines (the reference's cache policies as
// ring bookkeeping: plan_stream) - include/aha_amd.h.
#include "api_internal.h"

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
    c->attn_rows_pad = rows_pad;
    c->part_o_floats = (size_t)d->kv_heads * 16 * rows_pad * Dh;   // 16 key splits of a full step; more (up to AHA_MAX_KEY_SPLITS) for smaller batches: attn_geometry
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
    // ---- device error flag
    if ((rc = dalloc(c, &c->bar_err, 1))) return rc;
    if (hipMemset(c->bar_err, 0, sizeof(int)) != hipSuccess)
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
    for (const void* w : c->kb_keys) aha_gemm_tile_kb_register(w, nullptr, 0, 0);
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
    else if (k == "vit_prefetch") c->vit_prefetch_rows = value;   // rows up to which the tower's LayerNorm launches carry weight-prefetch riders (0 off)
    else if (k == "vit_akb") c->vit_akb = value;                  // k-blocked A operands on the tower's throughput path (0: row-major everywhere)
    else if (k == "vit_riders") c->vit_riders = value < 0 ? 0 : (value > 1024 ? 1024 : value);   // rider workgroups per prefetching launch
    else if (k == "vit_alias") c->vit_alias = value;              // diagnostic (tools/diag/vit_alias.py): wrong embeddings on purpose
    else if (k == "engine_exp") c->eng_exp = value;
    else if (k == "engine") {
        c->engine = value;
        // a hand-off that timed out inside an engine launch set the sticky error word (NaN scores from then on): choosing an engine level again
        // - 0 included - clears it, so that the experiment cannot leave a context unusable
        if (c->bar_err) { hipDeviceSynchronize(); hipMemset(c->bar_err, 0, sizeof(int)); }
    }                    // single-launch MLP-half experiments on single-stream steps: 1 lm_engine.hip, 2 lm_stream.hip; default 0
    else if (k == "use_graph") c->use_graph = value;              // 1 (default): replay frozen-static steps from a captured HIP graph
    else if (k == "kc_small") aha_gemm_ws_set_kc_small(value);
    else if (k == "attn_lm") aha_attention_set_lm_kernel(value);   // 1 (default): frame-sized LM steps use attn_lm_kernel (LDS-DMA, all row tiles per workgroup)
    else if (k == "attn_head") aha_attention_set_head_kernel(value);   // whole-head-in-LDS dense (ViT) attention: 0 off, 1 auto, 2 always when eligible
    else if (k == "attn_d96") aha_attention_set_d96(value);             // 96-wide dense attention template for head dims 65..96: 0 pads to 128
    else if (k == "attn_tpw") aha_attention_set_dense_tpw(value);   // dense attention: query tiles per wave (0 auto)
    else if (k == "tile_dma") aha_gemm_tile_set_dma(value);
    else if (k == "tile_p288") aha_gemm_tile_set_p288(value);    // 1 (default): persistent 288x256 tile kernel on the throughput shapes
    else if (k == "tile_wkb") aha_gemm_tile_set_wkb(value);      // 1 (default): the persistent tile kernel reads registered weights from their k-blocked twins
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

// k-blocked twin [K/32][N][32] of a row-major tile-GEMM weight [N][ld] (gemm_tile_p.hip reads it 1 KiB per DMA piece); K % 32 == 0
static int make_kb_twin(aha_ctx* c, const bf16* w, int N, int K, int ld, hipStream_t st) {
    if (!w || K % 32 || ld % 8) return 0;
    bf16* kb = nullptr;
    int rc = dalloc(c, &kb, (size_t)N * K);
    if (rc) return rc;
    HIPCHK(c, aha_rows_to_kblocked(w, N, K, ld, kb, st));
    aha_gemm_tile_kb_register(w, kb, N, K);
    c->kb_keys.push_back(w);
    return 0;
}

int alloc_packed(aha_ctx* c, PackedW* w, int n_tiles, int K) {
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
        if ((rc = make_kb_twin(c, c->patch_w, Dv, c->Kp, c->Kp, st))) return rc;
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
        // k-blocked twins for the persistent tile kernel (throughput path; the latency path's 64x64 kernels read the row-major weights)
        if ((rc = make_kb_twin(c, w.wqkv, 3 * Dv, Dv, Dv, st)) || (rc = make_kb_twin(c, w.wo, Dv, Dv, Dv, st)) ||
            (rc = make_kb_twin(c, w.w1, d.v_inter, Dv, Dv, st)) || (rc = make_kb_twin(c, w.w2, Dv, c->Fp, c->Fp, st))) return rc;
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
    if ((rc = make_kb_twin(c, c->p0w, H, Dv, Dv, st)) || (rc = make_kb_twin(c, c->p2w, H, H, H, st))) return rc;
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
int plan_stream(aha_ctx* c, aha_stream* s, int T, StreamStep* o) {
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

