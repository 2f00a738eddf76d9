// C ABI, part 4: operator-level entry points - the step's and the tower's kernels on caller tensors, Cache.update.
#include "api_internal.h"

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
    int sl, ns;
    attn_geometry(c, B, T, max_lk, split_len, &sl, &ns);            // the choice aha_lm_step makes for this shape (or the forced length)
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
    if (ldx < cols || ldo < cols || (ldx & 7) || (ldo & 7)) return fail(c, AHA_E_RANGE, "row strides must be >= cols and multiples of 8 elements");
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
    if (!c || !in || !out || n <= 0 || grid <= 0 || out_grid <= 0 || C_ <= 0 || (C_ & 7) || mode < 0 || mode > 3) return AHA_E_INVAL;   // 8-channel chunks
    if (frame_rows < grid * grid) return fail(c, AHA_E_RANGE, "frame_rows < grid^2");
    if ((mode == 1 || mode == 2) && (stride < 1 || (long)out_grid * stride > grid))
        return fail(c, AHA_E_RANGE, "avg / max pooling reads rows (oy*stride + dy): needs stride >= 1 and out_grid * stride <= grid");
    if ((mode == 0 || mode == 3) && out_grid > grid) return fail(c, AHA_E_RANGE, "out_grid > grid");
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

