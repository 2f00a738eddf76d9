// C ABI, part 2: frame ingest, the vision tower and the encode entry points (aha_vit_encode*).
#include "api_internal.h"

// --------------------------------------------------------------------------------------------
// vision
// --------------------------------------------------------------------------------------------
static void tile_args(GemmTileArgs& g, const bf16* A, int lda, int M, const bf16* W, int ldw, int N, int K, bf16* C, int ldc, const bf16* bias,
                      int act, const bf16* residual, int ldr, const bf16* rowadd, int period, int ldra) {
    g.A = A; g.lda = lda; g.M = M; g.W = W; g.ldw = ldw; g.N = N; g.K = K; g.C = C; g.ldc = ldc; g.bias = bias; g.act = act;
    g.residual = residual; g.ldr = ldr; g.rowadd = rowadd; g.rowadd_period = period > 0 ? period : 1; g.ldra = ldra;
    g.wide_epi = 0; g.Wkb = nullptr; g.akb = 0; g.ckb = 0;
}
hipError_t tile_gemm(const bf16* A, int lda, int M, const bf16* W, int ldw, int N, int K, bf16* C, int ldc, const bf16* bias,
                            int act, const bf16* residual, int ldr, const bf16* rowadd, int period, int ldra, hipStream_t st) {
    GemmTileArgs g;
    tile_args(g, A, lda, M, W, ldw, N, K, C, ldc, bias, act, residual, ldr, rowadd, period, ldra);
    return aha_gemm_tile(&g, st);
}

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
int vit_layers(aha_ctx* c, int n, int l0, int l1, hipStream_t st) {
    const aha_model_desc& d = c->d;
    const bool clip = d.v_kind == AHA_VISION_CLIP;
    const int Dv = d.v_hidden, T = c->Tt, rows = n * T, vhd = Dv / d.v_heads;
    const int act = clip ? ACT_QUICK_GELU : ACT_GELU_TANH;
    // Latency path (a few frames): the tower's kernels are latency-bound chains with HBM almost idle (25 MB of weights per ~75 us layer), and
    // every GEMM starts by waiting for its first weight tiles from HBM.  The two LayerNorm launches of a layer therefore carry RIDERS - extra
    // workgroups that only read bytes (prefetch_rider, aha_kernels.h) - for weights used from two launches on: LN1 the layer's out-proj and
    // fc1 weights, LN2 its fc2 weights and the next layer's QKV weights (after the last layer: the projector's first matrix, used by the
    // encode calls right after the tower); the first layer's QKV weights ride on its own LN1.  With the weights in the Infinity Cache the
    // one-frame encode measures 2.15 -> 1.88 ms (tools/diag/vit_prefetch.py; fully cache-resident weights, tuning "vit_alias": 1.74).
    // Riders change no output bit.  Tuning "vit_prefetch" = rows up to which it is on (0: off); the throughput path never prefetches
    // (its GEMMs are not latency-bound and the riders would only take CUs).
    const bool pf = rows <= c->vit_prefetch_rows && c->vit_riders > 0;
    // Throughput path: where the persistent tile kernel runs a GEMM (aha_gemm_tile_will_use_p288: from 8 frames of 576 patches up), its A
    // operand travels k-blocked [K/32][rows][32] - written that way by the producer (LayerNorm for QKV and fc1, fc1's epilogue for fc2) -
    // so that an LDS-DMA piece of 16 rows x 64 B is one contiguous KiB, as for the weights' twins (tuning "vit_akb"; same bits).
    GemmTileArgs gq, g1, g2;
    tile_args(gq, c->v_h, Dv, rows, c->V[l0 < d.v_layers ? l0 : 0].wqkv, Dv, 3 * Dv, Dv, c->v_qkv, 3 * Dv, nullptr, ACT_NONE, nullptr, 0, nullptr, 0, 0);
    tile_args(g1, c->v_h, Dv, rows, c->V[l0 < d.v_layers ? l0 : 0].w1, Dv, d.v_inter, Dv, c->v_f, c->Fp, nullptr, act, nullptr, 0, nullptr, 0, 0);
    tile_args(g2, c->v_f, c->Fp, rows, c->V[l0 < d.v_layers ? l0 : 0].w2, c->Fp, Dv, c->Fp, c->v_x, Dv, nullptr, ACT_NONE, c->v_x, Dv, nullptr, 0, 0);
    const bool kb_ok = c->vit_akb && !pf && Dv % 32 == 0;
    const bool kb_q = kb_ok && aha_gemm_tile_will_use_p288(&gq), kb_1 = kb_ok && aha_gemm_tile_will_use_p288(&g1);
    const bool kb_2 = kb_1 && c->Fp == d.v_inter && d.v_inter % 32 == 0 && aha_gemm_tile_will_use_p288(&g2);   // no zero-padded columns to keep clean
    const size_t b_qkv = (size_t)3 * Dv * Dv * 2, b_o = (size_t)Dv * Dv * 2, b_1 = (size_t)d.v_inter * Dv * 2, b_2 = (size_t)Dv * c->Fp * 2;
    for (int l = l0; l < l1; ++l) {
        const auto vw = [&](int i) -> const VLayerW& { return c->V[c->vit_alias > 0 ? i % c->vit_alias : i]; };
        const VLayerW& w = vw(l);
        const bool last = l + 1 >= d.v_layers;
        if (pf) {
            WeightPrefetch p1{{w.wo, w.w1, l == l0 ? w.wqkv : nullptr, nullptr}, {(long)b_o, (long)b_1, l == l0 ? (long)b_qkv : 0, 0}, c->vit_riders};
            HIPCHK(c, aha_layernorm_pf(c->v_x, Dv, w.ln1w, w.ln1b, c->v_h, Dv, rows, Dv, d.v_ln_eps, &p1, st));
        } else if (kb_q)
            HIPCHK(c, aha_layernorm_kb(c->v_x, Dv, w.ln1w, w.ln1b, c->v_h, rows, Dv, d.v_ln_eps, st));
        else
            HIPCHK(c, aha_layernorm(c->v_x, Dv, w.ln1w, w.ln1b, c->v_h, Dv, rows, Dv, d.v_ln_eps, st));
        {
            GemmTileArgs g;
            tile_args(g, c->v_h, Dv, rows, w.wqkv, Dv, 3 * Dv, Dv, c->v_qkv, 3 * Dv, w.bqkv, ACT_NONE, nullptr, 0, nullptr, 0, 0);
            g.akb = kb_q ? rows : 0;
            HIPCHK(c, aha_gemm_tile(&g, st));
        }
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
        if (pf) {
            WeightPrefetch p2{{w.w2, nullptr, nullptr, nullptr}, {(long)b_2, 0, 0, 0}, c->vit_riders};
            if (!last) { p2.p[1] = vw(l + 1).wqkv; p2.bytes[1] = (long)b_qkv; }
            else if (c->p0w) { p2.p[1] = c->p0w; p2.bytes[1] = (long)d.hidden * Dv * 2; }
            HIPCHK(c, aha_layernorm_pf(c->v_x, Dv, w.ln2w, w.ln2b, c->v_h, Dv, rows, Dv, d.v_ln_eps, &p2, st));
        } else if (kb_1) HIPCHK(c, aha_layernorm_kb(c->v_x, Dv, w.ln2w, w.ln2b, c->v_h, rows, Dv, d.v_ln_eps, st));
        else HIPCHK(c, aha_layernorm(c->v_x, Dv, w.ln2w, w.ln2b, c->v_h, Dv, rows, Dv, d.v_ln_eps, st));
        {
            GemmTileArgs g;
            tile_args(g, c->v_h, Dv, rows, w.w1, Dv, d.v_inter, Dv, c->v_f, c->Fp, w.b1, act, nullptr, 0, nullptr, 0, 0);
            g.akb = kb_1 ? rows : 0; g.ckb = kb_2 ? rows : 0;
            HIPCHK(c, aha_gemm_tile(&g, st));
            tile_args(g, c->v_f, c->Fp, rows, w.w2, c->Fp, Dv, c->Fp, c->v_x, Dv, w.b2, ACT_NONE, c->v_x, Dv, nullptr, 0, 0);
            g.akb = kb_2 ? rows : 0;
            HIPCHK(c, aha_gemm_tile(&g, st));
        }
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

