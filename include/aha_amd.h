/* aha_amd.h -- C ABI of the MI355X-native per-frame streaming-inference path.
 *
 * The reference (aiden200/Aha-) has no FFI or operator registry: its hot path sits behind
 * Python method calls.  Each entry point below names the reference interface it
 * serves; the Python host in aha-_amd/ (LiveInferForBenchmark / LiveInferForDemo / LiveLlava
 * mirrors) is a thin ctypes caller of exactly these symbols.  INTEGRATION.md shows the binding
 * a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success or a negative errno-style code and never
 * throws; aha_last_error() gives the message.  All tensor pointers are DEVICE pointers owned by
 * the caller (bf16 unless stated); KV caches are owned by aha_stream.  The per-frame calls
 * (aha_frame_ingest, aha_vit_encode*, aha_embed_tokens, aha_lm_step, aha_lm_heads_all,
 * aha_lm_last_hidden_all, aha_lm_logits*, aha_generate_greedy, aha_cache_update) enqueue work on the
 * given hipStream_t and never synchronise with the host, first use included (frame-ingest coefficient
 * tables are uploaded asynchronously on that stream; SinkCache re-rotation coefficients are computed
 * inside the kernel).  Set-up and tear-down calls
 * (aha_ctx_create / _load_weights / _set_rope_table / _set_rerotation_table / _destroy,
 * aha_stream_open / _destroy) may block.  One aha_ctx per process / GPU; not thread-safe.  The LM and
 * vision workspaces belong to the context: a call submitted on a different HIP stream than the
 * previous call of the same family is ordered behind it with an event (serialised, never racing).
 * If a per-frame call fails before any cache-changing work was enqueued, the streams' bookkeeping
 * (length, ring head, seen tokens) is restored to what it was before the call and the call may be
 * retried; if it fails after an evicting step's in-place re-rotation / ring overwrite was enqueued,
 * the streams are marked unusable instead (every later step returns -22) until aha_stream_reset.
 */
#ifndef AHA_AMD_H
#define AHA_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct aha_ctx aha_ctx;
typedef struct aha_stream aha_stream;
typedef void* aha_hip_stream;                 /* hipStream_t */

/* Shapes.  Vision fields <- LLaVA-NeXT SigLIP tower config (call site
 * models/live_llava/video_head_live_llava_qwen.py:107-115); LM fields <- Qwen2Config
 * (:43-47,74); pooling <- video_pooling_stride / mm_spatial_pool_mode (:117-136). */
typedef struct aha_model_desc {
    int32_t image_size, patch_size, v_hidden, v_layers, v_heads, v_inter;
    float v_ln_eps;
    int32_t hidden, layers, heads, kv_heads, head_dim, inter, vocab;
    float rope_theta, rms_eps;
    int32_t max_positions;                    /* rows of the RoPE table */
    int32_t pool_stride, pool_mode;           /* pool_mode: 0 bilinear, 1 average, 2 max */
    int32_t max_step_tokens;                  /* largest B*T of one aha_lm_step */
    int32_t max_vit_frames;                   /* largest n_frames of one aha_vit_encode */
    int32_t v_kind;                           /* AHA_VISION_SIGLIP (0) or AHA_VISION_CLIP (1: class token, pre_layrnorm, quick_gelu,
                                                 OpenAI mean/std; serves models/vision_live.py:34-54 via aha_vit_encode_pooled_first) */
} aha_model_desc;
enum { AHA_VISION_SIGLIP = 0, AHA_VISION_CLIP = 1 };

/* One named checkpoint tensor (bf16, row-major, device memory).  Names are the checkpoint's:
 * "model.layers.N.self_attn.q_proj.weight", "mm_projector.0.weight", "informative_head.weight",
 * "vision.encoder.layers.N.mlp.fc1.weight", ... (aha-_amd/synth.py lists all of them). */
typedef struct aha_tensor_view {
    const char* name;
    const void* data;
    int64_t shape[4];
    int32_t ndim;
    int32_t reserved;
} aha_tensor_view;

/* alt_cache of LiveInferForBenchmark.__init__/_init_cache (test/inference.py:39,133-155) */
enum { AHA_CACHE_NONE = 0,      /* past_key_values=None -> growing DynamicCache            */
       AHA_CACHE_SINK = 1,      /* test/sink_cache.py SinkCache(window_length, num_sink)   */
       AHA_CACHE_SLIDING = 2,   /* test/sliding_window_cache.py SlidingWindowCache(W)      */
       AHA_CACHE_STATIC = 3 };  /* test/static_cache.py TrulyStaticCache(window_size)      */

/* Attention-mask arithmetic (DESIGN.md "Mask semantics"): 0 = trailing-causal (parity target of
 * SURVEY.md 8c), 1 = transformers-4.49 sdpa mask arithmetic, 2 = flash-attn-2 (the reference's default
 * attn_implementation, models/arguments_live.py:30): bottom-right aligned causal mask - equal to 0 for every
 * cache policy except a frozen TrulyStaticCache, where new token t sees prefix key j iff j <= t + (L - T)
 * (rows that see no key give 0).  The scores the drivers read (position -1) are the same under 0 and 2. */
enum { AHA_ATTN_TRAILING = 0, AHA_ATTN_HF449_SDPA = 1, AHA_ATTN_FA2 = 2 };

/* ---- context ---------------------------------------------------------------------------- */
/* replaces build_model_and_tokenizer()/build_live() model construction
 * (models/__init__.py:8-11, models/modeling_live.py:96-181) for the inference path */
int aha_ctx_create(const aha_model_desc* desc, int device, aha_ctx** out);
/* replaces from_pretrained weight materialisation (models/modeling_live.py:137-144).  Tensors are
 * COPIED and repacked into the kernels' private layouts; the caller may free them afterwards.
 * Once per context: a second call is rejected (-22) rather than leaking the first set. */
int aha_ctx_load_weights(aha_ctx* ctx, const aha_tensor_view* tensors, size_t n, aha_hip_stream st);
/* RoPE cos/sin table, bf16 [n_pos][head_dim] (Qwen2RotaryEmbedding.forward output cast to bf16,
 * transformers modeling_qwen2.py:87-102); copied. */
int aha_ctx_set_rope_table(aha_ctx* ctx, const void* cos_bf16, const void* sin_bf16, int n_pos, aha_hip_stream st);
/* SinkCache._get_rerotation_cos_sin table for new-token count T (test/sink_cache.py:35-55),
 * bf16 [window - n_sink - T][head_dim]; copied, keyed by (window, n_sink, T).  OPTIONAL: without one the
 * re-rotation kernel computes the identical coefficients from the RoPE table on the fly (nothing is
 * allocated or built inside a per-frame call); this entry point lets a caller supply its own instead. */
int aha_ctx_set_rerotation_table(aha_ctx* ctx, int window, int n_sink, int T, const void* cos_bf16,
                                 const void* sin_bf16, aha_hip_stream st);
int aha_ctx_has_rerotation_table(aha_ctx* ctx, int window, int n_sink, int T);
/* tuning knobs (all optional; defaults are the measured best, DESIGN.md sections 4/5/8):
 *   "split_qkv" / "split_o" / "split_down"  split-K factors of the weight-streaming GEMMs (0 = heuristic; never depends on M)
 *   "wpb_qkv" / "wpb_o" / "wpb_gateup" / "wpb_down"  waves per workgroup (2..8) = over how many CUs a GEMM's wave-tasks spread
 *   "kc_small"        k-steps per pipeline chunk of the small split-K GEMMs (4 or 8)
 *   "attn_split_len"  keys per attention split (0 = heuristic)      "attn_tpw"  query tiles per wave of the dense (ViT) attention
 *   "tile_dma"        tiled-GEMM variant (0 register-staged, 1 auto, >= 2 forced variant id)
 *   "use_graph"       1: replay frozen static-cache steps from a captured HIP graph      "fuse_static"  experiment
 *   "time_gemm"       bit k: bracket GEMM kind k's launches with HIP events (aha_lm_last_gemm_time); such steps are launched
 *                     directly, not replayed from a graph, so the events are live
 *   "pool_subset"     1 (default): the projector runs only on the patch rows that bilinear pooling with an even integer stride samples
 *                     (24 -> 6: 144 of 576 per frame); bit-identical embeddings
 *   "engine"          round-6 experiments on single-stream steps (<= 48 rows), bit-identical to the launches, neither ahead of them
 *                     (profiles/r06_engine_mlp_stamps.txt): 1 = each layer's post-attention resid_norm, gate/up + SwiGLU and down_proj as ONE
 *                     persistent launch on an LDS-DMA weight ring (lm_engine.hip); 2 = gate/up + SwiGLU and down_proj as one launch
 *                     whose weight stream runs ahead of the hand-off in registers (lm_stream.hip); 0 (default): the launches
 *   "static_attn"     1 (default): frozen TrulyStaticCache steps whose prefix is <= 64 keys run qkv_finish + attention as one launch
 *   "attn_lm"         LM attention kernel for frame-sized steps: 0 attn_fwd_kernel always, 1 auto (default), 2 attn_lm_kernel always
 *   "use_wl"          1 (default): row chunks above 128 use the mid-M GEMM kernel (both operands staged through LDS); 0: never.
 *                     Bit-identical either way
 *   "act_kb"          2 (default): between mid-M kernels the normed gate/up input and the SwiGLU activation travel k-blocked
 *                     ([K/32][M][32]); 1: the activation only; 0: row-major.  "wl_bal" 1 (default): the 18-row-tile gate/up kernel
 *                     deals its MFMA units by SIMD occupancy; 0: nine units per wave.  Bit-identical either way
 *   "tile_epi"        1 (default): the LDS-DMA tiled GEMMs stage the finished tile in LDS and store full 128-byte rows; 0: direct
 *                     accumulator-layout stores.  Bit-identical either way
 *   "layer_first" / "layer_count"  run decoder layers [first, first+count) only (0 = all): teacher-forced per-layer parity */
int aha_ctx_set_tuning(aha_ctx* ctx, const char* key, int value);
void aha_ctx_destroy(aha_ctx* ctx);
const char* aha_last_error(aha_ctx* ctx);

/* ---- vision: LiveMixin.visual_embed (models/modeling_live.py:31-37) on raw uint8 frames, i.e.
 * image_processor.preprocess (test/inference.py:176) + vision_tower + mm_projector +
 * post_projector_pooling (video_head_live_llava_qwen.py:107-136) ----------------------------- */
/* frames_u8: [n,3,S,S] uint8 RGB; out_embeds: bf16 [n*Tf][hidden].  With v_kind = AHA_VISION_CLIP: CLIP tower, patch features
 * only (the class token is dropped, LLaVA's select_feature = 'patch'), OpenAI mean/std. */
int aha_vit_encode(aha_ctx* ctx, const uint8_t* frames_u8, int n_frames, void* out_embeds, aha_hip_stream st);
/* The encode contract of models/vision_live.py:11-31 (_siglip_vision_encode, frame_token_cls=False) followed
 * by LiveMixin's connector (models/modeling_live.py:31-37): tower -> post_layernorm (last_hidden_state) ->
 * adaptive_avg_pool2d to pooled x pooled (frame_token_pooled, models/arguments_live.py:21) -> mm_projector.
 * Needs "vision.post_layernorm.{weight,bias}".  With v_kind = AHA_VISION_CLIP it is _clip_vision_encode (:34-54):
 * OpenAI mean/std, CLIP tower (class token, pre_layrnorm, quick_gelu), last_hidden_state WITHOUT post-layernorm,
 * class token dropped, same pooling and connector.  out_embeds: bf16 [n*pooled*pooled][hidden] */
int aha_vit_encode_pooled_first(aha_ctx* ctx, const uint8_t* frames_u8, int n_frames, int pooled, void* out_embeds,
                                aha_hip_stream st);
/* The same contract with frame_token_cls (models/vision_live.py:26-31 / :50-54; models/arguments_live.py:20): cls != 0 puts the class
 * token in front of each frame's pooled tokens.  SigLIP: the vision model's pooler_output, i.e. its attention-pooling head on the
 * post-layernormed tokens (needs "vision.head.*": probe, attention.in_proj_{weight,bias}, attention.out_proj, layernorm, mlp.fc1/fc2);
 * CLIP: last_hidden_state[:, 0], which the reference can return only without pooling (its torch.cat at :54 raises otherwise; AHA_E_INVAL
 * here).  pooled = 0 with cls: the class token alone.  out_embeds: bf16 [n*(cls + pooled*pooled)][hidden] */
int aha_vit_encode_live(aha_ctx* ctx, const uint8_t* frames_u8, int n_frames, int pooled, int cls, void* out_embeds, aha_hip_stream st);
/* parity-test tap: copy the tower output of the last encode, bf16 [n_frames*Np][v_hidden] */
int aha_vit_last_tower_output(aha_ctx* ctx, int n_frames, void* out, aha_hip_stream st);

/* ---- frame ingest: what the reference does to a decoded frame before preprocess -- aspect-preserving resize to
 * image_size on the long side, centred zero pad to image_size x image_size, RGB, CHW.  Integer arithmetic, bit-exact.
 *   AHA_RESIZE_PIL_BICUBIC : PIL Image.resize((w,h)) default resample + ImageOps.expand, as
 *                            LiveInferForDemo.load_one_frame does (test/live_infer_for_video.py:98-121)
 *   AHA_RESIZE_CV2_LINEAR  : cv2.resize default INTER_LINEAR + copyMakeBorder(BORDER_CONSTANT 0) + BGR2RGB, the
 *                            per-frame body of load_video_for_testing / load_video (test/inference.py:538-562,
 *                            test/live_infer_for_video.py:49-71)
 * src_hwc_u8: uint8 [height][width][3] on the device, channel order B,G,R when src_is_bgr (cv2 decode) else R,G,B;
 * out_canvas_u8: uint8 [3][image_size][image_size], ready for aha_vit_encode. */
enum { AHA_RESIZE_PIL_BICUBIC = 0, AHA_RESIZE_CV2_LINEAR = 1 };
int aha_frame_ingest(aha_ctx* ctx, const uint8_t* src_hwc_u8, int height, int width, int src_is_bgr, int method,
                     uint8_t* out_canvas_u8, aha_hip_stream st);

/* model.get_input_embeddings()(ids) (test/inference.py:212) */
int aha_embed_tokens(aha_ctx* ctx, const int64_t* ids_dev, int n, void* out_embeds, aha_hip_stream st);

/* ---- per-stream KV state: the Cache object of test/{sink,sliding_window,static}_cache.py -------- */
/* capacity: slots for AHA_CACHE_NONE (others use window) */
int aha_stream_open(aha_ctx* ctx, int policy, int window, int n_sink, int capacity, aha_stream** out);
int aha_stream_reset(aha_stream* s);                       /* LiveInferForBenchmark.reset -> _init_cache */
int aha_stream_seq_length(const aha_stream* s);            /* Cache.get_seq_length()      */
int aha_stream_seen_tokens(const aha_stream* s);           /* Cache._seen_tokens          */
int aha_stream_set_attn_semantics(aha_stream* s, int semantics);
/* RoPE position of the next steps' new token 0 = seq_length + offset (default 0 = the reference's
 * `cache_position` rule).  Lets a caller that feeds only the tail of a chunk keep the positions the
 * whole chunk would have had (aha_amd.live_infer static last-token mode). */
int aha_stream_set_position_offset(aha_stream* s, int offset);
/* test tap: copy layer `layer`'s K or V as the attention sees it (logical order) into
 * out bf16 [kv_heads][seq_len][head_dim] */
int aha_stream_export_kv(aha_ctx* ctx, const aha_stream* s, int layer, int want_v, void* out, aha_hip_stream st);
void aha_stream_destroy(aha_stream* s);

/* ---- LM step: VideoHeadLiveLlavaQwenForCausalLM.forward inference branch
 * (video_head_live_llava_qwen.py:156-188) on inputs_embeds [B,T,hidden] for B independent streams,
 * including every layer's Cache.update, reduced to what _encode_frame reads
 * (test/inference.py:217-227):
 *   out_scores    fp32 [B][3] = softmax(informative_logits[b,-1])[1], relevance_logits[b,-1]
 *                               (post-sigmoid), exp(uncertainty[b,-1])
 *   out_raw_heads fp32 [B][4] (optional) = informative logits (2), relevance logit, log-variance
 *   out_last_hidden bf16 [B][hidden] (optional) = final-norm hidden state of the last token */
/* A stream may appear once per step; exception: a TrulyStaticCache stream after its first call is
 * frozen (test/static_cache.py:26-36), its frames are independent, so it may be listed several
 * times to score several frames of that stream with one pass over the weights. */
int aha_lm_step(aha_ctx* ctx, aha_stream* const* streams, int B, const void* embeds, int T, float* out_scores,
                float* out_raw_heads, void* out_last_hidden, aha_hip_stream st);
/* all-token head outputs of the last step: fp32 [B*T][4] (LiveLlava forward()'s informative_logits /
 * relevance_logits(pre-sigmoid) / uncertainty for every position) */
int aha_lm_heads_all(aha_ctx* ctx, float* out_raw_heads, aha_hip_stream st);
/* copy the final-norm hidden states of the last step, bf16 [B*T][hidden] (outputs[0] of Qwen2Model) */
int aha_lm_last_hidden_all(aha_ctx* ctx, void* out, aha_hip_stream st);
/* lm_head on the last token of each stream of the last step (outputs.logits[:, -1]; used by
 * _encode_query :261 and fast_greedy_generate, models/modeling_live.py:64-90):
 * logits fp32 [B][vocab] (optional), argmax int64 [B] (optional) */
int aha_lm_logits_last(aha_ctx* ctx, float* logits, int64_t* argmax, aha_hip_stream st);

/* lm_head on EVERY position of the last step: outputs.logits fp32 [B*T][vocab] of the reference forward
 * (video_head_live_llava_qwen.py:175).  Opt-in: the frame loop never reads it (53 GFLOP + 30 MB per frame), the model-API
 * mirror (aha_amd.model) calls it only when `.logits_all` is asked for. */
int aha_lm_logits_all(aha_ctx* ctx, float* logits, aha_hip_stream st);

/* ---- response generation: fast_greedy_generate (models/modeling_live.py:64-90) as called by _generate_response
 * (test/inference.py:264-281).  Greedy single-token steps from the `first_ids` prompt (device int64 [n_first]) against the
 * stream's cache, at most max_new_tokens, stopping after eos_token_id.  argmax -> embedding -> next step stay on the device;
 * the host polls the 8-byte token id behind each step to stop exactly at EOS, so a call blocks for the tokens it produces.
 * A response need not be one call: pass max_new_tokens = a chunk size and continue with first_ids = the chunk's last id
 * (n_first = 1) until EOS or the response limit - between chunks other streams may step (a multi-stream server is not
 * stalled by one stream's response).  aha_generate_greedy_cb additionally hands every new id to `on_token` as soon as it is
 * host-visible; a non-zero return stops after that token.  The callback may enqueue steps of OTHER streams on the same HIP
 * stream; it must not start another generation on this context.
 * repetition_penalty > 0 applies RepetitionPenaltyLogitsProcessor over `history` (device int64 [history_cap] holding
 * *history_len ids; generated non-EOS ids are appended like the reference's generated_token_ids list) - <= 0 disables it.
 * out_ids_host: host int64 [max_new_tokens]; *out_count: tokens produced (the last one is EOS unless the limit was hit or the
 * callback stopped it) - also set when the call fails part-way (the cache has advanced by the tokens fed so far). */
typedef int (*aha_token_cb)(void* user, int64_t token_id, int index);
int aha_generate_greedy_cb(aha_ctx* ctx, aha_stream* s, const int64_t* first_ids_dev, int n_first, int max_new_tokens,
                           int64_t eos_token_id, float repetition_penalty, int64_t* history_dev, int history_cap, int* history_len,
                           int64_t* out_ids_host, int* out_count, aha_token_cb on_token, void* user, aha_hip_stream st);
int aha_generate_greedy(aha_ctx* ctx, aha_stream* s, const int64_t* first_ids_dev, int n_first, int max_new_tokens,
                        int64_t eos_token_id, float repetition_penalty, int64_t* history_dev, int history_cap, int* history_len,
                        int64_t* out_ids_host, int* out_count, aha_hip_stream st);

/* ---- operator level: the pieces of the step as stand-alone operators on caller tensors.  The fused aha_lm_step is built
 * from exactly these kernels; the parity tests drive them one by one on oracle-supplied inputs, and an integrator can replace
 * individual modules of the reference with them. ------------------------------------------------------------------------ */
/* nn.Linear weight [N][K] (bf16, device) repacked into the streaming layout; `w_up` non-null makes a gate/up pair for the
 * SwiGLU epilogue (Qwen2MLP, transformers modeling_qwen2.py:35-49).  Set-up call (synchronises). */
typedef struct aha_linear aha_linear;
enum { AHA_EPI_SPLITK_F32 = 0,   /* out: fp32 [S][M][ldo] partial sums over S k-slices (S = aha_linear_split_k)  */
       AHA_EPI_BF16 = 1,         /* out: bf16 [M][ldo] = bf16(x W^T (+ bias))                                    */
       AHA_EPI_SWIGLU = 2,       /* out: bf16 [M][ldo] = bf16( bf16(silu(bf16(x Wg^T))) * bf16(x Wu^T) )         */
       AHA_EPI_F32 = 3 };        /* out: fp32 [M][ldo] holding the bf16-rounded product (lm_head(...).float())   */
int aha_linear_create(aha_ctx* ctx, const void* w, const void* w_up, int N, int K, aha_linear** out, aha_hip_stream st);
void aha_linear_destroy(aha_linear* lin);
int aha_linear_split_k(aha_ctx* ctx, const aha_linear* lin, int requested);   /* the S a request is clamped to */
int aha_linear_forward(aha_ctx* ctx, const aha_linear* lin, const void* x, int ldx, int M, int epilogue, int split_k,
                       const void* bias, void* out, int ldo, aha_hip_stream st);
/* nn.Linear on a row-major weight through the tiled MFMA GEMM of the vision tower (SigLIP / CLIP layers, mm_projector):
 * out = act(bf16(x W^T + bias)) (+ residual).  act: 0 none, 1 gelu_pytorch_tanh, 2 gelu (erf), 3 quick_gelu */
int aha_linear_tile_forward(aha_ctx* ctx, const void* x, int ldx, int M, const void* w, int ldw, int N, int K, const void* bias,
                            int act, const void* residual, int ldr, void* out, int ldo, aha_hip_stream st);
/* Qwen2RMSNorm (modeling_qwen2.py:236-254) */
int aha_rmsnorm_forward(aha_ctx* ctx, const void* x, int ldx, const void* w, void* out, int ldo, int M, int H, float eps,
                        aha_hip_stream st);
/* split-K reduce + residual add + RMSNorm: lin = bf16(sum_s partial[s]); h = bf16(h + lin) (in place); xn = w * norm(h) */
int aha_resid_rmsnorm_forward(aha_ctx* ctx, const float* partial, int S, void* h, const void* w, void* xn, int M, int H,
                              float eps, aha_hip_stream st);
/* the three scoring heads + score post-ops on hidden rows (video_head_live_llava_qwen.py:185-188, test/inference.py:222-227):
 * raw fp32 [rows][4] (optional), scores fp32 [rows][3] (optional) */
int aha_heads_forward(aha_ctx* ctx, const void* hidden, int ld, int rows, float* scores, float* raw, aha_hip_stream st);
/* Cache.update(key_states, value_states, layer_idx, cache_kwargs) of the reference's cache classes (test/sink_cache.py:74-164,
 * test/sliding_window_cache.py:17-44, test/static_cache.py:18-36) on one stream: k_new / v_new bf16 [kv_heads][T][head_dim]
 * (keys already rotated).  As in the reference, layer 0's call advances the bookkeeping and the remaining layers of the step
 * follow in order.  out_k / out_v (optional): the (K, V) update() returns, bf16 [kv_heads][seq_length][head_dim]. */
int aha_cache_update(aha_ctx* ctx, aha_stream* s, int layer_idx, const void* k_new, const void* v_new, int T, void* out_k,
                     void* out_v, aha_hip_stream st);
/* attention of T query rows per stream (bf16 [B][T][heads*head_dim], rotated) over the streams' caches as they are, layer
 * `layer`; key j visible to row t iff j <= causal_off[b] + t (null: seq_length - T, the trailing-causal rule); split_len: keys
 * per split (0 = default).  out: bf16 [B][T][heads*head_dim] */
int aha_attention_forward(aha_ctx* ctx, aha_stream* const* streams, int B, const void* q, int T, int layer, const int* causal_off,
                          int split_len, void* out, aha_hip_stream st);
/* parity tap: a workspace of the last aha_lm_step.  which: 0 residual stream [M][hidden] (hidden state after the last executed
 * layer), 1 normalised stream [M][hidden], 2 rotated queries, 3 attention output [M][heads*head_dim], 4 SwiGLU activation
 * [M][inter] (2-4: the last executed layer; tuning "layer_first" / "layer_count" select it) */
int aha_lm_debug_tap(aha_ctx* ctx, int which, void* out, aha_hip_stream st);

/* ---- multi-GPU: the one collective of the path.  Streams are independent; ranks only exchange their score
 * rows.  RCCL communicator from a unique id the host distributes (rank 0 calls aha_comm_unique_id, everyone
 * aha_comm_init_rank).  The reference has no inference collective (utils/dist_utils.py:46-78 is training-only). */
typedef struct aha_comm aha_comm;
#define AHA_COMM_ID_BYTES 128
int aha_comm_unique_id(void* id_out, size_t bytes);
int aha_comm_init_rank(const void* id, size_t bytes, int nranks, int rank, int device, aha_comm** out);
int aha_comm_size(const aha_comm* comm);
int aha_comm_rank(const aha_comm* comm);
/* local fp32 [rows][3] -> global fp32 [nranks][rows][3]; ncclAllGather on `st`, asynchronous */
int aha_allgather_scores(aha_comm* comm, const float* local, int rows, float* global, aha_hip_stream st);
void aha_comm_destroy(aha_comm* comm);
const char* aha_comm_last_error(void);

/* ---- introspection ------------------------------------------------------------------------- */
/* algorithmic bytes / flops of the last aha_lm_step (SURVEY.md 8d accounting) */
int aha_lm_last_step_work(aha_ctx* ctx, double* weight_bytes, double* kv_bytes, double* flops);
/* time of one kind of launch of the last step, measured with HIP events on the launch stream when enabled with
 * aha_ctx_set_tuning("time_gemm", mask) (bit k = kind k).  kind: 0 qkv, 1 o_proj, 2 gate/up(+SwiGLU), 3 down_proj
 * (weight-streaming GEMMs; -1 = those four together), 4 attention over the KV cache (+ split combine), 5 SinkCache
 * re-rotation.  Returns summed ms, launch-group count and the ALGORITHMIC bytes of those launches (packed weight bytes
 * streamed; K+V bytes read; kept keys read + written).  Synchronises on the recorded events. */
int aha_lm_last_gemm_time(aha_ctx* ctx, int kind, float* ms, int* launches, double* gemm_weight_bytes);
/* Diagnostic of the persistent layer engine (tuning "engine"; the launch that replaces the post-attention resid_norm, gate/up and
 * down_proj launches of a single-stream step - the MLP of the decoder layer /root/reference/test/inference.py:217 runs): `stamps`
 * = device buffer of [CUs][16] uint64, or null to stop.  While set, every engine launch overwrites it with 100-MHz wall-clock
 * stamps per workgroup (0 loader start, 1/2 last weight slot of gate/up / down issued, 3 loader drained, 4/5 gate/up's / down's
 * input published, 6/7/8 gate/up first slot in hand / last slot consumed / activation published, 9/10 the same for down,
 * 12 row normalised, 13 workgroup done); graph replay is bypassed for such steps. */
int aha_lm_engine_stamps(aha_ctx* ctx, void* stamps);
const char* aha_version(void);

/* ---- vision operators: the tower's non-GEMM kernels on caller tensors (its GEMMs are aha_linear_tile_forward).  The parity
 * tests bound each of them against an exact reference; an integrator can swap them in for the reference's modules. ------------ */
/* SiglipAttention / CLIPAttention core (the attention the vision tower of video_head_live_llava_qwen.py:113-115 runs):
 * softmax(q k^T / sqrt(head_dim)) v per frame and head, non-causal.  qkv: bf16 [n][T][3*heads*head_dim] (q | k | v per row);
 * out: bf16 [n][T][heads*head_dim]. */
int aha_vit_attention_forward(aha_ctx* ctx, const void* qkv, int n_frames, int T, int heads, int head_dim, void* out, aha_hip_stream st);
/* SiglipEncoderLayer / CLIPEncoderLayer layer_first .. layer_first + layer_count - 1 of the tower (the arithmetic the tower of
 * video_head_live_llava_qwen.py:113-115 runs per layer) on a caller-supplied hidden state: x bf16 [n * tokens][v_hidden] -> out. */
int aha_vit_layers_forward(aha_ctx* ctx, const void* x, int n_frames, int layer_first, int layer_count, void* out, aha_hip_stream st);
/* nn.LayerNorm (fp32 statistics) of the tower's layer_norm1 / layer_norm2 / post_layernorm: bf16 [rows][ldx] -> bf16 [rows][ldo]. */
int aha_layernorm_forward(aha_ctx* ctx, const void* x, int ldx, const void* weight, const void* bias, void* out, int ldo, int rows, int cols,
                          float eps, aha_hip_stream st);
/* image_processor.preprocess (test/inference.py:176) fused with the unfold of the patch-embedding conv: uint8 [n][3][S][S] ->
 * bf16 [n*Np][Kp] normalised patch vectors (column c*P*P + y*P + x; columns >= 3*P*P are zero padding); *out_cols = Kp. */
int aha_vit_patchify_forward(aha_ctx* ctx, const uint8_t* frames_u8, int n_frames, void* out, int* out_cols, aha_hip_stream st);
/* post_projector_pooling (video_head_live_llava_qwen.py:117-136) / adaptive_avg_pool2d (models/vision_live.py:21-24) of a token
 * grid: in bf16 [n][frame_rows][C], first grid^2 rows = the patch grid -> out bf16 [n][out_grid^2][C].  mode 0 bilinear
 * (align_corners=False), 1 average, 2 max (kernel = stride), 3 adaptive average. */
int aha_pool_forward(aha_ctx* ctx, const void* in, int n_frames, int grid, int out_grid, int channels, int stride, int mode, int frame_rows,
                     void* out, aha_hip_stream st);
/* the (2*out_grid)^2 patch rows per frame that bilinear pooling with an even integer stride reads, compacted (aha_vit_encode
 * runs the projector on these only): in bf16 [n][frame_rows][C] -> out bf16 [n][(2*out_grid)^2][C]. */
int aha_pool_gather_rows_forward(aha_ctx* ctx, const void* in, int n_frames, int grid, int out_grid, int channels, int frame_rows, void* out,
                                 aha_hip_stream st);

#ifdef __cplusplus
}
#endif
#endif /* AHA_AMD_H */
