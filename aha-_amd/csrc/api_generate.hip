// C ABI, part 5: response generation (fast_greedy_generate).
#include "api_internal.h"

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

