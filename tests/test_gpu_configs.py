"""BASELINE.json configs[0], [2], [3], [4] on the HIP path, plus the two "next" rows that need the GPU (response generation,
checkpoint loading).  Score tolerances here follow tests/test_gpu_parity.py (the oracle's own bf16 band);
the flat, depth-independent bounds live in tests/test_gpu_kernels.py and tests/test_gpu_layers.py.
"""
import hashlib
import json
import os
import sys

import numpy as np
import pytest
import torch

import aha_amd  # noqa: F401
from aha_amd.config import preset
from aha_amd.synth import make_frames, make_token_ids, make_weights

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCORE_TOL = 1e-3
KEYS = ("informative_score", "relevance_score")


def _rel(s):
    return torch.stack([s[:, 0], s[:, 1], torch.log(s[:, 2])], dim=-1)


def _driver_pair(cfg, w, rt, alt_cache, W, S, tok, dtypes=(torch.bfloat16, torch.float32), **kw):
    """The HIP driver and two oracle drivers (bf16, fp32) set up identically (a dtype left out of `dtypes` gives None)."""
    from aha_amd.arguments import LiveTestArguments
    from aha_amd.live_infer import LiveInferForBenchmark
    from oracle.live_driver import OracleLiveInfer
    a = dict(frame_fps=1)
    a.update({k: v for k, v in kw.items() if k in ("stream_end_prob_threshold", "stream_end_score_sum_threshold", "repetition_penalty")})
    if "stream_end_prob_threshold" not in a and "stream_end_score_sum_threshold" not in a:
        a["stream_end_prob_threshold"] = 9.0
    args = LiveTestArguments(**a)
    drv = LiveInferForBenchmark(args, alt_cache=alt_cache, runtime=rt, tokenizer=tok, window_length=W, num_sink_tokens=S)
    okw = dict(alt_cache=alt_cache, window_length=W, num_sink_tokens=S, frame_fps=1,
               start_ids=tok.apply_chat_template([{"role": "system", "content": args.system_prompt}]),
               stream_prompt_ids=tok.apply_chat_template([{}], add_stream_prompt=True),
               stream_generation_ids=tok.apply_chat_template([{}], add_stream_generation_prompt=True),
               stream_end_prob_threshold=args.stream_end_prob_threshold,
               stream_end_score_sum_threshold=args.stream_end_score_sum_threshold,
               repetition_penalty=args.repetition_penalty, eos_token_id=getattr(tok, "eos_token_id", 0),
               max_new_tokens=kw.get("max_new_tokens", 200))
    drv.max_new_tokens = kw.get("max_new_tokens", 200)
    return (drv, OracleLiveInfer(cfg, w, dtype=torch.bfloat16, **okw) if torch.bfloat16 in dtypes else None,
            OracleLiveInfer(cfg, w, dtype=torch.float32, **okw) if torch.float32 in dtypes else None)


# ---------------------------------------------------------------------------------------------------------------------
# configs[0]: the plumbing preset (SigLIP-base dims + 2-layer LM), 32 frames, every cache policy
# ---------------------------------------------------------------------------------------------------------------------
def test_config0_plumbing_preset_32_frames_every_policy():
    from aha_amd.runtime import Runtime
    from aha_amd.tokenization import SyntheticChatTokenizer
    from oracle.vision_tower import OracleVision
    cfg = preset("plumbing")
    w = make_weights(cfg, dtype=torch.bfloat16, jitter=True)
    rt = Runtime(cfg, w, max_step_tokens=256, max_vit_frames=32, max_positions=4096)
    tok = SyntheticChatTokenizer(cfg.lm.vocab_size)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    frames = make_frames(32, cfg.vision.image_size, seed=0)                  # seed 0, uint8 [32,3,336,336]
    tf = cfg.frame_num_tokens
    emb = {dt: OracleVision(cfg, w, dt).visual_embed(frames).split(tf) for dt in (torch.bfloat16, torch.float32)}
    got_e = rt.visual_embed(frames.cuda()).float().cpu()
    e32 = torch.cat(emb[torch.float32]).float()
    band_e = (torch.cat(emb[torch.bfloat16]).float() - e32).abs().max().item()
    assert (got_e - e32).abs().max().item() <= max(2.0 * band_e, 0.01 * e32.abs().max().item())
    q = "tell me when something happens"
    qids = tok.apply_chat_template([{"role": "user", "content": q}], add_stream_prompt=True)
    for alt in (None, "default_sink", "sliding_window", "static"):
        drv, ob, o32 = _driver_pair(cfg, w, rt, alt, 256, 8, tok)
        drv.input_video_stream(frames)
        drv.input_query_stream([{"role": "user", "content": q, "time": 0}])
        drv.inference()
        for o, dt in ((ob, torch.bfloat16), (o32, torch.float32)):
            o.frame_embeds_queue.extend([(i / 1.0, e) for i, e in enumerate(emb[dt])])
            o.input_query_stream([{"role": "user", "time": 0, "ids": qids}])
            o.inference()
        assert len(drv.debug_data_list) == 32
        assert drv.past_key_values.get_seq_length() == ob.past_key_values.get_seq_length(), alt
        assert [d["time"] for d in drv.debug_data_list] == [d["time"] for d in ob.debug_data_list]
        band = max(abs(a[k] - b[k]) for a, b in zip(ob.debug_data_list, o32.debug_data_list) for k in KEYS)
        d32 = max(abs(a[k] - b[k]) for a, b in zip(drv.debug_data_list, o32.debug_data_list) for k in KEYS)
        assert d32 <= max(SCORE_TOL, 2.0 * band), (alt, d32, band)
    rt.close()


# ---------------------------------------------------------------------------------------------------------------------
# configs[2]: long streams through the evicting caches at full size
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def bench_rt():
    from aha_amd.runtime import Runtime
    cfg = preset("bench")
    w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
    rt = Runtime(cfg, w, max_step_tokens=640, max_vit_frames=32)
    w_cpu = {k: v.cpu() for k, v in w.items() if not k.startswith("vision.")}
    del w
    torch.cuda.empty_cache()
    yield cfg, rt, w_cpu
    rt.close()


def _frames_batch(S, i0, n):
    """counter-based frames (frame i depends only on i): no multi-GB host buffer for a 10k-frame stream"""
    g = torch.Generator(device="cuda")
    out = []
    for i in range(i0, i0 + n):
        g.manual_seed(i)
        out.append(torch.randint(0, 256, (3, S, S), generator=g, device="cuda", dtype=torch.uint8))
    return torch.stack(out)


def _long_streams(cfg, rt, policies, n_frames, keep_first):
    """The policies' streams stepped TOGETHER (one batched LM step of len(policies) x tf rows per frame: a row's bits do not depend on its
    batch below 129 rows, tests/test_gpu_parity.py) over the same counter-based frames: one pass of vision batches and one weight pass
    per frame for all of them.  Returns per policy (scores [n, 3], (seq_len, seen_tokens)) and the first keep_first frames' embeddings."""
    tf, H, S, P = cfg.frame_num_tokens, cfg.lm.hidden_size, cfg.vision.image_size, len(policies)
    sts = [rt.open_stream(p, 2048, 32) for p in policies]
    rt.lm_step(sts, rt.embed_tokens(make_token_ids(20, cfg.lm.vocab_size, seed=101)).view(1, -1, H).expand(P, -1, -1).contiguous())
    pre = rt.embed_tokens(make_token_ids(35, cfg.lm.vocab_size, seed=100)).view(1, -1, H)
    scores = torch.empty((n_frames, P, 3), device="cuda")
    kept = []
    for i0 in range(0, n_frames, 32):
        n = min(32, n_frames - i0)
        emb = rt.visual_embed(_frames_batch(S, i0, n)).view(n, tf, H)
        if i0 < keep_first:
            kept.append(emb[:max(0, min(n, keep_first - i0))].cpu())
        for j in range(n):
            x = emb[j:j + 1] if i0 + j else torch.cat([pre, emb[:1]], 1)
            rt.lm_step(sts, x.expand(P, -1, -1).contiguous(), out=scores[i0 + j])
    sc = scores.cpu()
    info = [(st.get_seq_length(), st.seen_tokens) for st in sts]
    for st in sts:
        st.close()
    return [sc[:, k].contiguous() for k in range(P)], (torch.cat(kept) if kept else None), info


def _oracle_prefix(cfg, w_cpu, policy, emb, want_fp32=True):
    """Replay the first frames through the oracle (bf16, and fp32 for the noise band) on the same embeddings: the oracle's scores per
    frame, [n, 3] each (the fp32 ones None without want_fp32), and the cache length at the end."""
    from oracle.cache_policies import make_policy
    from oracle.qwen2_live import OracleLM, frame_scores
    H = cfg.lm.hidden_size
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ob = OracleLM(cfg.lm, w_cpu, torch.bfloat16)
    o32 = OracleLM(cfg.lm, w_cpu, torch.float32) if want_fp32 else None
    cb, c32 = make_policy(policy, 2048, 32), make_policy(policy, 2048, 32)
    q = ob.embed_tokens(make_token_ids(20, cfg.lm.vocab_size, seed=101)).view(1, -1, H)
    pre = ob.embed_tokens(make_token_ids(35, cfg.lm.vocab_size, seed=100)).view(1, -1, H)
    ob.step(q, cb)
    if o32 is not None:
        o32.step(q.float(), c32)
    sb, s32 = [], []
    for i in range(emb.shape[0]):
        x = emb[i:i + 1] if i else torch.cat([pre, emb[:1]], 1)
        sb.append(_rel(frame_scores(ob.step(x, cb)))[0])
        if o32 is not None:
            s32.append(_rel(frame_scores(o32.step(x.float(), c32)))[0])
    return torch.stack(sb), (torch.stack(s32) if s32 else None), cb.get_seq_length()


def test_config2_sink_and_sliding_10k_frame_streams_with_oracle_prefixes(bench_rt):
    """configs[2]: SinkCache(W=2048, sink=32) AND SlidingWindowCache(W=2048) over 10,000 frames at full model size, the two streams
    stepped together (one vision pass, one weight pass per frame; the sliding window turns over ~175 times, SinkCache evicts and
    re-rotates 1,980 keys x 28 layers on ~9,940 steps): bookkeeping exact, scores finite, the first 64 / 60 frames (through the first
    evictions, from frame ~56) replayed through the oracle, and a solo re-run of each policy's first 300 frames reproduces the batched
    run bit for bit (graph replay, ring state, re-used buffers, batched == solo).

    The oracle replays (CPU, ~1 s per frame) run on the main thread WHILE a worker thread drives the 10,000-frame HIP pass (the context
    is used by one thread at a time: the main thread touches it only before the worker starts and after it has joined).  On SURVEY
    8d's weights 28 untrained layers amplify every rounding chaotically, so |hip - oracle_fp32| and |oracle_bf16 - oracle_fp32| are samples
    of the same noise: the assertions compare MEANS and medians over >= 180 samples, and the SlidingWindowCache stream,
    replayed in bf16 only, is held to the band the SinkCache replay measured (the band is a property of weights and depth)."""
    import threading
    cfg, rt, w_cpu = bench_rt
    n, n_sink, n_slide = 10000, 64, 60
    tf, H, S = cfg.frame_num_tokens, cfg.lm.hidden_size, cfg.vision.image_size
    emb = rt.visual_embed(_frames_batch(S, 0, 64)).view(64, tf, H).cpu()      # the same 32-frame batches the stream below encodes
    box = {}

    def hip_pass():
        try:
            box["out"] = _long_streams(cfg, rt, ["default_sink", "sliding_window"], n, keep_first=64)
        except BaseException as e:                                             # re-raised on the main thread
            box["err"] = e
    worker = threading.Thread(target=hip_pass, name="hip-10k")
    worker.start()
    try:
        sb_k, s32_k, olen_k = _oracle_prefix(cfg, w_cpu, "default_sink", emb[:n_sink])
        sb_s, _, olen_s = _oracle_prefix(cfg, w_cpu, "sliding_window", emb[:n_slide], want_fp32=False)
    finally:
        worker.join()
    if "err" in box:
        raise box["err"]
    (sc_sink, sc_slide), emb_stream, ((len_k, seen_k), (len_s, seen_s)) = box["out"]
    assert torch.equal(emb_stream, emb)
    want_seen = 20 + 35 + n * tf
    assert torch.isfinite(sc_sink).all() and len_k == 2048 and seen_k == want_seen
    assert torch.isfinite(sc_slide).all() and len_s == 2048 and seen_s == want_seen
    print("sink 10k sha256", hashlib.sha256(sc_sink.numpy().tobytes()).hexdigest()[:16],
          "sliding 10k sha256", hashlib.sha256(sc_slide.numpy().tobytes()).hexdigest()[:16])
    assert not torch.equal(sc_sink[100:], sc_slide[100:])               # the policies really differ once they evict
    assert olen_k == 2048 and olen_s == 2048
    dev, band = (_rel(sc_sink[:n_sink]) - s32_k).abs(), (sb_k - s32_k).abs()
    print(f"SinkCache 10k stream, first {n_sink} frames: |hip - fp32| mean {dev.mean().item():.2e} median {dev.median().item():.2e} max "
          f"{dev.max().item():.2e}; |oracle_bf16 - fp32| mean {band.mean().item():.2e} median {band.median().item():.2e} max {band.max().item():.2e}")
    assert dev.numel() >= 180
    assert dev.mean().item() <= max(SCORE_TOL, 2.0 * band.mean().item()), (dev.mean().item(), band.mean().item())
    assert dev.median().item() <= max(SCORE_TOL, 2.0 * band.median().item()), (dev.median().item(), band.median().item())
    assert dev.max().item() <= max(SCORE_TOL, 5.0 * band.max().item()), (dev.max().item(), band.max().item())      # loose tail bound: one wrong score cannot hide in a mean
    # two bf16 evaluations are each one band away from the exact answer: their distance is held to sqrt(2) x 2 bands
    d_s = (_rel(sc_slide[:n_slide]) - sb_s).abs()
    print(f"SlidingWindowCache 10k stream, first {n_slide} frames: |hip - oracle_bf16| mean {d_s.mean().item():.2e} median {d_s.median().item():.2e} "
          f"max {d_s.max().item():.2e} (band from the SinkCache replay)")
    assert d_s.numel() >= 180
    assert d_s.mean().item() <= max(SCORE_TOL, 3.0 * band.mean().item()), (d_s.mean().item(), band.mean().item())
    # the sliding window's OWN fp32 band where it is free: until the window slides (frame ~56) both policies hold the same keys, so the fp32
    # oracle scores of the SinkCache replay ARE the sliding-window ones for the first 50 frames; the frames behind them get the tail bound
    dev_s50, band50 = (_rel(sc_slide[:50]) - s32_k[:50]).abs(), band[:50]
    assert dev_s50.mean().item() <= max(SCORE_TOL, 2.0 * band50.mean().item()), (dev_s50.mean().item(), band50.mean().item())
    assert dev_s50.max().item() <= max(SCORE_TOL, 5.0 * band50.max().item()), (dev_s50.max().item(), band50.max().item())
    assert d_s.max().item() <= max(SCORE_TOL, 7.0 * band.max().item()), (d_s.max().item(), band.max().item())       # bf16 vs bf16: sqrt(2) wider
    # before the window slides (frame < 56) the two policies hold the same keys: identical scores
    assert torch.equal(sc_sink[:50], sc_slide[:50])
    for policy, sc in (("default_sink", sc_sink), ("sliding_window", sc_slide)):
        (sc2,), _, _ = _long_streams(cfg, rt, [policy], 300, keep_first=0)
        assert torch.equal(sc2, sc[:300]), policy


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]: the per-GPU share of the 64-stream job: 8 streams batched, full 28 layers
# ---------------------------------------------------------------------------------------------------------------------
def test_config3_eight_streams_full_depth_against_oracle(bench_rt):
    from oracle.cache_policies import make_policy
    from oracle.qwen2_live import OracleLM, frame_scores
    cfg, rt, w_cpu = bench_rt
    tf, H, S, B = cfg.frame_num_tokens, cfg.lm.hidden_size, cfg.vision.image_size, 8
    frames = torch.stack([make_frames(3, S, seed=1000 + s) for s in range(B)])             # stream id -> seed
    emb = rt.visual_embed(frames.view(-1, 3, S, S).cuda()).view(B, 3, tf, H)
    q = rt.embed_tokens(make_token_ids(20, cfg.lm.vocab_size, seed=101)).view(1, -1, H).expand(B, -1, -1).contiguous()
    pre = rt.embed_tokens(make_token_ids(35, cfg.lm.vocab_size, seed=100)).view(1, -1, H).expand(B, -1, -1)
    steps = [q, torch.cat([pre, emb[:, 0]], 1).contiguous(), emb[:, 1].contiguous(), emb[:, 2].contiguous()]
    sts = [rt.open_stream("default_sink", 2048, 32) for _ in range(B)]
    batched = [rt.lm_step(sts, x).cpu() for x in steps]                                    # M = 160, 568, 288, 288 rows
    for s in sts:
        s.close()
    # every stream alone gives the same bits (streams never mix; split-K slices do not depend on M)
    for b in (0, 5):
        st = rt.open_stream("default_sink", 2048, 32)
        for i, x in enumerate(steps):
            assert torch.equal(rt.lm_step([st], x[b:b + 1].contiguous()).cpu()[0], batched[i][b]), (b, i)
        st.close()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ob, o32 = OracleLM(cfg.lm, w_cpu, torch.bfloat16), OracleLM(cfg.lm, w_cpu, torch.float32)
    dev, band = [], []
    for b in (0, 3, 7):
        cb, c32 = make_policy("default_sink", 2048, 32), make_policy("default_sink", 2048, 32)
        for i, x in enumerate(steps):
            xb = x[b:b + 1].cpu()
            sb, s32 = _rel(frame_scores(ob.step(xb, cb))), _rel(frame_scores(o32.step(xb.float(), c32)))
            if i:
                dev.append((_rel(batched[i][b:b + 1]) - s32).abs()[0])
                band.append((sb - s32).abs()[0])
    dev, band = torch.stack(dev), torch.stack(band)
    assert dev.median().item() <= max(SCORE_TOL, 2.0 * band.median().item()), (dev.median().item(), band.median().item())
    assert dev.max().item() <= max(SCORE_TOL, 3.0 * band.max().item()), (dev.max().item(), band.max().item())


# ---------------------------------------------------------------------------------------------------------------------
# configs[4]: the TVSum-shaped evaluation path with score-vector parity
# ---------------------------------------------------------------------------------------------------------------------
def test_config4_tvsum_shaped_eval_score_vectors_match_the_oracle_driver():
    from aha_amd.live_infer import LiveInferForDemo, round_numbers
    from aha_amd.postproc import evaluate_f1, evaluate_tvsum, fuse_scores
    from aha_amd.runtime import Runtime
    from aha_amd.tokenization import SyntheticChatTokenizer
    cfg = preset("tiny")
    w = make_weights(cfg, dtype=torch.bfloat16, jitter=True)
    rt = Runtime(cfg, w, max_step_tokens=128, max_vit_frames=32, max_positions=4096)
    tok = SyntheticChatTokenizer(cfg.lm.vocab_size)
    params = dict(alpha=0.0, beta=-1.0, epsilon=-5.0, uncertainty_threshold=0.04)          # outputs/grid_search_params.json "tvsum"
    lengths = np.random.default_rng(0).integers(40, 90, 4)
    q = "Which moments of this video are the highlights?"
    qids = tok.apply_chat_template([{"role": "user", "content": q}], add_stream_prompt=True)
    drv, ob, o32 = _driver_pair(cfg, w, rt, "default_sink", 512, 8, tok)
    pred, pred_o, gt = {}, {}, {}
    worst, band = 0.0, 0.0
    for v, n in enumerate(lengths):
        frames = make_frames(int(n), cfg.vision.image_size, seed=500 + v)
        rows = []
        for d in (drv, ob, o32):
            d.reset()
            d.set_fps(fps=1)
            if d is drv:
                d.input_query_stream([{"role": "user", "content": q, "time": 0}])
            else:
                d.input_query_stream([{"role": "user", "time": 0, "ids": qids}])
            d.input_video_stream(frames)
            d.inference()
            rows.append(d.debug_data_list)
        assert len(rows[0]) == int(n)
        k = f"synth_{v:03d}"
        pred[k] = fuse_scores(round_numbers(rows[0], 3), **params)
        pred_o[k] = fuse_scores(round_numbers(rows[1], 3), **params)
        gt[k] = np.random.default_rng(10_000 + v).integers(1, 6, (20, int(n))).mean(0) / 5.0
        for key in KEYS + ("uncertainty_score",):
            a = np.array([r[key] for r in rows[0]]); b = np.array([r[key] for r in rows[1]]); c = np.array([r[key] for r in rows[2]])
            if key == "uncertainty_score":
                a, b, c = np.log(a), np.log(b), np.log(c)
            worst, band = max(worst, np.abs(a - c).max()), max(band, np.abs(b - c).max())
    assert worst <= max(SCORE_TOL, 2.0 * band), (worst, band)                              # score vectors, every frame of every video
    # metrics are functions of the score vectors: both sets go through the same ported post-processing
    m_hip, m_or = evaluate_tvsum(gt, pred), evaluate_tvsum(gt, pred_o)
    assert all(np.isfinite(x) for x in m_hip) and np.isfinite(evaluate_f1(gt, pred))
    print("tvsum-shaped metrics HIP", [round(float(x), 4) for x in m_hip], "oracle", [round(float(x), 4) for x in m_or])
    ticks = LiveInferForDemo.find_ticks(None, pred["synth_000"], fps=1)
    assert all(0 <= t < lengths[0] for t in ticks)
    rt.close()


# ---------------------------------------------------------------------------------------------------------------------
# response generation on the HIP path
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("which", ["tiny", "tiny128"])
def test_response_generation_on_the_hip_path_matches_the_oracle_driver(which):
    """Responses are triggered by the running score sum (stream_end_score_sum_threshold) with repetition_penalty 1.2, as the
    reference's benchmark drivers do (test/inference.py:306-319, models/modeling_live.py:64-90): same trigger times, same
    token ids as the bf16 oracle driver (greedy loop, penalty over the growing generated_token_ids list, EOS / length stop,
    last_ids hand-back into the next frame's prompt), and the scores of the frames that FOLLOW a response agree within the
    band rule - they depend on every generated token sitting in the cache."""
    from aha_amd.runtime import Runtime
    from aha_amd.tokenization import SyntheticChatTokenizer

    class RecordingTokenizer(SyntheticChatTokenizer):
        def __init__(self, *a):
            super().__init__(*a)
            self.decoded = []

        def decode(self, ids, **kw):
            self.decoded.append([int(i) for i in ids])
            return super().decode(ids, **kw)

    cfg = preset(which)
    w = make_weights(cfg, dtype=torch.bfloat16, jitter=True)
    rt = Runtime(cfg, w, max_step_tokens=128, max_vit_frames=32, max_positions=8192)
    frames = make_frames(24, cfg.vision.image_size, seed=77)
    q = "describe what happens"
    # pick a threshold whose trigger decisions keep a margin in the oracle run (decisions must not hinge on bf16 noise)
    chosen = None
    for thr in (2.7, 3.1, 3.6, 4.2, 2.2, 5.0):
        tok = RecordingTokenizer(cfg.lm.vocab_size)
        qids = tok.apply_chat_template([{"role": "user", "content": q}], add_stream_prompt=True)
        drv, ob, o32 = _driver_pair(cfg, w, rt, None, 2048, 0, tok, stream_end_score_sum_threshold=thr, repetition_penalty=1.2,
                                    max_new_tokens=6)
        ob.input_video_stream(frames)
        ob.input_query_stream([{"role": "user", "time": 0, "ids": qids}])
        margins, orig = [], ob._encode_frame

        def spy():
            vs, unc = orig()
            margins.append(abs(ob.stream_end_score_sum + sum(v for k, v in vs.items() if k in ob.score_heads) - thr))
            return vs, unc
        ob._encode_frame = spy
        resp_o = ob.inference()
        if min(margins) > 0.05 and len(resp_o) >= 3:
            chosen = thr
            break
    assert chosen is not None, "no threshold with a safe decision margin"
    drv.input_video_stream(frames)
    drv.input_query_stream([{"role": "user", "content": q, "time": 0}])
    resp = [r for r in drv.inference() if r["role"] == "assistant"]
    assert [r["time"] for r in resp] == [r["time"] for r in resp_o], (resp, resp_o)
    hip_ids, oracle_ids = tok.decoded, [r["content"] for r in resp_o]
    assert len(hip_ids) == len(oracle_ids) >= 3
    same = [a == b for a, b in zip(hip_ids, oracle_ids)]
    print(f"{which}: threshold {chosen}, {len(resp)} responses at t={[r['time'] for r in resp]}; token sequences identical to the free-running oracle: {same}")
    assert same[0] and sum(same) >= len(same) - 2, (hip_ids, oracle_ids)
    # Where a sequence departs, the departure must be a genuine bf16 near-tie.  Replay the oracle driver FORCED onto the HIP
    # path's tokens (so both caches hold the same history throughout) and require every HIP token to be an argmax of the
    # oracle's penalised logits up to 2 bf16 ulps of the maximum.
    from ulp import bf16_ulp
    worst_gap = [0.0]

    def forced_driver(dtype):
        _, o_b, o_f = _driver_pair(cfg, w, rt, None, 2048, 0, tok, stream_end_score_sum_threshold=chosen, repetition_penalty=1.2, max_new_tokens=6)
        o = o_b if dtype == torch.bfloat16 else o_f
        forced = [list(x) for x in hip_ids]

        def gen():
            ids = forced.pop(0)
            o.last_ids = o._added_stream_generation_ids
            emb = o.lm.embed_tokens(o.last_ids)
            for tokid in ids:
                out = o.lm.step(emb, o.past_key_values, want_logits=True)
                logits = out["logits"][:, -1, :]
                if o.generated_token_ids:
                    idx = torch.tensor(o.generated_token_ids)[None]
                    sc = torch.gather(logits, 1, idx)
                    sc = torch.where(sc < 0, sc * 1.2, sc / 1.2)
                    logits = logits.scatter(1, idx, sc)
                if dtype == torch.bfloat16:
                    top = logits.max().item()
                    worst_gap[0] = max(worst_gap[0], (top - logits[0, tokid].item()) / bf16_ulp(torch.tensor(abs(top) + 1e-30)).item())
                if tokid != o.eos_token_id:
                    o.generated_token_ids.append(tokid)
                emb = o.lm.embed_tokens(torch.tensor([[tokid]]))
            o.last_ids = torch.tensor([[ids[-1]]])
            o.last_role = "assistant"
            return ids
        o._generate_response = gen
        o.input_video_stream(frames)
        o.input_query_stream([{"role": "user", "time": 0, "ids": qids}])
        o.inference()
        return o
    fb, f32 = forced_driver(torch.bfloat16), forced_driver(torch.float32)
    print(f"{which}: largest gap between the oracle's best penalised logit and the HIP path's token: {worst_gap[0]:.2f} bf16 ulp")
    assert worst_gap[0] <= 2.0
    assert drv.generated_token_ids == fb.generated_token_ids
    assert drv.past_key_values.get_seq_length() == fb.past_key_values.get_seq_length()
    assert [d["time"] for d in drv.debug_data_list] == [d["time"] for d in fb.debug_data_list]
    # same token history in all three caches: every frame's scores (before and after each response) within the band rule
    band = max(abs(a[k] - b[k]) for a, b in zip(fb.debug_data_list, f32.debug_data_list) for k in KEYS)
    d32 = max(abs(a[k] - b[k]) for a, b in zip(drv.debug_data_list, f32.debug_data_list) for k in KEYS)
    assert d32 <= max(SCORE_TOL, 3.0 * band), (d32, band)
    rt.close()


def test_all_position_logits_match_last_position_and_oracle():
    """outputs.logits [B,T,V] of the reference forward (video_head_live_llava_qwen.py:175), opt-in on the model mirror."""
    from aha_amd.cache import DynamicCache
    from aha_amd.model import LiveLlavaModel
    from aha_amd.runtime import Runtime
    from oracle.cache_policies import GrowingPolicy
    from oracle.qwen2_live import OracleLM
    from ulp import ulp_error
    cfg = preset("tiny")
    w = make_weights(cfg, dtype=torch.bfloat16, jitter=True)
    rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=2, max_positions=1024)
    model = LiveLlavaModel(rt, all_position_logits=True)
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(1, 9, cfg.lm.hidden_size, generator=g) * 0.5).bfloat16()
    out = model(inputs_embeds=x.cuda(), past_key_values=DynamicCache(), use_cache=True, return_dict=True)
    lg = out.logits
    assert lg.shape == (1, 9, cfg.lm.vocab_size)
    last, _ = rt.logits_last(1)
    assert torch.equal(lg[:, -1], last)
    # teacher-forced: lm_head on the HIP path's own final hidden rows, exact arithmetic, <= 1 ulp (0.5 + fp32 accumulation)
    hid = rt.last_hidden_all(1, 9)[0]
    exact = hid.double() @ w["lm_head.weight"].cuda().double().T
    e = ulp_error(lg[0], exact, slack=1e-5 * (hid.float().abs() @ w["lm_head.weight"].cuda().float().abs().T))
    assert e.max().item() <= 0.5 + 1e-6
    want = OracleLM(cfg.lm, w, torch.bfloat16).step(x, GrowingPolicy(), want_logits=True)["logits"]
    assert (lg.cpu() - want).abs().max().item() <= 0.08
    # caller-provided destinations (bench.py's score table and logits buffer): same values, written in place
    st = rt.open_stream(None, capacity=64)
    ref_scores = rt.lm_step([st], x.cuda()).clone()
    ref_logits = rt.logits_all(1, 9).clone()
    st.close()
    st = rt.open_stream(None, capacity=64)
    table = torch.full((3, 1, 3), -7.0, device="cuda")
    buf = torch.empty((9, cfg.lm.vocab_size), dtype=torch.float32, device="cuda")
    got = rt.lm_step([st], x.cuda(), out=table[1])
    assert got.data_ptr() == table[1].data_ptr() and torch.equal(table[1], ref_scores)
    assert (table[0] == -7.0).all() and (table[2] == -7.0).all()
    assert torch.equal(rt.logits_all(1, 9, out=buf).view(9, -1), ref_logits.view(9, -1)) and torch.equal(buf, ref_logits.view(9, -1))
    st.close()
    rt.close()


# ---------------------------------------------------------------------------------------------------------------------
# checkpoint loader on the GPU
# ---------------------------------------------------------------------------------------------------------------------
def test_checkpoint_with_lora_adapter_loads_into_the_runtime(tmp_path):
    """A synthetic llava-ov-named safetensors checkpoint + PEFT adapter -> load_checkpoint -> Runtime.  With B = 0 the
    scores equal those of the plain weights bit for bit; with a real adapter the merged weight is the correctly rounded
    W + (alpha/r) B A and the scores follow the oracle run with UNMERGED adapters W x + (alpha/r) B (A x)
    (models/modeling_live.py:171-179 keeps PEFT unmerged) within the band rule."""
    from safetensors.torch import save_file
    from aha_amd.checkpoint import load_checkpoint
    from aha_amd.runtime import Runtime
    from oracle.cache_policies import make_policy
    from oracle.qwen2_live import OracleLM, frame_scores
    from ulp import ulp_error
    cfg = preset("tiny")
    w = make_weights(cfg, dtype=torch.bfloat16, jitter=True)

    def hf_key(name):
        if name.startswith("vision."):
            return "model.vision_tower.vision_tower.vision_model." + name[len("vision."):]
        return "model." + name if name.startswith("mm_projector.") else name
    base_dir = tmp_path / "base"
    base_dir.mkdir()
    save_file({hf_key(k): v.contiguous() for k, v in w.items()}, str(base_dir / "model-00001-of-00001.safetensors"))
    r, alpha = 4, 8.0
    g = torch.Generator().manual_seed(0)
    mods = [f"model.layers.{l}.{m}" for l in range(cfg.lm.num_hidden_layers)
            for m in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")]
    adapters = {}
    for zero_b, name in ((True, "lora0"), (False, "lora1")):
        d = tmp_path / name
        d.mkdir()
        ad = {}
        for mod in mods:                                                   # r=16 on all 7 projections upstream (arguments_live.py:15-17)
            W = w[mod + ".weight"]
            A = (torch.randn(r, W.shape[1], generator=g) * 0.05).bfloat16()
            B = torch.zeros(W.shape[0], r).bfloat16() if zero_b else (torch.randn(W.shape[0], r, generator=g) * 0.05).bfloat16()
            ad[f"base_model.model.{mod}.lora_A.weight"], ad[f"base_model.model.{mod}.lora_B.weight"] = A, B
            if not zero_b:
                adapters[mod + ".weight"] = (A, B, alpha / r)
        save_file(ad, str(d / "adapter_model.safetensors"))
        json.dump({"r": r, "lora_alpha": alpha}, open(d / "adapter_config.json", "w"))
    gx = torch.Generator().manual_seed(9)
    xs = [(torch.randn(1, T, cfg.lm.hidden_size, generator=gx) * 0.5).bfloat16() for T in (9, 5, 5, 5)]

    def run(weights):
        rt = Runtime(cfg, weights, max_step_tokens=64, max_vit_frames=2, max_positions=1024)
        st = rt.open_stream("default_sink", 16, 2)
        out = torch.cat([rt.lm_step([st], x.cuda()).cpu() for x in xs])
        emb = rt.visual_embed(make_frames(2, cfg.vision.image_size, seed=1).cuda()).cpu()
        st.close()
        rt.close()
        return out, emb
    plain, plain_e = run(w)
    w0 = load_checkpoint(str(base_dir), str(tmp_path / "lora0"))
    assert set(w0) == set(w)
    got0, got0_e = run(w0)
    assert torch.equal(got0, plain) and torch.equal(got0_e, plain_e)
    w1 = load_checkpoint(str(base_dir), str(tmp_path / "lora1"))
    for name in ("model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight"):
        A, B, sc = adapters[name]
        exact = w[name].double() + sc * (B.double() @ A.double())
        assert ulp_error(w1[name], exact, slack=1e-6 * exact.abs() + 1e-12).max().item() <= 0.5 + 1e-6      # merged in fp32, rounded once
    got1, _ = run(w1)
    assert not torch.equal(got1, plain)
    ob, o32 = OracleLM(cfg.lm, w, torch.bfloat16), OracleLM(cfg.lm, w, torch.float32)
    ob.attach_lora(adapters)
    o32.attach_lora(adapters)
    cb, c32 = make_policy("default_sink", 16, 2), make_policy("default_sink", 16, 2)
    dev = band = 0.0
    for i, x in enumerate(xs):
        sb, s32 = _rel(frame_scores(ob.step(x, cb))), _rel(frame_scores(o32.step(x.float(), c32)))
        dev = max(dev, (_rel(got1[i:i + 1]) - s32).abs().max().item())
        band = max(band, (sb - s32).abs().max().item())
    assert dev <= max(SCORE_TOL, 2.0 * band), (dev, band)


# ---------------------------------------------------------------------------------------------------------------------
# multi-GPU plumbing that a 1-GPU box can exercise
# ---------------------------------------------------------------------------------------------------------------------
def test_bench_launches_ranks_itself_and_the_rccl_path_runs():
    """(1) `bench.py --gpus 2 --backend gloo` on ONE GPU: the launcher starts two ranks that share the card, each scores its
    own streams, the score rows are all-gathered (gloo): n_gpus = 2, ranks_seen = 2, value counts both ranks' frames.
    (2) one rank with --force-dist on the RCCL backend: barrier / all-gather / all-reduce through RCCL.
    (3) the C-ABI collective (aha_comm_* + aha_allgather_scores) in ITS OWN process group (tools/abi_allgather_check.py, launched
    through torch.distributed.run): the same rows as torch.distributed's all_gather; a stall would be that child's non-zero exit."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--preset", "tiny", "--steps", "2", "--warmup", "1", "--frames", "4", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"] + common,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert out["n_gpus"] == 2 and out["distributed"]["ranks_seen"] == 2 and out["value"] > 0
    assert out["config"]["frames_per_step"] == 8 and out["distributed"]["allgather_us"] > 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist"] + common, capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    d = out["distributed"]
    assert out["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["backend"].startswith("RCCL")
    assert "step" in out["roofline"] and "lm_step" in out["roofline"] and "vision" in out["roofline"]
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # the launcher and its rank(s) run in a process group of their own: if the 400 s backstop ever fires (every rank also carries its own
    # 180 s deadline that works inside C calls), the WHOLE group is killed - no rank is left holding the GPU - and nothing is retried
    import signal
    p = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "tools", "abi_allgather_check.py")],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        so, se = p.communicate(timeout=400)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        so, se = p.communicate()
        raise AssertionError(("abi_allgather_check timed out; process group killed", so[-2000:], se[-3000:]))
    assert p.returncode == 0, (so[-2000:], se[-3000:])
    chk = json.loads([ln for ln in so.splitlines() if ln.startswith('{"ok"')][-1])
    assert chk["ok"] is True and chk["ranks"] == 1 and chk["allgather_us"] > 0, chk


def test_bench_two_ranks_emit_the_configs3_datum():
    """`bench.py --gpus N` with N > 1 also measures the one multi-GPU configuration BASELINE.json names (configs[3]: 8 streams per GPU on
    SinkCache, score rows all-gathered): rehearsed here with two ranks sharing the card over gloo at the FULL model size (short run), so
    that the driver's scaling run does not meet that code path for the first time."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                        "--frames", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    e = out["eight_stream_sink"]
    assert out["n_gpus"] == 2 and out["distributed"]["ranks_seen"] == 2 and out["value"] > 0
    assert e["n_gpus"] == 2 and e["streams_total"] == 16 and e["frames_per_s"] > 0 and e["allgather_us"] > 0
    assert out["sink_w2048"] is None and out["growing_600"] is None          # single-GPU secondaries stay out of a multi-rank line
