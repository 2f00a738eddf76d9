"""The north star's tolerance, asserted FLAT and free-running: per-frame scores within 1e-3 of the reference arithmetic
(the bf16 oracle) at full width and depth, frames in -> scores out.

What the reference observes per frame is three floats (test/inference.py:222-227; heads at
models/live_llava/video_head_live_llava_qwen.py:185-188).  With SURVEY.md 8d's plain normal(0, 0.02) weights 28 untrained
layers are a chaotic map (two bf16 evaluations of the REFERENCE arithmetic differ by 1e-2 in score), so tests/test_gpu_parity.py
and tests/test_gpu_configs.py can only hold the HIP path to the oracle's own noise band there.  Here the weights are the
``regime="stable"`` set of aha_amd.synth (depth-scaled residual branches, unit-scale stream, small head logits - same
shapes, dtypes and arithmetic), in which the reference arithmetic itself is stable (tests/stable_regime_check.py,
profiles/r04_stable_regime_cpu.json), and the bound is the flat one:

    |hip - oracle_bf16| <= 1e-3   on all three scores, every frame, no band.

Both sides run END TO END and free: the HIP path from uint8 frames through its own tower / projector / pool / 28-layer LM
on its own cache, the oracle from the same uint8 frames through its own tower and LM - nothing is teacher-forced or shared
but the frames and the weights.  Policies: TrulyStaticCache (configs[1]), SinkCache through evictions and re-rotations and
SlidingWindowCache (configs[2]'s policies at a window the frames overflow), the growing cache, 8 streams batched (configs[3]); and configs[4]'s driver flow on two short
videos (score vectors through LiveInferForBenchmark vs the oracle driver, then the ported metrics).

Heads: `aha_amd.synth.calibrated_heads` - closed-form heads aligned with the three leading frame-to-frame directions of the
final hidden state on 24 calibration frames (a trained head reads a coherent feature; a random direction reads signal and
rounding noise through the same incoherent sum, which pins noise / spread at ~0.06 for ANY implementation of the reference
arithmetic, the reference's own sdpa-vs-eager pair included: profiles/r04_stable_regime_cpu*.json,
profiles/r04_flat_parity_random_heads*.json).  Frames carry three coherent frame-level features (make_frames(tint=True)).
"""
import json
import os
import sys

import pytest
import torch

import aha_amd  # noqa: F401
from aha_amd.config import preset
from aha_amd.synth import make_frames, make_token_ids, make_weights

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ulp import rms, ulp_error  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAT_TOL = 1e-3                 # BASELINE.json north_star: "within 1e-3 (bf16) on identical frame sequences"
HIDDEN_REL_L2_TOL = 0.05        # scale-free, unlike the scores; measured on MI355X: max 0.031-0.036 per policy, median 0.021-0.024 (profiles/r05_flat_parity_stats.json)
N_FRAMES = 84                   # 20 + 35 + 84 x 36 = 3,079 keys on the growing cache (an oracle prefix through >= 3,000 keys)
STATS = {}


def _dump():
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(STATS, open(os.path.join(ROOT, "gpurun_out", "flat_parity_stats.json"), "w"), indent=1)
    except OSError:
        pass


@pytest.fixture(scope="module")
def stable():
    from aha_amd.runtime import Runtime
    from oracle.qwen2_live import OracleLM
    from oracle.vision_tower import OracleVision
    from aha_amd.synth import calibrated_heads
    from oracle.cache_policies import GrowingPolicy
    cfg = preset("bench")
    wd = make_weights(cfg, device="cuda", dtype=torch.bfloat16, regime="stable")
    w = {k: v.cpu() for k, v in wd.items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    tf, H, S = cfg.frame_num_tokens, cfg.lm.hidden_size, cfg.vision.image_size
    ov, olm = OracleVision(cfg, w, torch.bfloat16), OracleLM(cfg.lm, w, torch.bfloat16)
    # calibrate the heads on the reference arithmetic (24 frames of another seed, each alone on an empty cache)
    cal = make_frames(24, S, seed=777, tint=True)
    emb_cal = torch.cat([ov.visual_embed(cal[i:i + 8]) for i in range(0, 24, 8)]).view(24, tf, H)
    hid = torch.stack([olm.step(emb_cal[i:i + 1], GrowingPolicy())["hidden"][0, -1] for i in range(24)])
    heads = calibrated_heads(hid.float())
    w.update(heads)
    olm.w.update(heads)
    wd.update({k: v.cuda() for k, v in heads.items()})
    rt = Runtime(cfg, wd, max_step_tokens=640, max_vit_frames=32)
    del wd
    torch.cuda.empty_cache()
    frames = make_frames(N_FRAMES, S, seed=4242, tint=True)
    emb_hip = rt.visual_embed(frames.cuda()).view(N_FRAMES, tf, H)
    emb_ref = torch.cat([ov.visual_embed(frames[i:i + 8]) for i in range(0, N_FRAMES, 8)]).view(N_FRAMES, tf, H)
    del ov
    yield cfg, rt, olm, emb_hip, emb_ref, w
    rt.close()


def test_stable_regime_embeddings_are_unit_scale_and_close(stable):
    """The frame embeddings the two sides feed their LMs: unit scale (the regime's premise) and a few bf16 ulps apart at the
    tensor's scale after 24 free-running tower layers + projector + pooling."""
    cfg, rt, olm, emb_hip, emb_ref, _ = stable
    want = emb_ref.float()
    e = ulp_error(emb_hip.float().cpu(), want, floor=rms(want))
    STATS["embeddings"] = {"rms": rms(want), "max_ulp_at_scale": e.max().item(), "p999_ulp": e.flatten().quantile(0.999).item(),
                           "frac_bit_equal": (emb_hip.cpu() == emb_ref).float().mean().item()}
    print("stable-regime frame embeddings:", STATS["embeddings"])
    _dump()
    assert 0.3 <= rms(want) <= 3.0
    assert e.max().item() <= 16.0


@pytest.mark.parametrize("policy,window,sink", [("static", 2048, 0), ("default_sink", 1024, 32), ("sliding_window", 1024, 0), (None, 0, 0)])
def test_free_running_scores_within_flat_1e3_of_the_bf16_oracle(stable, policy, window, sink):
    from oracle.cache_policies import make_policy
    from oracle.qwen2_live import frame_scores
    cfg, rt, olm, emb_hip, emb_ref, _ = stable
    H, V = cfg.lm.hidden_size, cfg.lm.vocab_size
    q_ids, pre_ids = make_token_ids(20, V, seed=101), make_token_ids(35, V, seed=100)
    # HIP side: query turn, then system prompt + frame 0, then a frame per step (the loop of test/inference.py:283-335)
    st = rt.open_stream(policy, window or 2048, sink, capacity=4096)
    rt.lm_step([st], rt.embed_tokens(q_ids).view(1, -1, H))
    pre = rt.embed_tokens(pre_ids).view(1, -1, H)
    got = torch.empty((N_FRAMES, 3), device="cuda")
    hid_hip = torch.empty((N_FRAMES, H), dtype=torch.bfloat16, device="cuda")
    for i in range(N_FRAMES):
        x = emb_hip[i:i + 1] if i else torch.cat([pre, emb_hip[:1]], 1)
        rt.lm_step([st], x.contiguous(), out=got[i:i + 1])
        hid_hip[i] = rt.last_hidden_all(1, x.shape[1])[0, -1]      # final normalised hidden state of the last token (what the heads read)
    got, hid_hip = got.cpu(), hid_hip.float().cpu()
    seq_hip = st.get_seq_length()
    st.close()
    # oracle side, on ITS OWN embeddings
    pol = make_policy(policy, window or 2048, sink)
    olm.step(olm.embed_tokens(q_ids), pol)
    opre = olm.embed_tokens(pre_ids)
    want, hid_ref = [], []
    for i in range(N_FRAMES):
        x = emb_ref[i:i + 1] if i else torch.cat([opre, emb_ref[:1]], 1)
        o = olm.step(x, pol)
        want.append(frame_scores(o)[0])
        hid_ref.append(o["hidden"][0, -1].float())
    want, hid_ref = torch.stack(want), torch.stack(hid_ref)
    d = (got.double() - want.double()).abs()
    # a bound that does not shrink with the head scale - the FINAL HIDDEN STATE of the scored token itself, free-running,
    # as a relative L2 distance per frame (scale-free) and in bf16 ulps at the tensor's rms scale.  A wrong eviction row or a
    # mis-rotated key changes the hidden state of every later frame by tens of percent; two valid bf16 evaluations differ by the
    # ~1 % per-channel rounding noise of a 56-add residual stream.
    rel = ((hid_hip - hid_ref).norm(dim=1) / hid_ref.norm(dim=1))
    e_h = ulp_error(hid_hip, hid_ref, floor=rms(hid_ref))
    key = str(policy)
    STATS[key] = {"frames": N_FRAMES, "window": window, "sink": sink, "seq_len": seq_hip,
                  "max_abs_diff": d.max(0).values.tolist(), "median_abs_diff": d.median(0).values.tolist(),
                  "frac_within_1e-3": (d <= FLAT_TOL).float().mean().item(),
                  "score_std": want.double().std(0).tolist(), "score_min": want.min(0).values.tolist(), "score_max": want.max(0).values.tolist(),
                  "hidden_rel_l2_max": rel.max().item(), "hidden_rel_l2_median": rel.median().item(),
                  "hidden_ulp_at_scale_max": e_h.max().item(), "hidden_ulp_at_scale_p999": e_h.flatten().quantile(0.999).item()}
    print(f"flat parity [{key}]:", json.dumps(STATS[key]))
    _dump()
    assert seq_hip == pol.get_seq_length()
    if policy in ("default_sink", "sliding_window"):
        assert seq_hip == window                                   # the window filled and evicted (SinkCache: re-rotations ran)
    assert want.double().std(0).min().item() >= 0.02, "degenerate scores: the regime must keep a real spread"
    assert d.max().item() <= FLAT_TOL, (d.max(0).values.tolist(), d.argmax(0).tolist())
    assert rel.max().item() <= HIDDEN_REL_L2_TOL, (rel.max().item(), rel.argmax().item())


def test_eight_streams_batched_within_flat_1e3_of_the_bf16_oracle(stable):
    """configs[3]'s per-GPU share at full size: 8 independent streams batched into every LM step (M = 288 rows per weight pass: the
    mid-M GEMM kernels and the multi-stream attention kernel), SinkCache W = 384 / sink 16 so that every stream evicts and re-rotates from
    its 10th frame on, 14 frames per stream, each stream held to the flat 1e-3 against its OWN oracle run on the oracle's own embeddings."""
    from oracle.cache_policies import make_policy
    from oracle.qwen2_live import frame_scores
    cfg, rt, olm, emb_hip, emb_ref, _ = stable
    H, V = cfg.lm.hidden_size, cfg.lm.vocab_size
    B, n, W, S = 8, 14, 384, 16
    q_ids, pre_ids = make_token_ids(20, V, seed=101), make_token_ids(35, V, seed=100)
    idx = [[(7 * s + 3 * i) % N_FRAMES for i in range(n)] for s in range(B)]            # stream s sees its own frame sequence
    sts = [rt.open_stream("default_sink", W, S) for _ in range(B)]
    rt.lm_step(sts, rt.embed_tokens(q_ids).view(1, -1, H).expand(B, -1, -1).contiguous())
    pre = rt.embed_tokens(pre_ids).view(1, -1, H).expand(B, -1, -1)
    got = torch.empty((n, B, 3), device="cuda")
    for i in range(n):
        x = torch.stack([emb_hip[idx[s][i]] for s in range(B)])
        if i == 0:
            x = torch.cat([pre, x], 1)
        rt.lm_step(sts, x.contiguous(), out=got[i])
    got = got.cpu()
    assert all(st.get_seq_length() == W for st in sts)
    for st in sts:
        st.close()
    worst = 0.0
    opre = olm.embed_tokens(pre_ids)
    for s in range(B):
        pol = make_policy("default_sink", W, S)
        olm.step(olm.embed_tokens(q_ids), pol)
        for i in range(n):
            x = emb_ref[idx[s][i]][None]
            if i == 0:
                x = torch.cat([opre, x], 1)
            want = frame_scores(olm.step(x, pol))[0]
            worst = max(worst, (got[i, s].double() - want.double()).abs().max().item())
        assert pol.get_seq_length() == W
    STATS["eight_streams_batched"] = {"streams": B, "frames_per_stream": n, "window": W, "sink": S, "max_abs_diff": worst}
    print("flat parity [8 streams batched]:", json.dumps(STATS["eight_streams_batched"]))
    _dump()
    assert worst <= FLAT_TOL, worst


def test_growing_cache_to_600_frames_bookkeeping_and_reproducibility(stable):
    """past_key_values=None (test/inference.py:154-155) for SURVEY.md 8d config 2's 600 frames: the cache grows to 20 + 35 + 600 x 36 =
    21,655 keys (attention over up to 64 key splits of 256-384 keys, 1.24 GB of K/V per step at the end).  The first 84 frames are the ones the test
    above holds to the flat 1e-3 against the oracle; here the stream runs on to 600 frames (embeddings recycled): exact bookkeeping,
    finite scores that keep their spread, and a second run reproduces every bit (graph replay across ~68 key-split shapes - the graph cache holds 128 - and re-used
    partial buffers).  The attention arithmetic at that length has its own flat-bound test (tests/test_gpu_kernels.py)."""
    cfg, rt, olm, emb_hip, emb_ref, _ = stable
    H, V, tf = cfg.lm.hidden_size, cfg.lm.vocab_size, cfg.frame_num_tokens
    q_ids, pre_ids = make_token_ids(20, V, seed=101), make_token_ids(35, V, seed=100)
    runs = []
    for rep in range(2):
        st = rt.open_stream(None, capacity=22016)
        rt.lm_step([st], rt.embed_tokens(q_ids).view(1, -1, H))
        pre = rt.embed_tokens(pre_ids).view(1, -1, H)
        got = torch.empty((600, 3), device="cuda")
        for i in range(600):
            x = emb_hip[i % N_FRAMES][None] if i else torch.cat([pre, emb_hip[:1]], 1)
            rt.lm_step([st], x.contiguous(), out=got[i:i + 1])
        runs.append(got.cpu())
        assert st.get_seq_length() == 20 + 35 + 600 * tf == st.seen_tokens
        st.close()
    assert torch.isfinite(runs[0]).all() and torch.equal(runs[0], runs[1])
    assert runs[0][300:].std(0).min().item() >= 0.005
    STATS["growing_600"] = {"keys_at_end": 20 + 35 + 600 * tf, "score_std_last_300": runs[0][300:].std(0).tolist()}
    _dump()


def test_config4_driver_score_vectors_at_full_size_within_flat_1e3(stable):
    """BASELINE configs[4] at FULL model size: two short videos through LiveInferForBenchmark (reset / set_fps /
    input_query_stream / input_video_stream / inference, test/inference.py:592-711) against the oracle driver on the same
    uint8 frames - every frame's three scores within the flat 1e-3, then both score sets through the ported TVSum
    post-processing (fused score, metrics, Savitzky-Golay peak picking)."""
    import numpy as np
    from aha_amd.live_infer import LiveInferForDemo, round_numbers
    from aha_amd.postproc import evaluate_tvsum, fuse_scores
    from aha_amd.tokenization import SyntheticChatTokenizer
    from test_gpu_configs import _driver_pair
    cfg, rt, olm, _, _, w = stable
    tok = SyntheticChatTokenizer(cfg.lm.vocab_size)
    q = "Which moments of this video are the highlights?"
    qids = tok.apply_chat_template([{"role": "user", "content": q}], add_stream_prompt=True)
    drv, ob, _o32 = _driver_pair(cfg, w, rt, "default_sink", 2048, 32, tok, dtypes=(torch.bfloat16,))
    params = dict(alpha=0.0, beta=-1.0, epsilon=-5.0, uncertainty_threshold=0.04)          # outputs/grid_search_params.json "tvsum"
    pred, pred_o, gt, worst = {}, {}, {}, 0.0
    for v, n in enumerate((16, 20)):
        frames = make_frames(n, cfg.vision.image_size, seed=900 + v, tint=True)
        rows = []
        for d in (drv, ob):
            d.reset()
            d.set_fps(fps=1)
            d.input_query_stream([{"role": "user", "content": q, "time": 0}] if d is drv else [{"role": "user", "time": 0, "ids": qids}])
            d.input_video_stream(frames.cuda() if d is drv else frames)
            d.inference()
            rows.append(d.debug_data_list)
        assert len(rows[0]) == n == len(rows[1])
        for key in ("informative_score", "relevance_score", "uncertainty_score"):
            a, b = np.array([r[key] for r in rows[0]]), np.array([r[key] for r in rows[1]])
            worst = max(worst, float(np.abs(a - b).max()))
        k = f"synth_{v:03d}"
        pred[k], pred_o[k] = fuse_scores(round_numbers(rows[0], 3), **params), fuse_scores(round_numbers(rows[1], 3), **params)
        gt[k] = np.random.default_rng(10_000 + v).integers(1, 6, (20, n)).mean(0) / 5.0
    m_hip, m_or = evaluate_tvsum(gt, pred), evaluate_tvsum(gt, pred_o)
    STATS["config4_full_size"] = {"videos": 2, "frames": [16, 20], "max_abs_score_diff": worst,
                                  "metrics_hip": [float(x) for x in m_hip], "metrics_oracle": [float(x) for x in m_or]}
    print("configs[4] full size:", json.dumps(STATS["config4_full_size"]))
    _dump()
    assert worst <= FLAT_TOL, worst
    assert all(np.isfinite(x) for x in m_hip)
    ticks = LiveInferForDemo.find_ticks(None, pred["synth_001"], fps=1)
    assert all(0 <= t < 20 for t in ticks)


def test_reference_faithful_geometry_within_flat_1e3():
    """The same flat statement on the REFERENCE-FAITHFUL shapes (models/arguments_live.py:22-24): so400m/14@384 tower
    (26 layers, width 1152, 16 heads x 72, MLP 4304, 729 patches), bilinear 27 -> 7 pooling = 49 tokens per frame, Qwen2-7B dims.  Its own
    stable-regime weights and calibrated heads; 20 frames end to end on both sides, TrulyStaticCache and SinkCache (W = 512: evicting from
    frame 9), every score within 1e-3 of the bf16 oracle."""
    from aha_amd.runtime import Runtime
    from aha_amd.synth import calibrated_heads
    from oracle.cache_policies import GrowingPolicy, make_policy
    from oracle.qwen2_live import OracleLM, frame_scores
    from oracle.vision_tower import OracleVision
    cfg = preset("ref")
    n, ncal = 20, 16
    wd = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True, regime="stable")
    w = {k: v.cpu() for k, v in wd.items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    tf, H, S, V = cfg.frame_num_tokens, cfg.lm.hidden_size, cfg.vision.image_size, cfg.lm.vocab_size
    assert tf == 49 and cfg.vision.head_dim == 72
    ov, olm = OracleVision(cfg, w, torch.bfloat16), OracleLM(cfg.lm, w, torch.bfloat16)
    cal = make_frames(ncal, S, seed=778, tint=True)
    emb_cal = torch.cat([ov.visual_embed(cal[i:i + 4]) for i in range(0, ncal, 4)]).view(ncal, tf, H)
    hid = torch.stack([olm.step(emb_cal[i:i + 1], GrowingPolicy())["hidden"][0, -1] for i in range(ncal)])
    heads = calibrated_heads(hid.float())
    w.update(heads)
    olm.w.update(heads)
    wd.update({k: v.cuda() for k, v in heads.items()})
    rt = Runtime(cfg, wd, max_step_tokens=160, max_vit_frames=32)
    del wd
    torch.cuda.empty_cache()
    frames = make_frames(n, S, seed=4243, tint=True)
    emb_hip = rt.visual_embed(frames.cuda()).view(n, tf, H)
    emb_ref = torch.cat([ov.visual_embed(frames[i:i + 4]) for i in range(0, n, 4)]).view(n, tf, H)
    q_ids, pre_ids = make_token_ids(20, V, seed=101), make_token_ids(35, V, seed=100)
    out = {}
    try:
        for policy, window, sink in (("static", 2048, 0), ("default_sink", 512, 32)):
            st = rt.open_stream(policy, window, sink)
            rt.lm_step([st], rt.embed_tokens(q_ids).view(1, -1, H))
            pre = rt.embed_tokens(pre_ids).view(1, -1, H)
            got = torch.empty((n, 3), device="cuda")
            for i in range(n):
                x = emb_hip[i:i + 1] if i else torch.cat([pre, emb_hip[:1]], 1)
                rt.lm_step([st], x.contiguous(), out=got[i:i + 1])
            got = got.cpu()
            seq_hip = st.get_seq_length()
            st.close()
            pol = make_policy(policy, window, sink)
            olm.step(olm.embed_tokens(q_ids), pol)
            opre = olm.embed_tokens(pre_ids)
            want = torch.stack([frame_scores(olm.step(emb_ref[i:i + 1] if i else torch.cat([opre, emb_ref[:1]], 1), pol))[0] for i in range(n)])
            d = (got.double() - want.double()).abs()
            out[policy] = {"seq_len": seq_hip, "max_abs_diff": d.max(0).values.tolist(), "score_std": want.double().std(0).tolist()}
            assert seq_hip == pol.get_seq_length()
            assert d.max().item() <= FLAT_TOL, (policy, d.max(0).values.tolist())
            assert want.double().std(0).min().item() >= 0.015
    finally:
        STATS["ref_so400m_384"] = out
        print("flat parity [reference-faithful geometry]:", json.dumps(out))
        _dump()
        rt.close()
