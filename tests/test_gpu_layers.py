"""Teacher-forced per-layer parity at BASELINE width and depth, flat bounds.

The end-to-end score checks of tests/test_gpu_parity.py compare against the oracle's own bf16 noise band, which at 28 random
layers is 30-100x wider than the 1e-3 the north star states.  A band cannot tell a correct kernel from
a subtly wrong one, so this file removes the depth: every decoder layer is run ALONE on the HIP path (tuning layer_first /
layer_count) from the bf16 oracle's input to that layer, and every intermediate tensor the layer produces is compared with
the oracle's at the same point.  With the input shared, the two evaluations differ only by fp32 summation order and the
bf16 roundings it flips, so the distance is a few bf16 ulps AT THE TENSOR'S SCALE, for every one of the 28 layers, with and
without cached keys - a wrong mask, slot, RoPE position, k-slice or rounding point moves it by orders of magnitude.

Bounds (bf16 ulps of max(|oracle|, rms(oracle)); calibrated once on MI355X):
    rotated q / k, v, SwiGLU activation, attention output, layer output: BOUND below.
Then the tail: final norm + the three heads on the oracle's last hidden rows (raw head logits <= 1 ulp -> what fraction of
scores lands within 1e-3 is reported), and head logits after a teacher-forced LAST layer (<= 2 ulp).
"""
import json
import os
import sys

import pytest
import torch

import aha_amd  # noqa: F401
from aha_amd.config import preset
from aha_amd.synth import make_weights

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ulp import report, rms, ulp_error  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# flat bounds, bf16 ulps at the tensor's scale (see module docstring)
# measured on MI355X (round 2, all 28 layers x 2 steps): q 2.0, k 1.5, v 1.0, attn_out 3.0, act 8.9, h_out 4.0.  q/k/v are one Linear
# (+RoPE) away from the shared input; the MLP activation sits behind the layer's OWN attention output (only the layer input is
# forced), where two one-ulp factors multiply: a few ulps of a PRODUCT are inherent.  A wrong key / slot / position / k-slice
# moves these by hundreds of ulps.
BOUND = {"q": 4.0, "k": 4.0, "v": 2.0, "attn_out": 6.0, "act": 12.0, "h_out": 8.0}


def _flat(name, got, want):
    """max ulp distance with the reference magnitude floored at the tensor's rms (elements near zero are judged at the
    tensor's scale: their absolute error comes from the other summands, not from their own size)."""
    w = want.float().cpu()
    return ulp_error(got.float().cpu(), w, floor=rms(w))


@pytest.fixture(scope="module")
def full():
    from aha_amd.runtime import Runtime
    from oracle.qwen2_live import OracleLM
    cfg = preset("bench")
    wd = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
    rt = Runtime(cfg, wd, max_step_tokens=160, max_vit_frames=1)
    w = {k: v.cpu() for k, v in wd.items() if not k.startswith(("vision.", "mm_projector"))}
    del wd
    torch.cuda.empty_cache()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    yield cfg, rt, OracleLM(cfg.lm, w, torch.bfloat16), w
    rt.set_tuning("layer_count", 0)
    rt.close()


def test_every_layer_teacher_forced_within_flat_ulp_bounds(full):
    from oracle.cache_policies import GrowingPolicy
    cfg, rt, ob, _ = full
    lm = cfg.lm
    H, Lyr, Hq, Hkv, D = lm.hidden_size, lm.num_hidden_layers, lm.num_attention_heads, lm.num_key_value_heads, lm.head_dim
    g = torch.Generator().manual_seed(123)
    xs = [(torch.randn(1, 56, H, generator=g) * 0.05).bfloat16(),          # system prompt + frame 0 (T = 20 + 36)
          (torch.randn(1, 36, H, generator=g) * 0.05).bfloat16()]          # a frame against the cached keys
    pol = GrowingPolicy()
    traces = []
    for x in xs:
        tr = []
        ob.step(x, pol, trace=tr)
        traces.append(tr)
    worst = {k: 0.0 for k in BOUND}
    lines = []
    for l in range(Lyr):
        st = rt.open_stream(None, capacity=128)
        rt.set_tuning("layer_first", l)
        rt.set_tuning("layer_count", 1)
        for si, tr in enumerate(traces):
            t = tr[l]
            T = t["x_in"].shape[1]
            rt.lm_step([st], t["x_in"].cuda())
            got = {"q": rt.debug_tap("q_rot", 1, T), "attn_out": rt.debug_tap("attn_out", 1, T), "act": rt.debug_tap("act", 1, T),
                   "h_out": rt.debug_tap("h", 1, T),
                   "k": st.export_kv(l)[:, -T:], "v": st.export_kv(l, True)[:, -T:]}
            want = {"q": t["q"][0].transpose(0, 1).reshape(T, Hq * D), "attn_out": t["attn_out"][0], "act": t["act"][0],
                    "h_out": t["h_out"][0], "k": t["k"][0], "v": t["v"][0]}
            for name in BOUND:
                e = _flat(name, got[name], want[name])
                m = e.max().item()
                worst[name] = max(worst[name], m)
                if m > BOUND[name]:
                    lines.append(f"layer {l} step {si} {name}: {m:.2f} ulp > {BOUND[name]}")
        st.close()
    rt.set_tuning("layer_count", 0)
    print("teacher-forced per-layer worst ulp at scale:", {k: round(v, 3) for k, v in worst.items()})
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(worst, open(os.path.join(ROOT, "gpurun_out", "layer_parity_stats.json"), "w"), indent=1)
    except OSError:
        pass
    assert not lines, "\n".join(lines[:20])


def test_tail_final_norm_and_heads_flat_bounds_and_score_fraction(full):
    """(1) heads on the oracle's own final hidden rows: raw logits within 1 ulp; (2) a teacher-forced LAST layer + final
    norm + heads on the HIP path: raw logits within 2 ulp of the oracle's; (3) the whole 28-layer step, free-running: the
    fraction of scores within the north star's 1e-3 is REPORTED (it is limited by the reference's own bf16 noise, which
    moves its scores by more than that: DESIGN.md section 2), the flat assertions are (1) and (2)."""
    from oracle.cache_policies import GrowingPolicy
    from oracle.qwen2_live import frame_scores
    cfg, rt, ob, _ = full
    H, Lyr = cfg.lm.hidden_size, cfg.lm.num_hidden_layers
    g = torch.Generator().manual_seed(321)
    n = 24
    steps = [(torch.randn(1, 36, H, generator=g) * 0.05).bfloat16() for _ in range(n)]
    pol = GrowingPolicy()
    st = rt.open_stream(None, capacity=1024)
    raws, hiddens, lasts, free_scores, want_scores = [], [], [], [], []
    for x in steps:
        tr = []
        out = ob.step(x, pol, trace=tr)
        raws.append(torch.cat([out["informative_logits"][0, -1], torch.logit(out["relevance_logits"][0, -1].double()).float(),
                               out["uncertainty"][0, -1]]))
        hiddens.append(out["hidden"][0, -1])
        lasts.append(tr[-1]["x_in"])
        want_scores.append(frame_scores(out)[0])
        free_scores.append(rt.lm_step([st], x.cuda()).cpu()[0])
    st.close()
    # (1) heads alone
    hid = torch.stack(hiddens).cuda()
    sc, raw = rt.heads(hid)
    w4 = torch.cat([ob.w["informative_head.weight"], ob.w["relevance_head.weight"], ob.w["uncertainty_head.weight"]], 0)
    want_raw = torch.nn.functional.linear(hid.cpu(), w4).float()           # the oracle's bf16 Linear outputs
    e = ulp_error(raw.cpu(), want_raw, floor=2.0 ** -6)
    print(report("heads raw vs oracle bf16", e, raw.cpu(), want_raw))
    assert e.max().item() <= 1.0 + 1e-6
    # (2) teacher-forced last layer -> model.norm -> heads
    rt.set_tuning("layer_first", Lyr - 1)
    rt.set_tuning("layer_count", 1)
    worst2, within = 0.0, []
    for i in range(8):
        s2 = rt.open_stream(None, capacity=64)
        got_s, got_raw = rt.lm_step([s2], lasts[i].cuda(), want_raw=True)
        s2.close()
        # only the last layer ran against an empty cache: redo the oracle's last layer the same way for the comparison
        from oracle.qwen2_live import OracleLM  # noqa: F401
        tail = _oracle_last_layer(ob, lasts[i], Lyr - 1)
        e2 = ulp_error(got_raw.cpu()[0], tail["raw"], floor=1.0)       # ulps at unit scale: |logit| ~ 1 is where one ulp moves a score by >= 1e-3
        worst2 = max(worst2, e2.max().item())
        within.append(((got_s.cpu()[0] - tail["scores"]).abs()[:2] <= 1e-3).float())
    rt.set_tuning("layer_count", 0)
    print(f"teacher-forced last layer + norm + heads: worst raw-logit distance {worst2:.2f} ulp at unit scale; "
          f"scores within 1e-3: {torch.stack(within).mean().item() * 100:.0f} %")
    assert worst2 <= 2.0 + 1e-6
    assert torch.stack(within).min().item() == 1.0, "teacher-forced tail: every informative / relevance score within the flat 1e-3"
    # (3) free-running report
    fs, ws = torch.stack(free_scores), torch.stack(want_scores)
    frac = ((fs - ws).abs()[:, :2] <= 1e-3).float().mean().item()
    print(f"free-running 28 layers, {n} frames: informative/relevance scores within 1e-3 of the bf16 oracle: {frac * 100:.0f} % "
          f"(median |d| {(fs - ws).abs()[:, :2].median().item():.2e})")
    # (4) the same frames through the oracle in FLOAT32 (the same bf16 weight values, exact arithmetic to first order): how far is
    # each bf16 path from it?  The reference's own bf16 run is one draw of that rounding noise, the HIP path another; the north
    # star's 1e-3 can only mean "no further from the truth than the reference is", which is what is asserted.
    from oracle.qwen2_live import OracleLM
    o32 = OracleLM(cfg.lm, ob.w, torch.float32)
    pol32 = GrowingPolicy()
    s32 = torch.stack([frame_scores(o32.step(x.float(), pol32))[0] for x in steps])
    d_hip = (fs.double() - s32.double()).abs()[:, :2]
    d_ref = (ws.double() - s32.double()).abs()[:, :2]
    stats = {"teacher_forced_last_layer_raw_ulp": worst2, "free_running_frac_within_1e-3": frac,
             "free_running_vs_fp32_oracle": {
                 "frames": n,
                 "hip_median": d_hip.median().item(), "hip_p90": d_hip.flatten().quantile(0.9).item(), "hip_max": d_hip.max().item(),
                 "bf16_oracle_median": d_ref.median().item(), "bf16_oracle_p90": d_ref.flatten().quantile(0.9).item(),
                 "bf16_oracle_max": d_ref.max().item(),
                 "hip_vs_bf16_oracle_median": (fs - ws).abs()[:, :2].median().item()}}
    print("free-running vs the fp32 oracle:", json.dumps(stats["free_running_vs_fp32_oracle"]))
    del o32
    try:
        json.dump(stats, open(os.path.join(ROOT, "gpurun_out", "tail_parity_stats.json"), "w"), indent=1)
    except OSError:
        pass
    # the HIP path's distance from the fp32 truth stays within the band of the reference's own bf16 run (factor 2 on the mean, as in
    # tests/test_gpu_configs.py; both are sums of many independent roundings)
    assert d_hip.mean().item() <= max(1e-3, 2.0 * d_ref.mean().item()), (d_hip.mean().item(), d_ref.mean().item())


def test_layer_engine_gives_the_bits_of_the_launches_it_replaces(full):
    """The two round-6 single-launch forms of the layer's MLP half (tuning "engine": 1 = lm_engine.hip, LDS-DMA loader ring incl. the row
    phase; 2 = lm_stream.hip, register-streaming gate/up -> down_proj; both opt-in: neither measured ahead) against the launches: same slab
    sums, same k-step order, same roundings -> every tap, the heads and the scores bit for bit, 28 layers deep, for a frame (36 rows), a
    single token and a full 48-row step, also when replayed from a captured graph."""
    cfg, rt, _ob, _w = full
    H = cfg.lm.hidden_size
    g = torch.Generator(device="cuda").manual_seed(11)
    st = rt.open_stream("static", 2048, 32)
    rt.lm_step([st], (torch.randn(1, 20, H, generator=g, device="cuda") * 0.05).bfloat16())
    try:
        for T in (36, 1, 48):
            x = (torch.randn(1, T, H, generator=g, device="cuda") * 0.05).bfloat16()
            outs = {}
            for lv in (0, 1, 2):
                rt.set_tuning("engine", lv)
                sc, raw, hid = rt.lm_step([st], x, want_raw=True, want_hidden=True)
                outs[lv] = {"scores": sc.clone(), "raw": raw.clone(), "hid": hid.clone()}
                for name in ("h", "xn", "act"):
                    outs[lv][name] = rt.debug_tap(name, 1, T).clone()
                replay = [rt.lm_step([st], x).clone() for _ in range(3)]           # the third call runs the captured graph
                outs[lv]["replay"] = replay[-1]
                assert torch.equal(replay[0], replay[-1])
            for lv in (1, 2):
                for k in outs[0]:
                    assert not torch.isnan(outs[lv][k].float()).any(), (T, lv, k)
                    assert torch.equal(outs[0][k], outs[lv][k]), f"T={T} {k}: engine {lv} differs from the launches"
        # ... and on an evicting SinkCache (the engines replace launches that do not touch the cache, but the step around them differs: tile
        # attention + combine instead of the fused static kernel, re-rotation in front): six frames through W = 128, every score equal
        xs = [(torch.randn(1, 36, H, generator=g, device="cuda") * 0.05).bfloat16() for _ in range(6)]
        want = None
        for lv in (0, 1, 2):
            rt.set_tuning("engine", lv)
            sk = rt.open_stream("default_sink", 128, 8)
            got = torch.stack([rt.lm_step([sk], x).clone() for x in xs])
            assert sk.get_seq_length() == 128 and torch.isfinite(got).all()
            sk.close()
            if want is None:
                want = got
            else:
                assert torch.equal(got, want), f"engine {lv} differs from the launches on the SinkCache stream"
    finally:
        rt.set_tuning("engine", 0)


def _oracle_last_layer(ob, x_in, l):
    """The oracle's decoder layer l alone on an empty cache, then model.norm and the heads (same torch calls as OracleLM.step)."""
    import torch.nn.functional as F
    from oracle.cache_policies import GrowingPolicy
    from oracle.qwen2_live import rms_norm, rope_cos_sin
    c = ob.c
    B, T, _ = x_in.shape
    pos = torch.arange(T)[None]
    cos, sin = rope_cos_sin(pos, c.head_dim, c.rope_theta, ob.dtype)
    p = f"model.layers.{l}."

    class OneLayer(GrowingPolicy):                         # the layer's K/V land at index 0 of a fresh cache
        def update(self, k, v, layer_idx, cache_kwargs=None):
            return super().update(k, v, 0, cache_kwargs)
    h = x_in.to(ob.dtype)
    x = rms_norm(h, ob.w[p + "input_layernorm.weight"], c.rms_norm_eps)
    h = h + ob._attention(l, x, cos, sin, OneLayer(), 0, False)
    x = rms_norm(h, ob.w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
    h = h + ob._mlp(l, x)
    h = rms_norm(h, ob.w["model.norm.weight"], c.rms_norm_eps)
    last = h[0, -1]
    raw = torch.cat([F.linear(last, ob.w["informative_head.weight"]).float(), F.linear(last, ob.w["relevance_head.weight"]).float(),
                     F.linear(last, ob.w["uncertainty_head.weight"]).float()])
    scores = torch.stack([raw[:2].softmax(-1)[1], torch.sigmoid(raw[2]), torch.exp(raw[3])])
    return {"raw": raw, "scores": scores}
