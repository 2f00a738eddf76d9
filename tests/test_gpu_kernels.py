"""Per-kernel parity with FLAT bounds (tests/ulp.py): every kernel of the LM step and of the vision tower is driven through
the operator-level C ABI (include/aha_amd.h, "operator level") on seeded inputs and compared with an exact (fp64) evaluation
of the same torch op.  Bounds are in bf16 ulps of the exact output and do not widen with depth or with the noise band of a
random network:

  Linear outputs (weight-streaming GEMM, all four epilogues, M in {1, 36, 71, 288}; tiled MFMA GEMM, every variant):
      |y - exact| <= 0.5 ulp (correct rounding) + 1e-5 * sum_k |x_k w_k|  (fp32 accumulation of K <= 18944 terms)  -> "<= 1 ulp"
  SwiGLU epilogue: given the SAME bf16 gate / up outputs (bit-identical k-order), <= 1 ulp, and >= 99.9 % bit-equal
  norms: h update bit-exact; normalised rows <= 1 ulp
  attention over the KV cache (ring addressing, sink wrap, key splits, causal edge): <= 1 ulp + the P->bf16 rounding bound
      2^-8 * sum_j p_j |v_jd| (unit roundoff of an 8-bit significand) that any flash/sdpa evaluation with bf16 probabilities carries; rows that put all their weight
      on ONE key at a boundary (first / last / sink edge / ring wrap / split edge / causal edge) must return that key's V.
  heads: raw logits <= 1 ulp; scores are exact functions of the raw logits.
  cache policies: ring contents after aha_cache_update == the reference's own SinkCache / SlidingWindowCache /
      TrulyStaticCache outputs (tests/golden/cache_policies.npz, generated from the imported reference), bit for bit.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

import aha_amd  # noqa: F401
from aha_amd import lib as L
from aha_amd.config import LiveConfig, LMConfig, VisionConfig
from aha_amd.synth import make_weights

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ulp import bf16_ulp, report, rms, ulp_error  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STATS = {}


def _note(key, err, got=None, want=None):
    line = report(key, err, got, want)
    print(line)
    STATS[key] = {"max_ulp": float(err.max()), "line": line}
    try:                                                     # calibration record for DESIGN.md (gpurun_out/ is scratch)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(STATS, open(os.path.join(ROOT, "gpurun_out", "kernel_parity_stats.json"), "w"), indent=1)
    except OSError:
        pass


@pytest.fixture(scope="module")
def op_rt():
    """Qwen2-7B-wide single layer (the kernels' real tile configurations) with a toy tower: the operator tests need the
    context, its RoPE table and its head weights, not a deep model."""
    from aha_amd.runtime import Runtime
    cfg = LiveConfig(vision=VisionConfig(image_size=56, patch_size=14, hidden_size=128, num_hidden_layers=1, num_attention_heads=2,
                                         intermediate_size=256),
                     lm=LMConfig(num_hidden_layers=1, vocab_size=512), video_pooling_stride=2, name="op7b")
    w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, jitter=True)
    rt = Runtime(cfg, w, max_step_tokens=320, max_vit_frames=1, max_positions=4096)
    yield cfg, w, rt
    rt.close()


def _gen(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


# ---------------------------------------------------------------------------------------------------------------------
# weight-streaming GEMM (gemm_ws): the LM step's Linears
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M", [1, 36, 71, 288])
def test_gemm_ws_every_epilogue_within_one_ulp(op_rt, M):
    cfg, _, rt = op_rt
    g = _gen(100 + M)
    for name, (N, K) in {"qkv": (4608, 3584), "o_proj": (3584, 3584), "down": (3584, 18944)}.items():
        x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).bfloat16()
        w = (torch.randn(N, K, generator=g, device="cuda") * 0.02).bfloat16()
        bias = (torch.randn(N, generator=g, device="cuda") * 0.1).bfloat16()
        exact = x.double() @ w.double().T
        slack = 1e-5 * (x.float().abs() @ w.float().abs().T)
        lin = rt.linear(w)
        y = lin(x, L.EPI_BF16, bias=bias)
        e = ulp_error(y, exact + bias.double(), slack=slack)
        _note(f"gemm_ws bf16+bias {name} M={M}", e)
        assert e.max().item() <= 0.5 + 1e-6, (name, M, e.max().item())
        yf = lin(x, L.EPI_F32)
        assert torch.equal(yf, yf.bfloat16().float())                         # values are bf16-rounded
        e = ulp_error(yf, exact, slack=slack)
        assert e.max().item() <= 0.5 + 1e-6, (name, M, "f32", e.max().item())
        for S in (1, 7, 8):
            slabs = lin(x, L.EPI_SPLITK_F32, split_k=S)
            assert slabs.shape[0] == rt.lib.aha_linear_split_k(rt.ctx, lin.handle, S)
            d = (slabs.double().sum(0) - exact).abs()
            assert (d <= slack.double() + 1e-30).all(), (name, M, S, float((d / slack.double()).max()))
        lin.close()
    # gate/up pair with the fused SwiGLU epilogue: same bits as silu(gate) * up applied to the two bf16 Linear outputs
    N, K = cfg.lm.intermediate_size, cfg.lm.hidden_size
    x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).bfloat16()
    wg = (torch.randn(N, K, generator=g, device="cuda") * 0.02).bfloat16()
    wu = (torch.randn(N, K, generator=g, device="cuda") * 0.02).bfloat16()
    lg, lu, pair = rt.linear(wg), rt.linear(wu), rt.linear(wg, wu)
    gt, ut = lg(x), lu(x)
    e = ulp_error(gt, x.double() @ wg.double().T, slack=1e-5 * (x.float().abs() @ wg.float().abs().T))
    assert e.max().item() <= 0.5 + 1e-6
    sg = (gt.float() / (1.0 + torch.exp(-gt.float()))).bfloat16()           # F.silu in bf16: fp32 math, one rounding
    want = (sg.float() * ut.float()).bfloat16()
    got = pair(x, L.EPI_SWIGLU)
    e = ulp_error(got, want.double(), floor=2.0 ** -20)
    _note(f"gemm_ws swiglu M={M}", e, got, want)
    assert e.max().item() <= 1.0 + 1e-6 and (got == want).float().mean().item() >= 0.999
    for h in (lg, lu, pair):
        h.close()


@pytest.mark.parametrize("M", [129, 160, 200, 288, 320, 568])
def test_mid_m_gemm_kernel_is_bit_identical_to_the_register_streaming_kernel(op_rt, M):
    """Row blocks above 128 run gemm_wl.hip (both operands staged through LDS by LDS-DMA, five stages); it sums each output
    element's k-steps in the same order with the same split-K slices as gemm_ws.hip, so switching it off (tuning use_wl = 0)
    must not change one bit - for the split-K slabs and for the fused SwiGLU epilogue, ragged row counts included."""
    cfg, _, rt = op_rt
    g = _gen(300 + M)
    try:
        for N, K, S in ((4608, 3584, 7), (3584, 18944, 8), (3584, 3584, 8)):
            x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).bfloat16()
            lin = rt.linear((torch.randn(N, K, generator=g, device="cuda") * 0.02).bfloat16())
            out = {}
            for use in (1, 0):
                rt.set_tuning("use_wl", use)
                out[use] = lin(x, L.EPI_SPLITK_F32, split_k=S).clone()
            assert torch.equal(out[0], out[1]) and torch.isfinite(out[1]).all(), (M, N, K)
            lin.close()
        N, K = cfg.lm.intermediate_size, cfg.lm.hidden_size
        x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).bfloat16()
        pair = rt.linear((torch.randn(N, K, generator=g, device="cuda") * 0.02).bfloat16(),
                         (torch.randn(N, K, generator=g, device="cuda") * 0.02).bfloat16())
        out = {}
        for use in (1, 0):
            rt.set_tuning("use_wl", use)
            out[use] = pair(x, L.EPI_SWIGLU).clone()
        assert torch.equal(out[0], out[1]) and torch.isfinite(out[1].float()).all(), M
        pair.close()
    finally:
        rt.set_tuning("use_wl", 1)


# ---------------------------------------------------------------------------------------------------------------------
# tiled MFMA GEMM (gemm_tile): the vision tower's and the projector's Linears
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1, 11, 12, 14, 21])
def test_gemm_tile_within_one_ulp(op_rt, mode):
    _, _, rt = op_rt
    rt.set_tuning("tile_dma", mode)
    g = _gen(7)
    try:
        for M in (576, 1731):
            for N, K in ((3072, 1024), (1024, 4096), (1024, 1024), (3584, 1024)):
                x = torch.randn(M, K, generator=g, device="cuda").bfloat16()
                w = (torch.randn(N, K, generator=g, device="cuda") * 0.03).bfloat16()
                bias = (torch.randn(N, generator=g, device="cuda") * 0.1).bfloat16()
                res = torch.randn(M, N, generator=g, device="cuda").bfloat16()
                exact = x.double() @ w.double().T + bias.double()
                slack = 1e-5 * (x.float().abs() @ w.float().abs().T)
                y = rt.linear_tile(x, w, bias)
                e = ulp_error(y, exact, slack=slack)
                _note(f"gemm_tile[{mode}] linear M={M} N={N} K={K}", e)
                assert e.max().item() <= 0.5 + 1e-6
                # activation / residual are applied to the bf16 Linear output y (checked above), each with one rounding
                ya = rt.linear_tile(x, w, bias, act=L.ACT_GELU_TANH)
                want = torch.nn.functional.gelu(y.double(), approximate="tanh")
                e = ulp_error(ya, want, floor=2.0 ** -24, slack=4e-6 * y.double().abs() + 1e-30)
                assert e.max().item() <= 0.5 + 1e-3, ("gelu_tanh", M, N, K, e.max().item())
                ye = rt.linear_tile(x, w, bias, act=L.ACT_GELU_ERF)
                e = ulp_error(ye, torch.nn.functional.gelu(y.double()), floor=2.0 ** -24, slack=4e-6 * y.double().abs() + 1e-30)
                assert e.max().item() <= 0.5 + 1e-3, ("gelu_erf", M, N, K, e.max().item())
                yr = rt.linear_tile(x, w, bias, residual=res)
                assert torch.equal(yr, (res.float() + y.float()).bfloat16())
        # K = 128 is the smallest K the tiled path takes; the persistent 288 x 256 kernel needs four k-steps per tile for its
        # bias line to land (aha_gemm_tile_p288_ok now says so), so this ragged, biased shape must come out right on EVERY variant
        x = torch.randn(1731, 128, generator=g, device="cuda").bfloat16()
        w = (torch.randn(1000, 128, generator=g, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(1000, generator=g, device="cuda").bfloat16()
        y = rt.linear_tile(x, w, bias)
        e = ulp_error(y, x.double() @ w.double().T + bias.double(), slack=1e-5 * (x.float().abs() @ w.float().abs().T))
        assert e.max().item() <= 0.5 + 1e-6, ("K=128 ragged tile with bias", mode, e.max().item())
    finally:
        rt.set_tuning("tile_dma", 1)


# ---------------------------------------------------------------------------------------------------------------------
# norms and heads
# ---------------------------------------------------------------------------------------------------------------------
def test_rmsnorm_and_resid_rmsnorm(op_rt):
    cfg, _, rt = op_rt
    H, eps = cfg.lm.hidden_size, cfg.lm.rms_norm_eps
    g = _gen(3)
    for M in (1, 36, 288):
        x = (torch.randn(M, H, generator=g, device="cuda") * 2).bfloat16()
        w = (1 + 0.1 * torch.randn(H, generator=g, device="cuda")).bfloat16()
        xd = x.double()
        exact_n = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + eps)                 # modeling_qwen2.py:247-251
        got = rt.rmsnorm(x, w, eps)
        # two roundings in the reference (normalised row -> bf16, then * weight -> bf16): the first may flip by one ulp
        # when the fp32 rstd differs in its last bits, so the flat bound is 1 ulp
        want = (w.float() * exact_n.float().bfloat16().float()).bfloat16()
        e = ulp_error(got, want.double(), floor=2.0 ** -20)
        _note(f"rmsnorm M={M}", e, got, want)
        assert e.max().item() <= 1.0 + 1e-6 and (got == want).float().mean().item() >= 0.999
        for S in (1, 7, 8):
            part = torch.randn(S, M, H, generator=g, device="cuda") * 0.3
            h = (torch.randn(M, H, generator=g, device="cuda")).bfloat16()
            lin = torch.zeros(M, H, device="cuda")
            for s in range(S):                                                              # the kernel's summation order
                lin = lin + part[s]
            h_want = (h.float() + lin.bfloat16().float()).bfloat16()
            h_got = h.clone()
            xn = rt.resid_rmsnorm(part, h_got, w, eps)
            assert torch.equal(h_got, h_want), (M, S)
            hd = h_want.double()
            n_exact = hd * torch.rsqrt(hd.pow(2).mean(-1, keepdim=True) + eps)
            want = (w.float() * n_exact.float().bfloat16().float()).bfloat16()
            e = ulp_error(xn, want.double(), floor=2.0 ** -20)
            assert e.max().item() <= 1.0 + 1e-6 and (xn == want).float().mean().item() >= 0.999, (M, S)


def test_heads_raw_logits_within_one_ulp_and_scores_follow(op_rt):
    cfg, w, rt = op_rt
    H = cfg.lm.hidden_size
    g = _gen(5)
    hid = (torch.randn(64, H, generator=g, device="cuda") * 3).bfloat16()
    W4 = torch.cat([w["informative_head.weight"], w["relevance_head.weight"], w["uncertainty_head.weight"]], 0).double()
    exact = hid.double() @ W4.T
    sc, raw = rt.heads(hid)
    e = ulp_error(raw, exact, slack=1e-5 * (hid.float().abs() @ W4.float().abs().T))
    _note("heads raw", e)
    assert e.max().item() <= 0.5 + 1e-6 and torch.equal(raw, raw.bfloat16().float())
    want = torch.stack([raw[:, :2].softmax(-1)[:, 1], torch.sigmoid(raw[:, 2]), torch.exp(raw[:, 3])], -1)
    assert (sc - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())


# ---------------------------------------------------------------------------------------------------------------------
# attention over the KV cache
# ---------------------------------------------------------------------------------------------------------------------
def _attention_exact(q, K, V, off, scale):
    """q [T,Hq,D], K/V [Hkv,L,D] (logical order), key j visible to row t iff j <= off + t.  fp64.
    Returns the exact output [T,Hq*D] and sum_j p_j |v_jd| (the P-rounding allowance)."""
    T, Hq, D = q.shape
    Hkv, Lk, _ = K.shape
    G = Hq // Hkv
    qd = q.double().view(T, Hkv, G, D)
    s = torch.einsum("thgd,hjd->hgtj", qd, K.double()) * scale
    vis = torch.arange(Lk, device=q.device)[None, :] <= (off + torch.arange(T, device=q.device))[:, None]
    s = s.masked_fill(~vis[None, None], float("-inf"))
    p = torch.softmax(s, -1)
    o = torch.einsum("hgtj,hjd->thgd", p, V.double()).reshape(T, Hq * D)
    pav = torch.einsum("hgtj,hjd->thgd", p, V.double().abs()).reshape(T, Hq * D)
    return o, pav


def _fill(rt, st, n_tokens, g, chunk=36):
    d = rt.desc
    done = 0
    while done < n_tokens:
        T = min(chunk, n_tokens - done)
        k = torch.randn(d.kv_heads, T, d.head_dim, generator=g, device="cuda").bfloat16()
        v = torch.randn(d.kv_heads, T, d.head_dim, generator=g, device="cuda").bfloat16()
        rt._chk(rt.lib.aha_cache_update(rt.ctx, st.handle, 0, k.data_ptr(), v.data_ptr(), T, None, None, torch.cuda.current_stream().cuda_stream))
        done += T
    torch.cuda.synchronize()


def _probe_queries(K, T, G, targets, g, scales=(1.0, 3.0)):
    """Query rows [T, Hq, D]: random rows (two temperature classes) plus rows that put all their weight on one chosen key."""
    Hkv, Lk, D = K.shape
    q = torch.randn(T, Hkv, G, D, generator=g, device="cuda")
    q[:, :, ::2] *= scales[0]
    q[:, :, 1::2] *= scales[1]
    i = 0
    for hk in range(Hkv):
        for gi in range(G):
            for t in range(T):
                if (t + gi) % 3 == 0 and i < 10 * len(targets):
                    j = targets[i % len(targets)]
                    q[t, hk, gi] = 2.0 * K[hk, j].float()
                    i += 1
    return q.reshape(T, Hkv * G, D).bfloat16()


@pytest.mark.parametrize("case", ["short_growing", "sink_wrapped_2048", "sliding_wrapped_2048", "static_frozen"])
def test_attention_over_the_cache_flat_bound(op_rt, case):
    cfg, _, rt = op_rt
    d = rt.desc
    T, G, D = 36, d.heads // d.kv_heads, d.head_dim
    scale = D ** -0.5
    g = _gen({"short_growing": 11, "sink_wrapped_2048": 12, "sliding_wrapped_2048": 13, "static_frozen": 14}[case])
    if case == "short_growing":
        st, n = rt.open_stream(None, capacity=4096), 20 + T
    elif case == "sink_wrapped_2048":
        st, n = rt.open_stream("default_sink", 2048, 32), 20 + 75 * T          # 20 + 2700 tokens: the ring has wrapped
    elif case == "sliding_wrapped_2048":
        st, n = rt.open_stream("sliding_window", 2048, 0), 20 + 75 * T
    else:
        st, n = rt.open_stream("static", 2048, 0), 20
    _fill(rt, st, n, g)
    Lk = st.get_seq_length()
    K, V = st.export_kv(0), st.export_kv(0, True)                               # logical order, as the reference's update() returns
    assert K.shape == (d.kv_heads, Lk, D)
    off = (1 << 29) if case == "static_frozen" else Lk - T
    edge = [0, 1, Lk - 1, Lk - 2, Lk - T, Lk - T - 1, Lk - T + 1, 31, 32, 33, 63, 64, 255, 256, 257, 511, 512, 1023, 1024, 2047]
    # the ring wrap point (logical index whose physical slot is the ring's first) sits somewhere in the window: probe a sweep
    edge += list(range(40, Lk, max(1, Lk // 23)))
    targets = sorted({j for j in edge if 0 <= j < Lk})
    q = _probe_queries(K, T, G, targets, g)
    exact, pav = _attention_exact(q.view(T, d.heads, D), K, V, min(off, 1 << 20), scale)
    for split_len in (0, 64, 256, 2048):            # 0: the step's own geometry
        got = rt.attention([st], q.view(1, T, -1), 0, causal_off=[off], split_len=split_len)[0]
        e = ulp_error(got, exact, floor=2.0 ** -10, slack=(2.0 ** -8 + 1e-4) * pav)
        _note(f"attention {case} split_len={split_len} Lk={Lk}", e)
        assert e.max().item() <= 1.0 + 1e-6, (case, split_len, e.max().item())
    st.close()


@pytest.mark.parametrize("case", ["short_growing", "sink_wrapped_2048"])
def test_both_lm_attention_kernels_give_a_row_the_same_bits(op_rt, case):
    """A step's attention kernel is chosen by its shape (attn_lm_kernel: all row tiles of a frame-sized step in one 8-wave
    workgroup, LDS-DMA staging, once that fills the chip; attn_fwd_kernel otherwise).  Batched, solo and last-token-only
    evaluations of a row must agree exactly, so the two kernels must: forced both ways (tuning attn_lm = 0 / 2), every key-split
    shape, and a single-token step (always attn_fwd_kernel) against the same row inside a frame-sized one."""
    cfg, _, rt = op_rt
    d = rt.desc
    T, D = 36, d.head_dim
    g = _gen(31)
    st = rt.open_stream(None, capacity=4096) if case == "short_growing" else rt.open_stream("default_sink", 2048, 32)
    _fill(rt, st, 20 + T if case == "short_growing" else 20 + 75 * T, g)
    Lk = st.get_seq_length()
    q = (torch.randn(1, T, d.heads * D, generator=g, device="cuda") * 2).bfloat16()
    try:
        for split_len in (64, 256, 2048):
            out = {}
            for mode in (0, 2):
                rt.set_tuning("attn_lm", mode)
                out[mode] = rt.attention([st], q, 0, causal_off=[Lk - T], split_len=split_len).clone()
            assert torch.equal(out[0], out[2]) and torch.isfinite(out[0].float()).all(), (case, split_len)
            last = rt.attention([st], q[:, -1:].contiguous(), 0, causal_off=[Lk - 1], split_len=split_len)   # the last row alone sees every key too
            assert torch.equal(last[0, 0], out[2][0, -1]), (case, split_len)
    finally:
        rt.set_tuning("attn_lm", 1)
    st.close()


@pytest.mark.parametrize("Lk_target", [8192, 21655])
def test_attention_over_a_long_growing_cache_flat_bound(Lk_target):
    """past_key_values=None in the reference: the cache only grows (test/inference.py:154-155); SURVEY.md 8d config 2 asks for 600
    frames = 20 + 35 + 600 x 36 = 21,655 keys.  Same flat bound as the 2,048-key cases, probe rows on single keys at the first /
    last keys and on a sweep of key-split edges of both geometries (the step's own: up to 64 splits, merged in chunks of 16 by the
    combine kernels; and 1,408-key splits) with the 64-key block edges around them; both LM attention kernels give the same bits."""
    from aha_amd.runtime import Runtime
    cfg = LiveConfig(vision=VisionConfig(image_size=56, patch_size=14, hidden_size=128, num_hidden_layers=1, num_attention_heads=2,
                                         intermediate_size=256),
                     lm=LMConfig(num_hidden_layers=1, vocab_size=512), video_pooling_stride=2, name="op7b_long")
    rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16, jitter=True), max_step_tokens=320, max_vit_frames=1,
                 max_positions=22016)
    d = rt.desc
    T, G, D = 36, d.heads // d.kv_heads, d.head_dim
    g = _gen(77 + Lk_target)
    st = rt.open_stream(None, capacity=22016)
    _fill(rt, st, Lk_target, g)
    Lk = st.get_seq_length()
    assert Lk == Lk_target
    K, V = st.export_kv(0), st.export_kv(0, True)
    own = 256 if Lk <= 64 * 256 else -(-(-(-Lk // 64)) // 64) * 64      # the step's own geometry at one stream: 256-key splits, at most 64 of them
    forced = 1408                                                       # what rounds 1-3 used at 21.6k keys (16 splits)
    edge = [0, 1, Lk - 1, Lk - 2, Lk - T, Lk - T - 1, Lk - T + 1]
    for sl in (own, forced):
        for e0 in range(sl, Lk, sl):
            edge += [e0 - 65, e0 - 64, e0 - 1, e0, e0 + 1, e0 + 63, e0 + 64]
    edge = edge[:7] + edge[7::max(1, (len(edge) - 7) // 160)]           # a sweep over the edges (the probe rows are limited)
    targets = sorted({j for j in edge if 0 <= j < Lk})
    q = _probe_queries(K, T, G, targets, g)
    exact, pav = _attention_exact(q.view(T, d.heads, D), K, V, Lk - T, D ** -0.5)
    for split_len in (0, forced):
        got = rt.attention([st], q.view(1, T, -1), 0, causal_off=[Lk - T], split_len=split_len)[0]
        e = ulp_error(got, exact, floor=2.0 ** -10, slack=(2.0 ** -8 + 1e-4) * pav)
        _note(f"attention growing Lk={Lk} split_len={split_len}", e)
        assert e.max().item() <= 1.0 + 1e-6, (Lk, split_len, e.max().item())
    out = {}
    for mode in (0, 2):
        rt.set_tuning("attn_lm", mode)
        out[mode] = rt.attention([st], q.view(1, T, -1), 0, causal_off=[Lk - T]).clone()
    rt.set_tuning("attn_lm", 1)
    assert torch.equal(out[0], out[2])
    st.close()
    rt.close()


def test_fused_static_finish_and_attention_matches_the_tile_kernels(op_rt):
    """Frozen TrulyStaticCache steps with a short prefix run qkv_finish + attention as ONE launch on the vector ALUs
    (qkv_finish_attn_static_kernel).  Its attention output must sit within 2 bf16 ulps (at the tensor's scale) of the
    qkv_finish -> attn_fwd_kernel pair on the same step (same rotated queries, same keys; only the summation order of the
    128-deep dot products and of the <= 64-key PV sums differs), for 20- and 64-key prefixes, T = 36 and T = 1, two streams."""
    cfg, _, rt = op_rt
    H = cfg.lm.hidden_size
    g = _gen(41)
    try:
        for n_prefix in (20, 64):
            for T, B in ((36, 1), (1, 2), (36, 2)):
                outs = {}
                for mode in (0, 1):
                    rt.set_tuning("static_attn", mode)
                    gg = torch.Generator(device="cuda").manual_seed(1000 + n_prefix)
                    sts = [rt.open_stream("static", 2048, 0) for _ in range(B)]
                    rt.lm_step(sts, (torch.randn(B, n_prefix, H, generator=gg, device="cuda") * 0.5).bfloat16())
                    x = (torch.randn(B, T, H, generator=gg, device="cuda") * 0.5).bfloat16()
                    sc = rt.lm_step(sts, x).clone()
                    outs[mode] = (rt.debug_tap("attn_out", B, T).clone(), sc)
                    for s_ in sts:
                        s_.close()
                a0, a1 = outs[0][0].float(), outs[1][0].float()
                e = ulp_error(a1, a0.double(), floor=rms(a0))
                _note(f"fused static attention prefix={n_prefix} T={T} B={B}", e)
                assert e.max().item() <= 2.0, (n_prefix, T, B, e.max().item())
                assert (outs[0][1] - outs[1][1]).abs().max().item() <= 2e-2 and torch.isfinite(outs[1][1]).all()
    finally:
        rt.set_tuning("static_attn", 1)


def test_attention_two_streams_of_different_length(op_rt):
    cfg, _, rt = op_rt
    d = rt.desc
    T, D = 36, d.head_dim
    g = _gen(21)
    a, b = rt.open_stream("default_sink", 2048, 32), rt.open_stream(None, capacity=1024)
    _fill(rt, a, 20 + 60 * T, g)
    _fill(rt, b, 20 + 5 * T, g)
    qs, exacts, pavs = [], [], []
    for st in (a, b):
        Lk = st.get_seq_length()
        K, V = st.export_kv(0), st.export_kv(0, True)
        q = _probe_queries(K, T, d.heads // d.kv_heads, [0, Lk - 1, Lk - T, Lk // 2], g)
        ex, pav = _attention_exact(q.view(T, d.heads, D), K, V, Lk - T, D ** -0.5)
        qs.append(q.view(T, -1)); exacts.append(ex); pavs.append(pav)
    got = rt.attention([a, b], torch.stack(qs), 0)
    for i in range(2):
        e = ulp_error(got[i], exacts[i], floor=2.0 ** -10, slack=(2.0 ** -8 + 1e-4) * pavs[i])
        assert e.max().item() <= 1.0 + 1e-6, (i, e.max().item())
    a.close(); b.close()


# ---------------------------------------------------------------------------------------------------------------------
# cache policies: the device ring against the reference's own outputs (bit-exact)
# ---------------------------------------------------------------------------------------------------------------------
def test_device_ring_reproduces_the_reference_cache_classes_bit_for_bit():
    """tests/golden/cache_policies.npz holds what the reference's SinkCache / SlidingWindowCache / TrulyStaticCache
    (test/*_cache.py, imported from the reference when the fixture was generated) return from update() on a seeded K/V
    stream with ragged chunk sizes.  The same stream goes through aha_cache_update (ring addressing + in-place re-rotation
    with the device-built table); the K/V the attention would see must carry exactly those bits."""
    import dataclasses
    from aha_amd.config import preset
    from aha_amd.runtime import Runtime
    from make_golden import CACHE_D, CACHE_HKV, CACHE_LAYERS, CACHE_SINK, CACHE_STEPS, CACHE_THETA, CACHE_W, bf16_bits, cache_inputs
    base = preset("tiny")
    cfg = dataclasses.replace(base, lm=dataclasses.replace(base.lm, num_hidden_layers=CACHE_LAYERS, num_key_value_heads=CACHE_HKV,
                                                           head_dim=CACHE_D, rope_theta=CACHE_THETA), name="cachegold")
    rt = Runtime(cfg, make_weights(cfg, dtype=torch.bfloat16), max_step_tokens=64, max_vit_frames=1, max_positions=1024)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "cache_policies.npz"))
    checked = 0
    for name, alt in (("sink", "default_sink"), ("sliding", "sliding_window"), ("static", "static")):
        st = rt.open_stream(alt, CACHE_W, CACHE_SINK if name == "sink" else 0)
        for step, (T, layers) in enumerate(cache_inputs()):
            assert st.get_seq_length() == int(gold[f"{name}_len_before"][step]), (name, step)
            for l, (k, v) in enumerate(layers):
                Kr, Vr = rt.cache_update(st, l, k.cuda(), v.cuda())
                if f"{name}_k_s{step}_l{l}" in gold:
                    assert np.array_equal(bf16_bits(Kr.cpu()), gold[f"{name}_k_s{step}_l{l}"]), (name, step, l, "K")
                    assert np.array_equal(bf16_bits(Vr.cpu()), gold[f"{name}_v_s{step}_l{l}"]), (name, step, l, "V")
                    checked += 1
        assert st.get_seq_length() == int(gold[f"{name}_len_final"])
        assert st.seen_tokens == sum(CACHE_STEPS)
        st.close()
    assert checked == 3 * 4 * CACHE_LAYERS
    rt.close()


@pytest.mark.parametrize("name,alt", [("sink", "default_sink"), ("sliding", "sliding_window")])
def test_device_ring_reproduces_the_reference_caches_at_the_benchmark_geometry(name, alt):
    """tests/golden/cache_bench.npz: the reference's own SinkCache / SlidingWindowCache at W 2048, sink 32,
    head_dim 128, theta 1e6, 4 KV heads; chunks 20, 71, then 36 x 120, so every kept key goes through its whole life of ~55 bf16
    re-rotations by sink_rerotate_kernel<128>.  The K and V that aha_cache_update returns carry the reference's bits at EVERY step
    and layer (sha256), and the sampled rows (sink edge, first kept key, ring middle, new-chunk edge, last key) match one by one."""
    import dataclasses
    from aha_amd.config import preset
    from aha_amd.runtime import Runtime
    import make_golden as mg
    base = preset("tiny")
    cfg = dataclasses.replace(base, lm=dataclasses.replace(base.lm, num_hidden_layers=mg.BENCH_LAYERS, num_attention_heads=mg.BENCH_HKV,
                                                           num_key_value_heads=mg.BENCH_HKV, head_dim=mg.BENCH_D, rope_theta=mg.BENCH_THETA),
                              name="cachebench")
    rt = Runtime(cfg, make_weights(cfg, dtype=torch.bfloat16), max_step_tokens=128, max_vit_frames=1, max_positions=4096)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "cache_bench.npz"))
    st = rt.open_stream(alt, mg.BENCH_W, mg.BENCH_SINK if name == "sink" else 0)
    first_bad = None
    for step, (T, layers) in enumerate(mg.bench_cache_inputs()):
        assert st.get_seq_length() == int(gold[f"{name}_len_before"][step]), (name, step)
        for l, (k, v) in enumerate(layers):
            Kr, Vr = rt.cache_update(st, l, k.cuda(), v.cuda())
            Kc, Vc = Kr.cpu(), Vr.cpu()
            if step in mg.BENCH_SAMPLE_STEPS:
                rows = [r for r in mg.BENCH_SAMPLE_ROWS if r < Kc.shape[2]]
                assert np.array_equal(mg.bf16_bits(Kc[0, 1, rows]), gold[f"{name}_k_s{step}_l{l}"]), (name, step, l, "K rows")
                assert np.array_equal(mg.bf16_bits(Vc[0, 1, rows]), gold[f"{name}_v_s{step}_l{l}"]), (name, step, l, "V rows")
            ok = np.array_equal(mg.kv_digest(Kc), gold[f"{name}_digest"][step, l, 0]) and np.array_equal(mg.kv_digest(Vc), gold[f"{name}_digest"][step, l, 1])
            if not ok and first_bad is None:
                first_bad = (step, l)
    assert first_bad is None, f"{name}: returned K/V differ from the reference's from step {first_bad[0]}, layer {first_bad[1]}"
    assert st.get_seq_length() == int(gold[f"{name}_len_final"]) == mg.BENCH_W
    st.close()
    rt.close()


@pytest.mark.gpu
def test_python_cache_classes_update_reproduces_the_reference_fixture():
    """The drop-in's SinkCache / SlidingWindowCache / TrulyStaticCache objects take the reference's operator call
    `update(key_states, value_states, layer_idx, cache_kwargs)` (test/sink_cache.py:74-80) and return, bit for bit, what the
    imported reference classes returned when tests/golden/cache_policies.npz was generated."""
    import dataclasses
    from aha_amd.cache import SinkCache, SlidingWindowCache, TrulyStaticCache
    from aha_amd.config import preset
    from aha_amd.runtime import Runtime
    from make_golden import CACHE_D, CACHE_HKV, CACHE_LAYERS, CACHE_SINK, CACHE_STEPS, CACHE_THETA, CACHE_W, bf16_bits, cache_inputs, rope_table
    base = preset("tiny")
    cfg = dataclasses.replace(base, lm=dataclasses.replace(base.lm, num_hidden_layers=CACHE_LAYERS, num_key_value_heads=CACHE_HKV,
                                                           head_dim=CACHE_D, rope_theta=CACHE_THETA), name="cachegold")
    rt = Runtime(cfg, make_weights(cfg, dtype=torch.bfloat16), max_step_tokens=64, max_vit_frames=1, max_positions=1024)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "cache_policies.npz"))
    checked = 0
    for name, cache in (("sink", SinkCache(CACHE_W, CACHE_SINK)), ("sliding", SlidingWindowCache(CACHE_W)), ("static", TrulyStaticCache(CACHE_W))):
        with pytest.raises(RuntimeError):
            cache.update(torch.zeros(1, CACHE_HKV, 1, CACHE_D), torch.zeros(1, CACHE_HKV, 1, CACHE_D), 0)      # unbound: loud
        cache.bind(rt)
        assert cache.get_max_length() == CACHE_W and cache.get_max_cache_shape() == CACHE_W
        for step, (T, layers) in enumerate(cache_inputs()):
            assert cache.get_seq_length() == int(gold[f"{name}_len_before"][step]), (name, step)
            L = cache.get_seq_length()
            cos, sin = rope_table((L + torch.arange(T))[None], CACHE_D, CACHE_THETA, torch.bfloat16)
            for l, (k, v) in enumerate(layers):
                kw = {"sin": sin.cuda(), "cos": cos.cuda(), "cache_position": None}    # 4.49-style kwargs, as Qwen2Attention passes them
                if name == "sink" and step == 0 and l == 0:
                    with pytest.raises(ValueError):                                    # the reference would shift WITHOUT re-rotating: refused, loudly
                        cache.update(k.cuda(), v.cuda(), l, {"sin": None, "cos": None})
                    with pytest.raises(NotImplementedError):
                        cache.update(k.cuda(), v.cuda(), l, {"sin": sin.cuda(), "cos": cos.cuda(), "partial_rotation_size": 32})
                    with pytest.raises(ValueError):                                    # positions that are not get_seq_length() + arange(T)
                        bad = rope_table((L + 5 + torch.arange(T))[None], CACHE_D, CACHE_THETA, torch.bfloat16)
                        cache.update(k.cuda(), v.cuda(), l, {"cos": bad[0].cuda(), "sin": bad[1].cuda()})
                    assert cache.get_seq_length() == L                                 # a refused call changes nothing
                Kr, Vr = cache.update(k.cuda()[None] if k.dim() == 3 else k.cuda(), v.cuda()[None] if v.dim() == 3 else v.cuda(), l, kw)
                assert Kr.dim() == 4 and Kr.shape[0] == 1
                if f"{name}_k_s{step}_l{l}" in gold:
                    assert np.array_equal(bf16_bits(Kr.cpu()), gold[f"{name}_k_s{step}_l{l}"]), (name, step, l, "K")
                    assert np.array_equal(bf16_bits(Vr.cpu()), gold[f"{name}_v_s{step}_l{l}"]), (name, step, l, "V")
                    checked += 1
        assert cache.get_seq_length() == int(gold[f"{name}_len_final"]) and cache._seen_tokens == sum(CACHE_STEPS)
        cache.reset()
        assert cache.get_seq_length() == 0
    assert checked == 3 * 4 * CACHE_LAYERS
    rt.close()
