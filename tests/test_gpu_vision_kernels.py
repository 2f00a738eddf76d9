"""Per-kernel parity of the VISION half with flat bounds (tests/ulp.py), through the operator-level C ABI
(include/aha_amd.h "vision operators"): every non-GEMM kernel of the tower against an exact (fp64) evaluation of the torch op
the reference runs, and every encoder layer teacher-forced from the bf16 oracle's input to it at ViT-L and so400m width.
(The tower's GEMMs are bounded in tests/test_gpu_kernels.py::test_tiled_gemm_*.)

  attention (SiglipAttention core, models via video_head_live_llava_qwen.py:113-115): <= 1 ulp + the P->bf16 rounding allowance
      2^-8 * sum_j p_j |v_jd|, at (576 keys, d 64: the head-resident kernel AND the restaging kernel, bit-identical to each
      other), (729 keys, d 72: the zero-padded 96-wide template, checked bit for bit against the 128-wide one), (500 keys: ragged tail block); probe rows that put all
      their weight on ONE key at the first / last / 64-key block edges must return that key's V.
  LayerNorm: <= 0.5 ulp + fp32 noise of an fp32-statistics evaluation (asserted <= 1 ulp, >= 99 % bit-equal to the rounded exact value)
  preprocess + patch unfold: EXACT (integer pixels through fixed fp32 constants)
  pooling: bilinear 24->6 / 27->7, average, max, adaptive average vs fp64: <= 1 ulp; the gathered-rows route == the direct route, bit for bit
  encoder layer, teacher-forced: output within 6 ulps at the tensor's scale of the bf16 oracle layer (measured 3-4; p99.9 2)
"""
import json
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import aha_amd  # noqa: F401
from aha_amd.config import LiveConfig, LMConfig, VisionConfig
from aha_amd.synth import make_frames, make_weights

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ulp import bf16_ulp, report, rms, ulp_error  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STATS = {}


def _note(key, err, got=None, want=None):
    line = report(key, err, got, want)
    print(line)
    STATS[key] = {"max_ulp": float(err.max()), "line": line}
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(STATS, open(os.path.join(ROOT, "gpurun_out", "vision_kernel_parity_stats.json"), "w"), indent=1)
    except OSError:
        pass


def _gen(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


def _tiny_lm():
    return LMConfig(hidden_size=256, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=2, head_dim=64, intermediate_size=512,
                    vocab_size=512)


@pytest.fixture(scope="module")
def vit_l():
    """ViT-L/14@336 geometry (the bench tower: 576 patches, width 1024, 16 heads x 64, MLP 4096), 3 layers, with a toy LM."""
    from aha_amd.runtime import Runtime
    cfg = LiveConfig(vision=VisionConfig(image_size=336, patch_size=14, hidden_size=1024, num_hidden_layers=3, num_attention_heads=16,
                                         intermediate_size=4096), lm=_tiny_lm(), video_pooling_stride=4, name="vitl_ops")
    w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, jitter=True)
    rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=8, max_positions=256)
    yield cfg, w, rt
    rt.close()


@pytest.fixture(scope="module")
def so400m():
    """so400m/14@384 geometry (the reference-faithful tower: 729 patches, width 1152, 16 heads x 72, MLP 4304), 2 layers."""
    from aha_amd.runtime import Runtime
    cfg = LiveConfig(vision=VisionConfig(image_size=384, patch_size=14, hidden_size=1152, num_hidden_layers=2, num_attention_heads=16,
                                         intermediate_size=4304), lm=_tiny_lm(), video_pooling_stride=4, name="so400m_ops")
    w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, jitter=True)
    rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=2, max_positions=256)
    yield cfg, w, rt
    rt.close()


# ---------------------------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------------------------
def _dense_exact(q, k, v, scale):
    """q,k,v bf16 [n,T,H,D] -> exact softmax(q k^T scale) v in fp64 [n,T,H,D] and sum_j p_j |v_jd| (the P->bf16 rounding bound)."""
    qd, kd, vd = q.double().transpose(1, 2), k.double().transpose(1, 2), v.double().transpose(1, 2)      # [n,H,T,D]
    p = torch.softmax(qd @ kd.transpose(-1, -2) * scale, dim=-1)
    return (p @ vd).transpose(1, 2), (p @ vd.abs()).transpose(1, 2)


def _qkv_with_probes(n, T, H, D, targets, g):
    """Random q/k/v; in frame 0 / head h the first len(targets) query rows are 24 x the key at `targets[i]` (all the softmax
    weight goes to that key: the row must return that key's V)."""
    q = (torch.randn(n, T, H, D, generator=g, device="cuda") * 1.0).bfloat16()
    k = (torch.randn(n, T, H, D, generator=g, device="cuda") * 1.0).bfloat16()
    v = (torch.randn(n, T, H, D, generator=g, device="cuda") * 1.0).bfloat16()
    for i, j in enumerate(targets):
        for h in range(H):
            q[0, i, h] = (k[0, j, h].float() * 24.0).bfloat16()
    return q, k, v


@pytest.mark.parametrize("case", ["vitl_576x64", "vitl_576x64_3frames", "vitl_576x64_1frame", "ragged_500x64", "so400m_729x72"])
def test_dense_attention_flat_bound(vit_l, case):
    """8 frames: the 12-wave head-resident kernel; 1 and 3 frames: its 4-wave row-group form (the latency path's choice) -
    each against the restaging kernel (attn_head = 0) bit for bit and against the exact result within the flat bound."""
    cfg, _, rt = vit_l
    T, H, D, n = {"vitl_576x64": (576, 16, 64, 8), "vitl_576x64_3frames": (576, 16, 64, 3), "vitl_576x64_1frame": (576, 16, 64, 1),
                  "ragged_500x64": (500, 16, 64, 8), "so400m_729x72": (729, 16, 72, 2)}[case]
    g = _gen({"vitl_576x64": 1, "vitl_576x64_3frames": 6, "vitl_576x64_1frame": 7, "ragged_500x64": 2, "so400m_729x72": 3}[case])
    targets = sorted({j for j in (0, 1, 15, 16, 63, 64, 65, 127, 128, 255, 256, 319, 320, 447, 448, 511, 512, T - 65, T - 64, T - 2, T - 1) if 0 <= j < T})
    q, k, v = _qkv_with_probes(n, T, H, D, targets, g)
    qkv = torch.cat([q.reshape(n, T, H * D), k.reshape(n, T, H * D), v.reshape(n, T, H * D)], dim=-1)
    exact, pav = _dense_exact(q, k, v, D ** -0.5)
    outs = {}
    try:
        for mode in ((0, 1, 2) if D == 64 else (1,)):          # 1 (auto): the 4-wave row-group form below 8 frames, the 12-wave form from 8 up; 2: always the 12-wave form
            rt.set_tuning("attn_head", mode)
            got = rt.vit_attention(qkv, H, D).view(n, T, H, D)
            e = ulp_error(got, exact, floor=2.0 ** -10, slack=(2.0 ** -8 + 1e-4) * pav)
            _note(f"dense attention {case} attn_head={mode}", e)
            assert torch.isfinite(got.float()).all()
            assert e.max().item() <= 1.0 + 1e-6, (case, mode, e.max().item())
            # probe rows: all weight on one key -> that key's V, within the same bound (their P rounding allowance is ~2^-8 |v|)
            for i, j in enumerate(targets):
                assert (got[0, i].float() - v[0, j].float()).abs().max().item() <= 2.0 ** -6, (case, mode, j)
            outs[mode] = got.clone()
    finally:
        rt.set_tuning("attn_head", 1)
    if D == 64:
        assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[1]), "head-resident (both forms) and restaging dense attention kernels must agree bit for bit"
    else:
        # 72 channels run the 96-wide template; tuning attn_d96 = 0 pads them to the 128-wide one (round 2): same bits
        try:
            rt.set_tuning("attn_d96", 0)
            wide = rt.vit_attention(qkv, H, D).view(n, T, H, D)
        finally:
            rt.set_tuning("attn_d96", 1)
        assert torch.equal(outs[1], wide), "the 96- and 128-wide dense attention templates must agree bit for bit"


def test_vision_path_operand_layouts_and_riders_do_not_change_a_bit(vit_l):
    """the round-5 defaults of the tower - k-blocked activations (vit_akb), k-blocked weight twins (tile_wkb), the LayerNorm
    launches' weight-prefetch riders (vit_prefetch) and the head-resident attention in both forms (attn_head) - against the row-major /
    restaging paths they replaced: whole encodes of 1, 3 and 8 frames, bit for bit."""
    cfg, _, rt = vit_l
    fr = make_frames(8, cfg.vision.image_size, seed=9).cuda()
    base = {n: rt.visual_embed(fr[:n]).clone() for n in (1, 3, 8)}
    for n in base:
        assert torch.isfinite(base[n].float()).all()
    for key, off, on in (("vit_akb", 0, 1), ("tile_wkb", 0, 1), ("vit_prefetch", 0, 2400), ("attn_head", 0, 1), ("tile_p288", 0, 1)):
        try:
            rt.set_tuning(key, off)
            for n in (1, 3, 8):
                assert torch.equal(rt.visual_embed(fr[:n]), base[n]), (key, n)
        finally:
            rt.set_tuning(key, on)
    for n in (1, 3, 8):
        assert torch.equal(rt.visual_embed(fr[:n]), base[n]), n


def test_dense_attention_kernel_choice_does_not_depend_on_the_batch(vit_l):
    """1 frame (restaging kernel: too few workgroups for the head-resident one) and 8 frames (head-resident) give frame 0 the same bits."""
    cfg, _, rt = vit_l
    g = _gen(5)
    qkv = (torch.randn(8, 576, 3 * 1024, generator=g, device="cuda")).bfloat16()
    many = rt.vit_attention(qkv, 16, 64)
    one = rt.vit_attention(qkv[:1].contiguous(), 16, 64)
    assert torch.equal(many[0], one[0])


# ---------------------------------------------------------------------------------------------------------------------
# LayerNorm, preprocess + unfold, pooling
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cols", [1024, 1152])
def test_layernorm_within_one_ulp_of_exact(vit_l, cols):
    cfg, _, rt = vit_l
    g = _gen(10 + cols)
    x = (torch.randn(777, cols, generator=g, device="cuda") * 1.7 + 0.3).bfloat16()
    w = (1.0 + 0.2 * torch.randn(cols, generator=g, device="cuda")).bfloat16()
    b = (0.1 * torch.randn(cols, generator=g, device="cuda")).bfloat16()
    exact = F.layer_norm(x.double(), (cols,), w.double(), b.double(), 1e-6)
    got = rt.layernorm(x, w, b, 1e-6)
    e = ulp_error(got, exact, floor=2.0 ** -10)
    _note(f"layernorm cols={cols}", e, got, exact.bfloat16())
    assert e.max().item() <= 1.0 + 1e-6
    assert (got == exact.bfloat16()).float().mean().item() >= 0.99


def test_preprocess_and_patch_unfold_is_exact(vit_l):
    from oracle.vision_tower import preprocess
    cfg, _, rt = vit_l
    fr = make_frames(3, 336, seed=3)
    got = rt.vit_patchify(fr.cuda()).cpu()
    P, gpatch = 14, 24
    want = preprocess(fr, torch.bfloat16)                                    # [n,3,S,S] bf16: x/255 then (x - .5)/.5 in fp32
    want = want.view(3, 3, gpatch, P, gpatch, P).permute(0, 2, 4, 1, 3, 5).reshape(3 * gpatch * gpatch, 3 * P * P)
    assert got.shape == (3 * 576, 640)
    assert torch.equal(got[:, :588], want)
    assert (got[:, 588:] == 0).all()


@pytest.mark.parametrize("grid,stride,mode", [(27, 4, "bilinear"), (24, 4, "bilinear"), (27, 4, "average"), (24, 4, "average"), (27, 4, "max"), (24, 4, "max")])
def test_pooling_against_the_reference_function_itself(vit_l, grid, stride, mode):
    """tests/golden/ref_pooling.npz = outputs of the reference's own post_projector_pooling (video_head_live_llava_qwen.py:117-136,
    compiled unmodified from its source text by tests/make_golden.py) on seeded features, in bf16 and fp32.  The HIP pool kernel on the
    same bf16 features: within one bf16 ulp of the reference's fp32 result (the reference's bf16 result itself is), max pooling exact."""
    import make_golden as mg
    _, _, rt = vit_l
    gold = np.load(os.path.join(ROOT, "tests", "golden", "ref_pooling.npz"))
    want32 = torch.from_numpy(gold[f"{mode}_{grid}_f32"])
    want16 = torch.from_numpy(gold[f"{mode}_{grid}_bf16"].view(np.int16)).view(torch.bfloat16)
    out_grid = int(round(want32.shape[1] ** 0.5))
    x = mg.pool_input(grid).bfloat16().cuda()
    got = rt.pool(x, grid, out_grid, stride, {"bilinear": 0, "average": 1, "max": 2}[mode]).cpu()
    assert got.shape == want16.shape
    # the fp32 fixture was computed from fp32 features; judge both bf16 results against an fp64 evaluation of the bf16 features
    xs = mg.pool_input(grid).bfloat16().double().view(2, grid, grid, -1).permute(0, 3, 1, 2)
    if mode == "bilinear":
        ex = F.interpolate(xs, size=[out_grid, out_grid], mode="bilinear")
    elif mode == "average":
        ex = F.avg_pool2d(xs, stride)
    else:
        ex = F.max_pool2d(xs, stride)
    ex = ex.permute(0, 2, 3, 1).reshape(2, out_grid * out_grid, -1)
    e_ref = ulp_error(want16, ex, floor=2.0 ** -10)
    e_got = ulp_error(got, ex, floor=2.0 ** -10)
    assert e_ref.max().item() <= 1.0 + 1e-6 and e_got.max().item() <= 1.0 + 1e-6, (e_ref.max().item(), e_got.max().item())
    d = ulp_error(got, want16.double(), floor=2.0 ** -10)
    assert d.max().item() <= 1.0 + 1e-6, d.max().item()            # never further than one ulp from what the reference returns
    assert (got == want16).float().mean().item() >= 0.98
    if mode == "max":
        assert torch.equal(got, want16)


@pytest.mark.parametrize("grid,out_grid,mode", [(24, 6, 0), (27, 7, 0), (24, 6, 1), (24, 6, 2), (24, 7, 3), (27, 7, 3), (24, 1, 3)])
def test_pooling_within_one_ulp_of_exact(vit_l, grid, out_grid, mode):
    cfg, _, rt = vit_l
    C_ = 256
    g = _gen(20 + grid + mode)
    x = (torch.randn(2, grid * grid + 1, C_, generator=g, device="cuda") * 2).bfloat16()       # one surplus row per frame (CLIP's class token)
    xs = x[:, :grid * grid].double().view(2, grid, grid, C_).permute(0, 3, 1, 2)
    stride = 4
    if mode == 0:
        exact = F.interpolate(xs, size=[out_grid, out_grid], mode="bilinear", align_corners=False)
    elif mode == 1:
        exact = F.avg_pool2d(xs, stride)
    elif mode == 2:
        exact = F.max_pool2d(xs, stride)
    else:
        exact = F.adaptive_avg_pool2d(xs, (out_grid, out_grid))
    exact = exact.permute(0, 2, 3, 1).reshape(2, out_grid * out_grid, C_)
    got = rt.pool(x, grid, out_grid, stride, mode)
    e = ulp_error(got, exact, floor=2.0 ** -10)
    _note(f"pool mode={mode} {grid}->{out_grid}", e, got, exact.bfloat16())
    assert e.max().item() <= 1.0 + 1e-6
    if mode == 2:
        assert torch.equal(got, exact.bfloat16())


def test_gathered_rows_route_is_the_direct_route(vit_l):
    """aha_vit_encode pools a compact (2*go)^2 grid of the rows bilinear pooling samples; it must equal pooling the full grid."""
    cfg, _, rt = vit_l
    g = _gen(31)
    x = (torch.randn(3, 576, 512, generator=g, device="cuda") * 2).bfloat16()
    direct = rt.pool(x, 24, 6, 4, 0)
    rows = rt.pool_gather_rows(x, 24, 6)
    assert rows.shape == (3, 144, 512)
    xs = x.view(3, 24, 24, 512)
    for cy in range(12):
        for cx in range(12):
            y, xx = 4 * (cy >> 1) + 1 + (cy & 1), 4 * (cx >> 1) + 1 + (cx & 1)
            assert torch.equal(rows[:, cy * 12 + cx], xs[:, y, xx])
    assert torch.equal(rt.pool(rows, 12, 6, 2, 0), direct)


# ---------------------------------------------------------------------------------------------------------------------
# encoder layers, teacher-forced
# ---------------------------------------------------------------------------------------------------------------------
def _ulp_at_scale(got, want):
    w = want.double()
    ref = w.abs().clamp_min(rms(w))
    return (got.double() - w).abs() / bf16_ulp(ref)


@pytest.mark.parametrize("which", ["vit_l", "so400m"])
def test_every_encoder_layer_teacher_forced(which, request):
    """Each encoder layer runs alone on the HIP path from the bf16 oracle's input to that layer (no depth amplification):
    LayerNorm -> QKV GEMM -> attention -> out-proj + residual -> LayerNorm -> fc1 + GELU -> fc2 + residual."""
    from oracle.vision_tower import OracleVision, preprocess
    cfg, w, rt = request.getfixturevalue(which)
    n = 2
    fr = make_frames(n, cfg.vision.image_size, seed=7)
    ov = OracleVision(cfg, {k: t.cpu() for k, t in w.items()}, torch.bfloat16)
    taps, final = ov.layer_inputs(preprocess(fr, torch.bfloat16))
    worst = 0.0
    for i, x_in in enumerate(taps):
        want = taps[i + 1] if i + 1 < len(taps) else final
        got = rt.vit_layers(x_in.reshape(-1, x_in.shape[-1]).cuda(), n, i, 1).cpu().view_as(want)
        e = _ulp_at_scale(got, want)
        _note(f"{which} encoder layer {i} teacher-forced", e)
        worst = max(worst, e.max().item())
    assert worst <= 6.0, worst                       # measured 3-4 (p99.9: 2); a wrong row, head or k-step is tens of ulps
    # and the whole tower from the pixels agrees with the chain of teacher-forced layers' input (same kernels, free-running)
    emb = rt.visual_embed(fr.cuda())
    tower = rt.tower_output(n).cpu().view_as(final)
    assert torch.isfinite(emb.float()).all()
    e = _ulp_at_scale(tower, final)
    _note(f"{which} tower free-running ({len(taps)} layers)", e)


def test_pool_and_layernorm_operators_refuse_shapes_their_kernels_cannot_serve(vit_l):
    """aha_pool_forward works in 8-channel chunks and, for avg / max pooling, reads rows (oy*stride + dy): channel counts
    that are not multiples of 8, a non-positive stride or a window that leaves the grid are refused, not executed; aha_layernorm_forward
    refuses row strides below the row length or off the 16-byte grid."""
    from aha_amd.runtime import AhaError
    import ctypes as C
    cfg, _, rt = vit_l
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(1, 24 * 24, 16, device="cuda", dtype=torch.bfloat16)
    for ch, grid, out_grid, stride, mode in ((12, 24, 6, 4, 0), (4, 24, 6, 4, 1), (16, 24, 6, 0, 1), (16, 24, 7, 4, 2), (16, 24, 25, 1, 0), (16, 24, 25, 1, 3)):
        xs = torch.zeros(1, grid * grid, ch, device="cuda", dtype=torch.bfloat16)
        with pytest.raises(AhaError):
            rt.pool(xs, grid, out_grid, stride, mode)
    assert rt.pool(x, 24, 6, 4, 1).shape == (1, 36, 16)              # the valid neighbour of those calls still runs
    w = torch.ones(64, device="cuda", dtype=torch.bfloat16)
    xin, out = torch.zeros(4, 64, device="cuda", dtype=torch.bfloat16), torch.zeros(4, 64, device="cuda", dtype=torch.bfloat16)
    for ldx, ldo in ((32, 64), (64, 56), (68, 64), (64, 60)):
        rc = rt.lib.aha_layernorm_forward(rt.ctx, xin.data_ptr(), ldx, w.data_ptr(), w.data_ptr(), out.data_ptr(), ldo, 4, 64, C.c_float(1e-6), st)
        assert rc != 0, (ldx, ldo)
    assert rt.lib.aha_layernorm_forward(rt.ctx, xin.data_ptr(), 64, w.data_ptr(), w.data_ptr(), out.data_ptr(), 64, 4, 64, C.c_float(1e-6), st) == 0
