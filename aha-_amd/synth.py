"""Seeded synthetic weights and frames (there is no network for checkpoints or datasets).

Tensor names follow the checkpoints the reference loads (HF Qwen2 names under
``model.``/``lm_head``, the three head names of
models/live_llava/video_head_live_llava_qwen.py:83-85, ``mm_projector.{0,2}`` of the
llava ``mlp2x_gelu`` projector, SigLIP names under ``vision.``), so the same dict format
serves a real safetensors loader later.  Per-tensor seed = crc32(name) (SURVEY.md 8d:
never Python's salted ``hash``).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, Tuple

import torch

from .config import LiveConfig


def tensor_specs(cfg: LiveConfig) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """Yield (name, shape, kind) with kind in {'w','b','norm_w','norm_b','qkv_b','emb'}."""
    v, lm = cfg.vision, cfg.lm
    Dv, H = v.hidden_size, lm.hidden_size
    yield "vision.embeddings.patch_embedding.weight", (Dv, 3, v.patch_size, v.patch_size), "w"
    if v.kind == "clip":                 # CLIPVisionEmbeddings: bias-free patch conv, class token, Np + 1 positions, pre-LN
        yield "vision.embeddings.class_embedding", (Dv,), "emb"
        yield "vision.embeddings.position_embedding.weight", (v.num_patches + 1, Dv), "emb"
        yield "vision.pre_layrnorm.weight", (Dv,), "norm_w"
        yield "vision.pre_layrnorm.bias", (Dv,), "norm_b"
    else:
        yield "vision.embeddings.patch_embedding.bias", (Dv,), "b"
        yield "vision.embeddings.position_embedding.weight", (v.num_patches, Dv), "emb"
    for i in range(v.num_hidden_layers):
        p = f"vision.encoder.layers.{i}."
        yield p + "layer_norm1.weight", (Dv,), "norm_w"
        yield p + "layer_norm1.bias", (Dv,), "norm_b"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield p + f"self_attn.{n}.weight", (Dv, Dv), "w"
            yield p + f"self_attn.{n}.bias", (Dv,), "b"
        yield p + "layer_norm2.weight", (Dv,), "norm_w"
        yield p + "layer_norm2.bias", (Dv,), "norm_b"
        yield p + "mlp.fc1.weight", (v.intermediate_size, Dv), "w"
        yield p + "mlp.fc1.bias", (v.intermediate_size,), "b"
        yield p + "mlp.fc2.weight", (Dv, v.intermediate_size), "w"
        yield p + "mlp.fc2.bias", (Dv,), "b"
    yield "mm_projector.0.weight", (H, Dv), "w"
    yield "mm_projector.0.bias", (H,), "b"
    yield "mm_projector.2.weight", (H, H), "w"
    yield "mm_projector.2.bias", (H,), "b"
    yield "model.embed_tokens.weight", (lm.vocab_size, H), "emb"
    qd, kd = lm.num_attention_heads * lm.head_dim, lm.num_key_value_heads * lm.head_dim
    for i in range(lm.num_hidden_layers):
        p = f"model.layers.{i}."
        yield p + "input_layernorm.weight", (H,), "norm_w"
        yield p + "self_attn.q_proj.weight", (qd, H), "w"
        yield p + "self_attn.q_proj.bias", (qd,), "qkv_b"
        yield p + "self_attn.k_proj.weight", (kd, H), "w"
        yield p + "self_attn.k_proj.bias", (kd,), "qkv_b"
        yield p + "self_attn.v_proj.weight", (kd, H), "w"
        yield p + "self_attn.v_proj.bias", (kd,), "qkv_b"
        yield p + "self_attn.o_proj.weight", (H, qd), "w"
        yield p + "post_attention_layernorm.weight", (H,), "norm_w"
        yield p + "mlp.gate_proj.weight", (lm.intermediate_size, H), "w"
        yield p + "mlp.up_proj.weight", (lm.intermediate_size, H), "w"
        yield p + "mlp.down_proj.weight", (H, lm.intermediate_size), "w"
    yield "model.norm.weight", (H,), "norm_w"
    yield "lm_head.weight", (lm.vocab_size, H), "w"
    yield "informative_head.weight", (2, H), "w"
    yield "relevance_head.weight", (1, H), "w"
    yield "uncertainty_head.weight", (1, H), "w"


def stable_regime_scale(cfg: LiveConfig, name: str) -> float:
    """Multiplier on a tensor's normal(0, std) draw in the ``regime="stable"`` weight set.

    SURVEY.md 8d's plain normal(0, 0.02) set makes 28 untrained layers a chaotic map: every residual branch is as large as
    the stream it is added to, so one flipped bf16 rounding is amplified layer after layer and two equally valid bf16
    evaluations of the REFERENCE arithmetic (sdpa vs eager) land 1e-2 apart in score - no implementation can then be held
    to the north star's 1e-3.  A trained checkpoint is not like that, and this regime restores the property with the
    standard depth-scaled initialisation: the output projection of every residual branch (LM o_proj / down_proj, tower
    out_proj / fc2) is scaled by 1/sqrt(2 L), the stream itself starts at unit scale (token embeddings, patch embedding,
    second projector matrix), and the three heads are sized so that their bf16 logits stay below 0.25 (0.125 for the
    log-variance), where one bf16 ulp moves a score by less than 2.5e-4 (6e-4).  The arithmetic, shapes and dtypes are
    untouched; tests/stable_regime_check.py shows on the CPU that the reference arithmetic itself is then stable
    (|bf16 sdpa - bf16 eager| and |bf16 - fp32| well under 1e-3 over 64 full-depth frames) with non-degenerate scores."""
    v, lm = cfg.vision, cfg.lm
    if name.startswith("model.layers.") and name.endswith(("o_proj.weight", "down_proj.weight")):
        return (2.0 * lm.num_hidden_layers) ** -0.5
    if name.startswith("vision.encoder.layers.") and name.endswith(("out_proj.weight", "fc2.weight")):
        return (2.0 * v.num_hidden_layers) ** -0.5
    if name == "model.embed_tokens.weight":
        return 50.0                                  # unit-scale token rows beside unit-scale frame embeddings
    if name == "vision.embeddings.patch_embedding.weight":
        return 4.0
    if name == "mm_projector.2.weight":
        return 4.0
    if name in ("informative_head.weight", "relevance_head.weight"):
        return 0.03 / (0.02 * lm.hidden_size ** 0.5)
    if name == "uncertainty_head.weight":
        return 0.01 / (0.02 * lm.hidden_size ** 0.5)
    return 1.0


def calibrated_heads(hidden: torch.Tensor, *, spans=(0.3, 0.3, 0.08), dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Head weights that read COHERENT features of the final hidden state, as a trained head does.

    A random head direction w sees the frame-to-frame signal and the bf16 rounding noise of the hidden state through the
    same incoherent sum over 3,584 channels, so noise / spread of its score is the per-channel relative noise of a 28-layer
    bf16 residual stream (~0.02 median, ~0.06 worst of 64 frames - measured on the reference arithmetic itself, sdpa vs
    eager, profiles/r04_stable_regime_cpu*.json) whatever its scale.  A trained head is aligned with a feature direction
    along which the hidden state moves coherently from frame to frame; the signal then adds up over the channels and the
    rounding noise does not.  This builds such heads from data, in closed form: `hidden` = final (normalised) hidden rows
    [n, H] of calibration frames; the three leading principal directions of their frame-to-frame variation, made
    orthogonal to the mean row (so the logits are centred), become  informative = (-v1, +v1)/2, relevance = v2,
    uncertainty = v3, each scaled so that the largest calibration logit equals its `spans` entry (bf16 logits of that
    size: one ulp moves a score by <= 2.5e-4)."""
    X = hidden.detach().to("cpu", torch.float64)
    mu = X.mean(0)
    Xc = X - mu
    _, _, Vt = torch.linalg.svd(Xc, full_matrices=False)
    vs = []
    for i in range(3):
        v = Vt[i].clone()
        v -= (v @ mu) / (mu @ mu) * mu                       # no response to the common component: centred logits
        for u in vs:
            v -= (v @ u) / (u @ u) * u
        vs.append(v)
    out = {}
    g = [spans[i] / (X @ vs[i]).abs().max().clamp_min(1e-12) for i in range(3)]
    out["informative_head.weight"] = torch.stack([-0.5 * g[0] * vs[0], 0.5 * g[0] * vs[0]]).to(dtype)
    out["relevance_head.weight"] = (g[1] * vs[1])[None].to(dtype)
    out["uncertainty_head.weight"] = (g[2] * vs[2])[None].to(dtype)
    return out


def make_weights(cfg: LiveConfig, *, device="cpu", dtype=torch.bfloat16, jitter: bool = False,
                 std: float = 0.02, skip_lm_head: bool = False, regime: str = "default") -> Dict[str, torch.Tensor]:
    """normal(0, std) matrices, norm weights 1, biases 0 except q/k/v bias ~ normal(0, std)
    (SURVEY.md 8d config 2).  ``jitter`` perturbs norm weights/biases and linear biases so
    parity tests exercise those terms too.  Values are drawn in fp32 then cast to ``dtype``.
    ``regime="stable"``: the same draws with the per-tensor multipliers of ``stable_regime_scale``.
    """
    assert regime in ("default", "stable")
    out: Dict[str, torch.Tensor] = {}
    dev = torch.device(device)
    for name, shape, kind in tensor_specs(cfg):
        if skip_lm_head and name == "lm_head.weight":
            continue
        g = torch.Generator(device=dev)
        g.manual_seed(zlib.crc32(name.encode()))
        if kind in ("w", "emb", "qkv_b"):
            t = torch.empty(shape, device=dev, dtype=torch.float32).normal_(0.0, std, generator=g)
            if regime == "stable":
                t *= stable_regime_scale(cfg, name)
        elif kind == "norm_w":
            t = torch.ones(shape, device=dev, dtype=torch.float32)
            if jitter:
                t += torch.empty(shape, device=dev, dtype=torch.float32).normal_(0.0, 0.1, generator=g)
        else:  # 'b', 'norm_b'
            t = torch.zeros(shape, device=dev, dtype=torch.float32)
            if jitter:
                t += torch.empty(shape, device=dev, dtype=torch.float32).normal_(0.0, std, generator=g)
        out[name] = t.to(dtype)
    return out


def make_vision_head_weights(cfg: LiveConfig, *, device="cpu", dtype=torch.bfloat16, std: float = 0.05) -> Dict[str, torch.Tensor]:
    """The SigLIP vision model's tail that only the models/vision_live.py contract reads: post_layernorm and the attention-
    pooling head (`head.*`: probe, nn.MultiheadAttention in/out projections, layernorm, MLP).  Seeded like make_weights."""
    v = cfg.vision
    d, f = v.hidden_size, v.intermediate_size
    specs = [("vision.post_layernorm.weight", (d,), "norm_w"), ("vision.post_layernorm.bias", (d,), "b"),
             ("vision.head.probe", (1, 1, d), "probe"),
             ("vision.head.attention.in_proj_weight", (3 * d, d), "w"), ("vision.head.attention.in_proj_bias", (3 * d,), "b"),
             ("vision.head.attention.out_proj.weight", (d, d), "w"), ("vision.head.attention.out_proj.bias", (d,), "b"),
             ("vision.head.layernorm.weight", (d,), "norm_w"), ("vision.head.layernorm.bias", (d,), "b"),
             ("vision.head.mlp.fc1.weight", (f, d), "w"), ("vision.head.mlp.fc1.bias", (f,), "b"),
             ("vision.head.mlp.fc2.weight", (d, f), "w"), ("vision.head.mlp.fc2.bias", (d,), "b")]
    out: Dict[str, torch.Tensor] = {}
    dev = torch.device(device)
    for name, shape, kind in specs:
        g = torch.Generator(device=dev)
        g.manual_seed(zlib.crc32(name.encode()))
        t = torch.empty(shape, device=dev, dtype=torch.float32).normal_(0.0, 1.0 if kind == "probe" else (0.1 if kind == "norm_w" else std), generator=g)
        if kind == "norm_w":
            t += 1.0
        out[name] = t.to(dtype)
    return out


def make_frames(n: int, resolution: int, *, seed: int = 0, device="cpu", tint: bool = False) -> torch.Tensor:
    """uint8 [n,3,S,S] RGB CHW, the layout load_video_for_testing hands to the driver
    (test/inference.py:497-582).  `tint`: half-contrast noise plus a per-frame offset on every colour channel (three
    frame-level features - brightness and two colour balances - that a frame's tokens carry coherently), for the
    calibrated-head parity tests; the default is full-range iid noise."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    if not tint:
        f = torch.randint(0, 256, (n, 3, resolution, resolution), generator=g, dtype=torch.uint8)
        return f.to(device)
    f = torch.randint(64, 192, (n, 3, resolution, resolution), generator=g, dtype=torch.int16)
    off = torch.randint(-64, 64, (n, 3, 1, 1), generator=g, dtype=torch.int16)
    return (f + off).clamp_(0, 255).to(torch.uint8).to(device)


def make_token_ids(n: int, vocab: int, *, seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randint(0, vocab, (1, n), generator=g, dtype=torch.long)
