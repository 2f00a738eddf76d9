"""Pin oracle/qwen2_live.py and oracle/vision_tower.py against local transformers
(Qwen2Model + DynamicCache, SiglipVisionModel) live, and against the committed fixtures."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
import aha_amd  # noqa: F401
from aha_amd.config import preset
from aha_amd.synth import make_frames, make_weights
from oracle.cache_policies import GrowingPolicy, SinkPolicy, SlidingPolicy, StaticPolicy
from oracle.qwen2_live import OracleLM, frame_scores
from oracle.vision_tower import OracleVision, preprocess


def _hf_qwen2(lm, w, dt, impl):
    from transformers import Qwen2Config, Qwen2Model
    hc = Qwen2Config(hidden_size=lm.hidden_size, num_hidden_layers=lm.num_hidden_layers,
                     num_attention_heads=lm.num_attention_heads, num_key_value_heads=lm.num_key_value_heads,
                     intermediate_size=lm.intermediate_size, vocab_size=lm.vocab_size, rope_theta=lm.rope_theta,
                     rms_norm_eps=lm.rms_norm_eps, max_position_embeddings=lm.max_position_embeddings,
                     head_dim=lm.head_dim, attn_implementation=impl)
    m = Qwen2Model(hc).to(dt).eval()
    m.load_state_dict({k[len("model."):]: v for k, v in w.items() if k.startswith("model.")})
    return hc, m


def test_lm_matches_golden_fp32():
    gold = np.load(os.path.join(GOLDEN, "qwen2_tiny_steps.npz"))
    cfg = preset("tiny")
    w = make_weights(cfg, dtype=torch.float32, jitter=True)
    o = OracleLM(cfg.lm, w, torch.float32)
    cache = GrowingPolicy()
    g = torch.Generator().manual_seed(3)
    for step, T in enumerate(gold["steps"].tolist()):
        x = torch.randn(1, T, cfg.lm.hidden_size, generator=g)
        y = o.step(x, cache)["hidden"]
        np.testing.assert_allclose(y.numpy(), gold[f"hidden_s{step}"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("preset_name", ["tiny", "tiny128"])
@pytest.mark.parametrize("impl", ["sdpa", "eager"])
def test_lm_matches_transformers_live(preset_name, impl):
    from transformers import DynamicCache
    cfg = preset(preset_name)
    w32 = make_weights(cfg, dtype=torch.float32, jitter=True)
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 0.07)):
        w = {k: v.to(dt) for k, v in w32.items()}
        hc, m = _hf_qwen2(cfg.lm, w, dt, impl)
        o = OracleLM(cfg.lm, w, dt, attn_impl=impl)
        ch, co = DynamicCache(config=hc), GrowingPolicy()
        g = torch.Generator().manual_seed(11)
        for T in [7, 4, 4, 1, 4]:
            x = torch.randn(1, T, cfg.lm.hidden_size, generator=g).to(dt)
            with torch.no_grad():
                yh = m(inputs_embeds=x, past_key_values=ch, use_cache=True).last_hidden_state
            yo = o.step(x, co)["hidden"]
            assert ch.get_seq_length() == co.get_seq_length()
            assert (yh.float() - yo.float()).abs().max().item() <= tol


def test_vision_matches_golden_and_transformers():
    from transformers import SiglipVisionConfig, SiglipVisionModel
    gold = np.load(os.path.join(GOLDEN, "siglip_tiny.npz"))
    cfg = preset("tiny")
    v = cfg.vision
    w = make_weights(cfg, dtype=torch.float32, jitter=True)
    fr = make_frames(2, v.image_size, seed=0)
    ov = OracleVision(cfg, w, torch.float32)
    y = ov.tower(preprocess(fr, torch.float32))
    np.testing.assert_allclose(y.numpy(), gold["hidden"], rtol=0, atol=2e-5)
    vc = SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                            num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                            image_size=v.image_size, patch_size=v.patch_size, layer_norm_eps=v.layer_norm_eps,
                            hidden_act="gelu_pytorch_tanh", attn_implementation="sdpa")
    for dt in (torch.float32, torch.bfloat16):
        vm = SiglipVisionModel(vc).to(dt).eval()
        vm.load_state_dict({k[len("vision."):]: t.to(dt) for k, t in w.items() if k.startswith("vision.")}, strict=False)
        px = preprocess(fr, dt)
        with torch.no_grad():
            hs = vm(pixel_values=px, output_hidden_states=True).hidden_states[-1]
        yo = OracleVision(cfg, w, dt).tower(px)
        assert (hs.float() - yo.float()).abs().max().item() <= (2e-5 if dt == torch.float32 else 0.04)


def test_siglip_pooling_head_matches_transformers_pooler_output():
    """models/vision_live.py:26-31 (frame_token_cls): `pooler_output` of the SigLIP vision model = its attention-pooling head on
    the post-layernormed tokens.  The oracle's restatement against the local transformers SiglipVisionModel, fp32 and bf16,
    and the encode contract's token order (class token first, then the pooled grid)."""
    from transformers import SiglipVisionConfig, SiglipVisionModel
    from aha_amd.synth import make_vision_head_weights
    from oracle.vision_tower import siglip_pooling_head, vision_live_encode
    cfg = preset("tiny")
    v = cfg.vision
    w = make_weights(cfg, dtype=torch.float32, jitter=True)
    w.update(make_vision_head_weights(cfg, dtype=torch.float32))
    fr = make_frames(2, v.image_size, seed=3)
    vc = SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                            num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                            image_size=v.image_size, patch_size=v.patch_size, layer_norm_eps=v.layer_norm_eps,
                            hidden_act="gelu_pytorch_tanh", attn_implementation="sdpa")
    for dt in (torch.float32, torch.bfloat16):
        vm = SiglipVisionModel(vc).to(dt).eval()
        missing = vm.load_state_dict({k[len("vision."):]: t.to(dt) for k, t in w.items() if k.startswith("vision.")}, strict=True)
        px = preprocess(fr, dt)
        with torch.no_grad():
            out = vm(pixel_values=px)
        ov = OracleVision(cfg, w, dt)
        x = torch.nn.functional.layer_norm(ov.tower(px), (v.hidden_size,), ov.w["vision.post_layernorm.weight"],
                                           ov.w["vision.post_layernorm.bias"], v.layer_norm_eps)
        assert (x.float() - out.last_hidden_state.float()).abs().max().item() <= (3e-5 if dt == torch.float32 else 0.06)
        got = siglip_pooling_head(ov, out.last_hidden_state)           # same input: isolates the head
        tol = 3e-5 if dt == torch.float32 else 0.03 * max(1.0, out.pooler_output.float().abs().max().item())
        assert (got.float() - out.pooler_output.float()).abs().max().item() <= tol, dt
    ov = OracleVision(cfg, w, torch.float32)
    plw, plb = w["vision.post_layernorm.weight"], w["vision.post_layernorm.bias"]
    both = vision_live_encode(ov, fr, plw, plb, (2, 2), frame_token_cls=True).view(2, 5, -1)
    only_sp = vision_live_encode(ov, fr, plw, plb, (2, 2)).view(2, 4, -1)
    only_cls = vision_live_encode(ov, fr, plw, plb, None, frame_token_cls=True).view(2, 1, -1)
    assert torch.allclose(both[:, 1:], only_sp, atol=1e-6) and torch.allclose(both[:, :1], only_cls, atol=1e-6)


def test_clip_tower_matches_transformers_and_encode_contract():
    """The CLIP half of models/vision_live.py (_clip_vision_encode): the oracle's CLIP tower against local transformers
    CLIPVisionModel.last_hidden_state (class token, pre_layrnorm, quick_gelu, no post-layernorm), and the encode contract
    (OpenAI mean/std, class token dropped, adaptive average pool) against the same calls on the transformers output."""
    import dataclasses
    import math
    import torch.nn.functional as F
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from oracle.vision_tower import OracleCLIPVision, clip_live_encode, preprocess_clip
    base = preset("tiny")
    cfg = dataclasses.replace(base, vision=dataclasses.replace(base.vision, kind="clip", layer_norm_eps=1e-5), name="tiny_clip")
    v = cfg.vision
    w = make_weights(cfg, dtype=torch.float32, jitter=True)
    assert w["vision.embeddings.position_embedding.weight"].shape == (v.num_patches + 1, v.hidden_size)
    fr = make_frames(2, v.image_size, seed=3)
    vc = CLIPVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                          num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size,
                          layer_norm_eps=v.layer_norm_eps, hidden_act="quick_gelu", attn_implementation="sdpa")
    for dt in (torch.float32, torch.bfloat16):
        vm = CLIPVisionModel(vc).to(dt).eval()
        sd = {k[len("vision."):]: t.to(dt) for k, t in w.items() if k.startswith("vision.")}
        target = vm.vision_model if hasattr(vm, "vision_model") else vm
        missing = target.load_state_dict(sd, strict=False)
        assert not [k for k in missing.missing_keys if "post_layernorm" not in k and "position_ids" not in k], missing.missing_keys
        px = preprocess_clip(fr, dt)
        with torch.no_grad():
            hs = vm(pixel_values=px).last_hidden_state
        ov = OracleCLIPVision(cfg, w, dt)
        yo = ov.tower(px)
        assert hs.shape == yo.shape == (2, v.num_patches + 1, v.hidden_size)
        assert (hs.float() - yo.float()).abs().max().item() <= (2e-5 if dt == torch.float32 else 0.06)
        if dt == torch.float32:
            # the reference's own post-processing of last_hidden_state (vision_live.py:40-49) + the connector
            s_ = int(math.sqrt(hs.shape[1]))
            sp = F.adaptive_avg_pool2d(hs[:, 1:].reshape(2, s_, s_, -1).permute(0, 3, 1, 2), (2, 2)).flatten(2, 3).permute(0, 2, 1)
            want = ov.connector(sp).reshape(-1, cfg.lm.hidden_size)
            got = clip_live_encode(ov, fr, (2, 2))
            assert got.shape == want.shape and (got - want).abs().max().item() <= 2e-5


def test_visual_embed_shapes_and_pool():
    for name, tf in (("tiny", 4), ("tiny128", 9)):
        cfg = preset(name)
        assert cfg.frame_num_tokens == tf
        w = make_weights(cfg, dtype=torch.float32, jitter=True)
        ov = OracleVision(cfg, w, torch.float32)
        e = ov.visual_embed(make_frames(3, cfg.vision.image_size, seed=1))
        assert e.shape == (3 * tf, cfg.lm.hidden_size)
    assert preset("bench").frame_num_tokens == 36 and preset("ref").frame_num_tokens == 49


@pytest.mark.parametrize("policy", ["sink", "sliding", "static", "none"])
def test_policies_run_through_lm_and_positions(policy):
    cfg = preset("tiny")
    w = make_weights(cfg, dtype=torch.float32, jitter=True)
    o = OracleLM(cfg.lm, w, torch.float32)
    cache = {"sink": SinkPolicy(24, 4), "sliding": SlidingPolicy(24), "static": StaticPolicy(24),
             "none": GrowingPolicy()}[policy]
    g = torch.Generator().manual_seed(5)
    lens = []
    for T in [10, 4, 4, 4, 4, 4, 4]:
        out = o.step(torch.randn(1, T, cfg.lm.hidden_size, generator=g), cache)
        s = frame_scores(out)
        assert s.shape == (1, 3) and torch.isfinite(s).all()
        assert 0 < s[0, 0] < 1 and 0 < s[0, 1] < 1 and s[0, 2] > 0
        lens.append(cache.get_seq_length())
    want = {"sink": [10, 14, 18, 22, 24, 24, 24], "sliding": [10, 14, 18, 22, 24, 24, 24],
            "static": [10] * 7, "none": [10, 14, 18, 22, 26, 30, 34]}[policy]
    assert lens == want


def test_static_frozen_sees_prefix_only():
    """After the first call a StaticPolicy step must not depend on earlier frame steps
    (test/static_cache.py:26-36: the cache never changes again)."""
    cfg = preset("tiny")
    w = make_weights(cfg, dtype=torch.float32, jitter=True)
    o = OracleLM(cfg.lm, w, torch.float32)
    g = torch.Generator().manual_seed(9)
    prefix = torch.randn(1, 6, cfg.lm.hidden_size, generator=g)
    a, b = torch.randn(1, 4, cfg.lm.hidden_size, generator=g), torch.randn(1, 4, cfg.lm.hidden_size, generator=g)
    c1, c2 = StaticPolicy(32), StaticPolicy(32)
    o.step(prefix, c1), o.step(prefix, c2)
    o.step(a, c1)
    assert torch.equal(o.step(b, c1)["hidden"], o.step(b, c2)["hidden"])


def test_flash_attn2_mask_semantics_of_the_oracle():
    """attn_semantics="fa2" (the reference's default attn_implementation, models/arguments_live.py:30): bottom-right aligned
    causal mask.  It equals the trailing rule for every policy that returns the new keys; under a frozen TrulyStaticCache the
    first T - L new tokens see no key (output 0, finite), the other rows differ from the all-visible rule, and the scores the
    drivers read at position -1 are identical."""
    import torch
    from aha_amd.config import preset
    from aha_amd.synth import make_weights
    from oracle.cache_policies import make_policy
    from oracle.qwen2_live import OracleLM, frame_scores
    cfg = preset("tiny")
    w = make_weights(cfg, dtype=torch.float32, jitter=True)
    g = torch.Generator().manual_seed(0)
    pre = torch.randn(1, 5, cfg.lm.hidden_size, generator=g) * 0.5
    xs = [torch.randn(1, 9, cfg.lm.hidden_size, generator=g) * 0.5 for _ in range(3)]
    for policy in ("static", "default_sink", None):
        outs = {}
        for sem in ("trailing", "fa2"):
            o = OracleLM(cfg.lm, w, torch.float32, attn_semantics=sem)
            pol = make_policy(policy, 16, 2)
            o.step(pre, pol)
            outs[sem] = [o.step(x, pol) for x in xs]
        for a, b in zip(outs["trailing"], outs["fa2"]):
            assert torch.equal(frame_scores(a), frame_scores(b)) and torch.isfinite(b["hidden"]).all()
            if policy == "static":
                assert not torch.equal(a["hidden"][:, 0], b["hidden"][:, 0])          # row 0 sees nothing under fa2 (9 new tokens, 5 keys)
            else:
                assert torch.equal(a["hidden"], b["hidden"])


def test_stable_regime_is_stable_on_a_narrow_deep_model():
    """aha_amd.synth's regime="stable" weight set on a 28-layer model narrow enough for the CPU suite: two bf16 evaluations of
    the reference arithmetic (sdpa vs eager) and the fp32 evaluation agree to a small fraction of the frame-to-frame spread of
    the scores, while SURVEY.md 8d's plain normal(0, 0.02) set (a chaotic map at this depth) does not.  The full-width figures
    are in profiles/r04_stable_regime_cpu*.json (tests/stable_regime_check.py); tests/test_gpu_flat_parity.py holds the HIP
    path to the flat 1e-3 in that regime."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from stable_regime_check import stability_stats, synth_embeds
    from aha_amd.config import LiveConfig, LMConfig, VisionConfig
    from aha_amd.synth import make_weights
    cfg = LiveConfig(vision=VisionConfig(image_size=56, patch_size=14, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                                         intermediate_size=256),
                     lm=LMConfig(hidden_size=512, num_hidden_layers=28, num_attention_heads=8, num_key_value_heads=2, head_dim=64,
                                 intermediate_size=1024, vocab_size=512), video_pooling_stride=2, name="narrow28")
    ratio = {}
    for regime in ("default", "stable"):
        w = make_weights(cfg, dtype=torch.bfloat16, regime=regime)
        steps = synth_embeds(32, 9, 512, scale=1.0 if regime == "stable" else 0.05)
        st = stability_stats(cfg, w, steps, log=lambda *a: None)
        noise = torch.tensor([st["sdpa_vs_eager_max"], st["bf16_vs_fp32_max"]]).max(0).values
        ratio[regime] = (noise / torch.tensor(st["score_std"])).max().item()
        if regime == "stable":
            assert noise.max().item() <= 1e-3, st
    assert ratio["stable"] <= 0.08 and ratio["default"] >= 2.0 * ratio["stable"], ratio
