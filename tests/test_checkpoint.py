"""Checkpoint loader: llava-ov / PEFT key layouts -> runtime names, LoRA merged at load; and the
prediction JSON writer.  Parity: merged weights vs the unmerged evaluation W x + (alpha/r) B (A x)
that the reference runs (models/modeling_live.py:171-179), through the oracle LM."""
import json
import os

import pytest
import torch

import aha_amd  # noqa: F401
from aha_amd.checkpoint import canonical_name, load_checkpoint, merge_lora, write_predictions
from aha_amd.config import preset
from aha_amd.synth import make_weights, tensor_specs


def _hf_key(name):
    if name.startswith("vision."):
        return "model.vision_tower.vision_tower.vision_model." + name[len("vision."):]
    if name.startswith("mm_projector."):
        return "model." + name
    return name


def test_roundtrip_through_safetensors_with_lora(tmp_path):
    from safetensors.torch import save_file
    cfg = preset("tiny")
    w = make_weights(cfg, dtype=torch.bfloat16, jitter=True)
    base_dir, lora_dir = tmp_path / "base", tmp_path / "lora"
    base_dir.mkdir(), lora_dir.mkdir()
    hf = {_hf_key(k): v.contiguous() for k, v in w.items()}
    hf["model.image_newline"] = torch.zeros(cfg.lm.hidden_size, dtype=torch.bfloat16)            # present upstream, unused here
    hf["model.vision_tower.vision_tower.vision_model.post_layernorm.weight"] = torch.ones(cfg.vision.hidden_size, dtype=torch.bfloat16)
    keys = sorted(hf)
    save_file({k: hf[k] for k in keys[::2]}, str(base_dir / "model-00001-of-00002.safetensors"))
    save_file({k: hf[k] for k in keys[1::2]}, str(base_dir / "model-00002-of-00002.safetensors"))
    # PEFT adapter: r=4, alpha=8 on q_proj / down_proj of layer 0, plus a saved head
    g = torch.Generator().manual_seed(0)
    r, alpha = 4, 8.0
    ad = {}
    for mod in ("model.layers.0.self_attn.q_proj", "model.layers.1.mlp.down_proj"):
        W = w[mod + ".weight"]
        ad[f"base_model.model.{mod}.lora_A.weight"] = (torch.randn(r, W.shape[1], generator=g) * 0.05).bfloat16()
        ad[f"base_model.model.{mod}.lora_B.weight"] = (torch.randn(W.shape[0], r, generator=g) * 0.05).bfloat16()
    new_head = (torch.randn(1, cfg.lm.hidden_size, generator=g) * 0.02).bfloat16()
    ad["base_model.model.relevance_head.weight"] = new_head
    save_file(ad, str(lora_dir / "adapter_model.safetensors"))
    json.dump({"r": r, "lora_alpha": alpha}, open(lora_dir / "adapter_config.json", "w"))

    got = load_checkpoint(str(base_dir), str(lora_dir))
    names = {n for n, _, _ in tensor_specs(cfg)}
    assert set(got) == names                                                   # nothing missing, nothing extra
    assert torch.equal(got["relevance_head.weight"], new_head)
    for mod in ("model.layers.0.self_attn.q_proj", "model.layers.1.mlp.down_proj"):
        A, B = ad[f"base_model.model.{mod}.lora_A.weight"].float(), ad[f"base_model.model.{mod}.lora_B.weight"].float()
        want = (w[mod + ".weight"].float() + (alpha / r) * (B @ A)).bfloat16()
        assert torch.equal(got[mod + ".weight"], want)
    untouched = "model.layers.0.mlp.up_proj.weight"
    assert torch.equal(got[untouched], w[untouched])
    plain = load_checkpoint(str(base_dir))
    assert all(torch.equal(plain[k], w[k]) for k in names)

    # merged vs unmerged evaluation through the LM (fp32 oracle): the only difference is rounding W' to bf16
    from oracle.cache_policies import GrowingPolicy
    from oracle.qwen2_live import OracleLM
    import torch.nn.functional as F
    x = torch.randn(1, 6, cfg.lm.hidden_size, generator=g) * 0.5
    merged = OracleLM(cfg.lm, {k: v.float() for k, v in got.items()}, torch.float32).step(x, GrowingPolicy())["hidden"]
    exact_w = {k: v.float() for k, v in w.items()}
    for mod in ("model.layers.0.self_attn.q_proj", "model.layers.1.mlp.down_proj"):
        A, B = ad[f"base_model.model.{mod}.lora_A.weight"].float(), ad[f"base_model.model.{mod}.lora_B.weight"].float()
        exact_w[mod + ".weight"] = exact_w[mod + ".weight"] + (alpha / r) * (B @ A)        # == W x + (alpha/r) B (A x)
    exact_w["relevance_head.weight"] = new_head.float()
    unmerged = OracleLM(cfg.lm, exact_w, torch.float32).step(x, GrowingPolicy())["hidden"]
    assert (merged - unmerged).abs().max().item() <= 0.02                      # bf16 ulp of the two merged matrices


def test_canonical_names_and_merge_errors():
    assert canonical_name("base_model.model.model.layers.3.self_attn.v_proj.base_layer.weight") == "model.layers.3.self_attn.v_proj.weight"
    assert canonical_name("base_model.model.informative_head.modules_to_save.default.weight") == "informative_head.weight"
    assert canonical_name("model.vision_tower.vision_tower.vision_model.encoder.layers.0.mlp.fc1.bias") == "vision.encoder.layers.0.mlp.fc1.bias"
    assert canonical_name("model.vision_tower.vision_tower.vision_model.head.probe") is None
    assert canonical_name("model.image_newline") is None
    with pytest.raises(KeyError):
        merge_lora({}, {"base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight": torch.zeros(2, 4),
                        "base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight": torch.zeros(4, 2)}, 8.0)


def test_write_predictions_schema(tmp_path):
    rec = {"video_uuid": "v0", "model_response_list": [], "video_duration": 3.0, "true_frames_list": [0, 1, 2],
           "debug_data": [{"time": 0.0, "informative_score": 0.123, "relevance_score": 0.5, "uncertainty_score": 1.01}]}
    p = tmp_path / "pred.json"
    write_predictions(str(p), [rec])
    assert json.load(open(p))[0] == rec
    with pytest.raises(ValueError):
        write_predictions(str(p), [{"video_uuid": "v0"}])
