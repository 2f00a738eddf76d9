"""Wire / on-disk formats of the path.

* `load_checkpoint`: read `*.safetensors` shards of the base model (llava-onevision-qwen2-7b-ov
  naming) and, optionally, a PEFT LoRA adapter, and return the flat `{name: tensor}` dict the runtime
  consumes (names of aha_amd.synth.tensor_specs).  The reference keeps the adapter UNMERGED at
  inference (`PeftModel.from_pretrained(..., is_trainable=False)`, models/modeling_live.py:171-179),
  i.e. every projection runs `W x + (alpha/r) B (A x)`.  Here the adapter is merged once at load,
  `W' = W + (alpha/r) * B @ A` in fp32, cast to bf16 once (SURVEY.md section 7 "LoRA"): the streamed
  weights stay 13 GB and the step has no extra low-rank GEMMs; the difference from the unmerged
  evaluation is bf16 rounding of W' (covered by tests/test_checkpoint.py).
* `modules_to_save` tensors of the adapter (the three heads, `mm_projector`, `lm_head`;
  models/arguments_live.py:18) override the base tensors.
* `write_predictions`: the prediction JSON of test/inference.py:697-711.

Name mapping is by suffix rules, so both the HF layout of the merged model and PEFT's
`base_model.model.` prefix are accepted.  [upstream] the exact checkpoint key names come from the
llava-ov / peft releases, which are not available offline; the rules below are the documented
layouts and are exercised on synthetic checkpoints written with those names.
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Dict, Iterable, Optional

import torch

_VISION_PREFIXES = ("model.vision_tower.vision_tower.vision_model.", "vision_tower.vision_model.", "vision_model.")


def canonical_name(key: str) -> Optional[str]:
    """Checkpoint key -> runtime tensor name (None: not used by the inference path)."""
    k = key
    for pre in ("base_model.model.",):
        if k.startswith(pre):
            k = k[len(pre):]
    k = k.replace(".modules_to_save.default.", ".").replace(".base_layer.", ".")
    for pre in _VISION_PREFIXES:
        if k.startswith(pre):
            rest = k[len(pre):]
            if rest.startswith(("embeddings.", "encoder.layers.")):
                return "vision." + rest
            return None                                  # post_layernorm / pooling head: not on the llava path
    if k.startswith("model.mm_projector."):
        return k[len("model."):]
    if k.startswith(("mm_projector.", "informative_head.", "relevance_head.", "uncertainty_head.", "lm_head.")):
        return k
    if k.startswith("model.") and not k.startswith(("model.vision_tower", "model.image_newline")):
        return k                                         # model.embed_tokens / model.layers.N.* / model.norm
    return None


_LORA_RE = re.compile(r"^(?:base_model\.model\.)?(.*)\.lora_([AB])(?:\.default)?\.weight$")


def merge_lora(base: Dict[str, torch.Tensor], adapter: Dict[str, torch.Tensor], lora_alpha: float, r: Optional[int] = None,
               dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """W' = W + (alpha/r) B A, fp32 merge, one cast.  `adapter` holds `<module>.lora_A.weight` [r,in] and
    `<module>.lora_B.weight` [out,r] (PEFT layout); other adapter tensors override base tensors."""
    out = dict(base)
    pairs: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, t in adapter.items():
        m = _LORA_RE.match(k)
        if m:
            pairs.setdefault(m.group(1), {})[m.group(2)] = t
        else:
            name = canonical_name(k)
            if name is not None:
                out[name] = t.to(dtype)
    for module, ab in pairs.items():
        name = canonical_name(module + ".weight")
        if name is None or name not in out:
            raise KeyError(f"LoRA target {module} has no base weight")
        A, B = ab["A"].float(), ab["B"].float()
        rank = r or A.shape[0]
        out[name] = (out[name].float() + (lora_alpha / rank) * (B @ A)).to(dtype)
    return out


def _read_safetensors(paths: Iterable[str]) -> Dict[str, torch.Tensor]:
    from safetensors import safe_open
    tensors: Dict[str, torch.Tensor] = {}
    for p in paths:
        with safe_open(p, framework="pt", device="cpu") as f:
            for k in f.keys():
                tensors[k] = f.get_tensor(k)
    return tensors


def load_checkpoint(base_dir: str, lora_dir: Optional[str] = None, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """base_dir: directory with *.safetensors shards of the base model; lora_dir: PEFT adapter directory
    (adapter_model.safetensors + adapter_config.json with r / lora_alpha)."""
    shards = sorted(glob.glob(os.path.join(base_dir, "*.safetensors")))
    if not shards:
        raise FileNotFoundError(f"no *.safetensors under {base_dir}")
    base: Dict[str, torch.Tensor] = {}
    for k, t in _read_safetensors(shards).items():
        name = canonical_name(k)
        if name is not None:
            base[name] = t.to(dtype)
    if lora_dir:
        cfg = json.load(open(os.path.join(lora_dir, "adapter_config.json")))
        adapter = _read_safetensors(sorted(glob.glob(os.path.join(lora_dir, "*.safetensors"))))
        base = merge_lora(base, adapter, float(cfg["lora_alpha"]), int(cfg["r"]), dtype)
    return base


def write_predictions(path: str, results: list) -> None:
    """[{video_uuid, model_response_list, video_duration, true_frames_list, debug_data}, ...] with
    debug_data already passed through round_numbers(…, 3) (test/inference.py:697-711)."""
    need = {"video_uuid", "model_response_list", "video_duration", "true_frames_list", "debug_data"}
    for r in results:
        missing = need - set(r)
        if missing:
            raise ValueError(f"prediction record lacks {sorted(missing)}")
    with open(path, "w") as f:
        f.write(json.dumps(results, indent=4))
        f.flush()
