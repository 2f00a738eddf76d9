"""Shape/config records for the per-frame streaming path.

Field names follow the reference's config surface so a user of the reference finds the
same knobs (models/arguments_live.py:5-75, models/configuration_live.py:22-36,
models/live_llava/video_head_live_llava_qwen.py:43-47).  Presets carry the two shape
sets of SURVEY.md section 8(d): the benchmark shapes (ViT-L/14 @336 -> 36 tokens/frame)
and the reference-faithful shapes (so400m/14 @384 -> 49 tokens/frame).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, asdict


@dataclass(frozen=True)
class VisionConfig:
    image_size: int = 336
    patch_size: int = 14
    hidden_size: int = 1024
    num_hidden_layers: int = 24          # layers actually executed (llava drops the last one upstream)
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    layer_norm_eps: float = 1e-6
    kind: str = "siglip"                 # "siglip" (no CLS) or "clip" (CLS + pre-LN, models/vision_live.py:34-54)

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclass(frozen=True)
class LMConfig:
    hidden_size: int = 3584
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    head_dim: int = 128
    intermediate_size: int = 18944
    vocab_size: int = 152064
    rope_theta: float = 1e6
    rms_norm_eps: float = 1e-6
    max_position_embeddings: int = 32768


@dataclass(frozen=True)
class LiveConfig:
    """Everything the hot path needs. `video_pooling_stride`/`mm_spatial_pool_mode` are the
    post_projector_pooling knobs (video_head_live_llava_qwen.py:117-136)."""
    vision: VisionConfig = field(default_factory=VisionConfig)
    lm: LMConfig = field(default_factory=LMConfig)
    video_pooling_stride: int = 4
    mm_spatial_pool_mode: str = "bilinear"
    name: str = "bench"

    @property
    def pooled_grid(self) -> int:
        g, s = self.vision.grid, self.video_pooling_stride
        if self.mm_spatial_pool_mode == "bilinear":
            return math.ceil(g / s)
        return g // s                     # avg_pool2d / max_pool2d floor semantics

    @property
    def frame_num_tokens(self) -> int:
        return self.pooled_grid ** 2

    @property
    def frame_resolution(self) -> int:
        return self.vision.image_size

    def to_dict(self):
        return asdict(self)


def preset(name: str) -> LiveConfig:
    """Named shape sets.

    bench    : BASELINE.json configs[1..3]  SigLIP-L/14@336 + Qwen2-7B            (Tf=36)
    ref      : reference-faithful           so400m/14@384 (26 layers) + Qwen2-7B  (Tf=49)
    plumbing : BASELINE.json configs[0]     SigLIP-base dims + 2-layer LM         (Tf=36)
    tiny     : parity-test size the oracle finishes in well under a second
    """
    if name == "bench":
        return LiveConfig(name="bench")
    if name == "ref":
        return LiveConfig(
            vision=VisionConfig(image_size=384, patch_size=14, hidden_size=1152, num_hidden_layers=26,
                                num_attention_heads=16, intermediate_size=4304),
            name="ref")
    if name == "plumbing":
        return LiveConfig(
            vision=VisionConfig(image_size=336, patch_size=14, hidden_size=768, num_hidden_layers=12,
                                num_attention_heads=12, intermediate_size=3072),
            lm=LMConfig(hidden_size=512, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2,
                        head_dim=64, intermediate_size=1024, vocab_size=1024),
            name="plumbing")
    if name == "tiny":
        return LiveConfig(
            vision=VisionConfig(image_size=56, patch_size=14, hidden_size=128, num_hidden_layers=2,
                                num_attention_heads=2, intermediate_size=256),
            lm=LMConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                        head_dim=64, intermediate_size=512, vocab_size=512),
            video_pooling_stride=2, name="tiny")
    if name == "tiny128":                # exercises head_dim 128 / GQA 7:1 like Qwen2-7B, small otherwise
        return LiveConfig(
            vision=VisionConfig(image_size=84, patch_size=14, hidden_size=128, num_hidden_layers=2,
                                num_attention_heads=2, intermediate_size=256),
            lm=LMConfig(hidden_size=896, num_hidden_layers=2, num_attention_heads=7, num_key_value_heads=1,
                        head_dim=128, intermediate_size=1024, vocab_size=512),
            video_pooling_stride=2, name="tiny128")
    raise ValueError(f"unknown preset {name!r}")
