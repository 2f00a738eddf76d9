"""LiveLlava model API on top of the runtime: the call surface of
VideoHeadLiveLlavaQwenForCausalLM + LiveMixin that the drivers use
(models/live_llava/video_head_live_llava_qwen.py:138-154,317-330; models/modeling_live.py:31-61).

    model.visual_embed(frames_u8)                        -> bf16 [N*Tf, H]
    model.get_input_embeddings()(ids)                    -> bf16 [..., H]
    model(inputs_embeds=[B,T,H], past_key_values=cache, use_cache=True, return_dict=True, **ignored)
        -> VideoHeadCausalLMOutputWithPast(logits, past_key_values, informative_logits,
                                           relevance_logits, uncertainty, loss=0., ...)

`past_key_values` is one cache object (B == 1) or a list of B cache objects from aha_amd.cache
(SinkCache / SlidingWindowCache / TrulyStaticCache / DynamicCache); None creates a DynamicCache,
like transformers does.  `logits` is computed on first access; by default it covers the LAST position only
([B,1,V]) - the reference materialises lm_head over all T positions every frame (53 GFLOP + 30 MB that the
frame loop never reads); LiveLlavaModel(rt, all_position_logits=True) returns the
reference's full [B,T,V] (aha_lm_logits_all).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List, Optional, Sequence, Union

import torch

from .cache import DynamicCache, _BoundCache
from .runtime import Runtime


@dataclass
class VideoHeadCausalLMOutputWithPast:
    loss: Any = 0.0
    past_key_values: Any = None
    hidden_states: Any = None
    attentions: Any = None
    lm_loss: Any = None
    video_loss: Any = None
    informative_logits: Optional[torch.Tensor] = None     # fp32 [B,T,2]
    relevance_logits: Optional[torch.Tensor] = None       # fp32 [B,T,1], post-sigmoid (:187)
    uncertainty: Optional[torch.Tensor] = None            # fp32 [B,T,1], log-variance
    _rt: Any = None
    _B: int = 0
    _T: int = 0
    _all_positions: bool = False
    _logits: Optional[torch.Tensor] = None

    @property
    def logits(self) -> torch.Tensor:
        """fp32 [B,1,V] (last position; what the drivers read), or [B,T,V] like the reference's forward
        (video_head_live_llava_qwen.py:175) when the model was built with all_position_logits=True.  Computed on first access."""
        if self._logits is None:
            if self._all_positions:
                self._logits = self._rt.logits_all(self._B, self._T)
            else:
                lg, _ = self._rt.logits_last(self._B)
                self._logits = lg.view(self._B, 1, -1)
        return self._logits


class LiveLlavaModel:
    def __init__(self, runtime: Runtime, all_position_logits: bool = False):
        self.rt = runtime
        self.all_position_logits = all_position_logits
        self.config = runtime.cfg
        self.device = runtime.device

    def eval(self):
        return self

    def visual_embed(self, frames: torch.Tensor) -> torch.Tensor:
        return self.rt.visual_embed(frames)

    def get_input_embeddings(self):
        rt = self.rt
        return lambda ids: rt.embed_tokens(ids).view(*ids.shape, rt.hidden_size)

    def joint_embed(self, input_ids=None, frames=None):
        if frames is None:
            return self.get_input_embeddings()(input_ids)
        if input_ids is None:
            return self.visual_embed(frames)
        raise NotImplementedError("placeholder-scatter joint_embed is the training path (modeling_live.py:39-61)")

    def _bind(self, c) -> _BoundCache:
        if c is None:
            c = DynamicCache()
        if not isinstance(c, _BoundCache):
            raise TypeError("past_key_values must come from aha_amd.cache")
        c.bind(self.rt)
        return c

    def __call__(self, inputs_embeds: torch.Tensor = None, past_key_values: Union[None, _BoundCache, Sequence] = None,
                 use_cache: bool = True, return_dict: bool = True, **ignored) -> VideoHeadCausalLMOutputWithPast:
        assert inputs_embeds is not None and inputs_embeds.dim() == 3
        B, T, _ = inputs_embeds.shape
        if isinstance(past_key_values, (list, tuple)):
            caches = [self._bind(c) for c in past_key_values]
        else:
            caches = [self._bind(past_key_values)]
        assert len(caches) == B, "one cache object per stream"
        self.rt.lm_step([c.stream for c in caches], inputs_embeds.to(torch.bfloat16))
        raw = self.rt.heads_all(B, T)                               # [B,T,4] bf16-rounded head logits
        return VideoHeadCausalLMOutputWithPast(
            past_key_values=caches[0] if not isinstance(past_key_values, (list, tuple)) else caches,
            informative_logits=raw[..., 0:2].contiguous(), relevance_logits=torch.sigmoid(raw[..., 2:3]),
            uncertainty=raw[..., 3:4].contiguous(), _rt=self.rt, _B=B, _T=T, _all_positions=self.all_position_logits)

    forward = __call__
