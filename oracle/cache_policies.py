"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the reference's KV-cache policies.  Each class follows the reference
method it names line by line in behaviour, without the HF ``Cache`` base class (whose
constructor changed in transformers 5.x):

  SinkPolicy     <- test/sink_cache.py:8-164        (SinkCache)
  SlidingPolicy  <- test/sliding_window_cache.py:5-52 (SlidingWindowCache)
  StaticPolicy   <- test/static_cache.py:5-46       (TrulyStaticCache)
  GrowingPolicy  <- transformers DynamicCache (what ``past_key_values=None`` becomes,
                    test/inference.py:154-155)

Interface = the operator API of SURVEY.md 8(b): ``update(k, v, layer_idx, cache_kwargs)``
returns the (K, V) the attention sees; ``get_seq_length()`` feeds position ids.

Pinned by tests/test_oracle_cache.py against the imported reference classes (when
/root/reference is present) and by tests/golden/cache_policies.npz.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    # test/sink_cache.py:21-25
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class _Base:
    name = "base"

    def __init__(self) -> None:
        self.key_cache: List[torch.Tensor] = []
        self.value_cache: List[torch.Tensor] = []
        self._seen_tokens = 0

    def get_seq_length(self, layer_idx: int = 0) -> int:
        if len(self.key_cache) <= layer_idx:
            return 0
        return self.key_cache[layer_idx].shape[-2]


class GrowingPolicy(_Base):
    """Unbounded append (DynamicCache semantics)."""
    name = "none"

    def update(self, k, v, layer_idx, cache_kwargs=None):
        if len(self.key_cache) <= layer_idx:
            self.key_cache.append(k)
            self.value_cache.append(v)
        else:
            self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], k], dim=-2)
            self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], v], dim=-2)
        return self.key_cache[layer_idx], self.value_cache[layer_idx]


class SlidingPolicy(_Base):
    """test/sliding_window_cache.py:17-44: cat, keep last W, no re-rotation."""
    name = "sliding_window"

    def __init__(self, window_length: int) -> None:
        super().__init__()
        self.window_length = window_length

    def update(self, k, v, layer_idx, cache_kwargs=None):
        if len(self.key_cache) <= layer_idx:            # :28-31 first call stores as is (even if > W)
            self.key_cache.append(k)
            self.value_cache.append(v)
            return k, v
        fk = torch.cat([self.key_cache[layer_idx], k], dim=-2)
        fv = torch.cat([self.value_cache[layer_idx], v], dim=-2)
        self.key_cache[layer_idx] = fk[:, :, -self.window_length:]
        self.value_cache[layer_idx] = fv[:, :, -self.window_length:]
        return self.key_cache[layer_idx], self.value_cache[layer_idx]


class StaticPolicy(_Base):
    """test/static_cache.py:18-36: freeze the first call's [:W]; later calls return the
    frozen prefix only (the new tokens' K/V are NOT part of what attention sees)."""
    name = "static"

    def __init__(self, window_size: int) -> None:
        super().__init__()
        self.window_size = window_size

    def update(self, k, v, layer_idx, cache_kwargs=None):
        if len(self.key_cache) <= layer_idx:
            self.key_cache.append(k[:, :, : self.window_size])
            self.value_cache.append(v[:, :, : self.window_size])
        return self.key_cache[layer_idx], self.value_cache[layer_idx]


class SinkPolicy(_Base):
    """test/sink_cache.py:74-164 with helpers :21-55."""
    name = "default_sink"

    def __init__(self, window_length: int, num_sink_tokens: int) -> None:
        super().__init__()
        self.window_length = window_length
        self.num_sink_tokens = num_sink_tokens
        self.cos_sin_rerotation_cache: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._cos_cache: Optional[torch.Tensor] = None
        self._sin_cache: Optional[torch.Tensor] = None

    def _rerotation(self, T: int, dtype, cos: torch.Tensor, sin: torch.Tensor):
        # :35-55.  cos/sin are rows [0, W) of the accumulated table, in the model dtype.
        if T not in self.cos_sin_rerotation_cache:
            cos = cos.to(torch.float32)
            sin = sin.to(torch.float32)
            s = self.num_sink_tokens
            original_cos = cos[s + T:]
            shifted_cos = cos[s:-T]
            original_sin = sin[s + T:]
            shifted_sin = sin[s:-T]
            rc = original_cos * shifted_cos + original_sin * shifted_sin
            rs = -original_sin * shifted_cos + original_cos * shifted_sin
            self.cos_sin_rerotation_cache[T] = (rc.to(dtype).unsqueeze(0), rs.to(dtype).unsqueeze(0))
        return self.cos_sin_rerotation_cache[T]

    def update(self, k, v, layer_idx, cache_kwargs=None):
        cache_kwargs = cache_kwargs or {}
        sin = cache_kwargs.get("sin")
        cos = cache_kwargs.get("cos")
        using_rope = cos is not None and sin is not None
        T = k.shape[-2]
        if layer_idx == 0:
            self._seen_tokens += T
        if using_rope and layer_idx == 0:               # :109-121 (3-dim cos branch)
            if self._cos_cache is None:
                self._cos_cache = cos[0, ...]
                self._sin_cache = sin[0, ...]
            elif self._cos_cache.shape[0] < self.window_length:
                self._cos_cache = torch.cat([self._cos_cache, cos[0, ...]], dim=0)
                self._sin_cache = torch.cat([self._sin_cache, sin[0, ...]], dim=0)

        W, s = self.window_length, self.num_sink_tokens
        if len(self.key_cache) <= layer_idx:            # :124-127 empty
            self.key_cache.append(k)
            self.value_cache.append(v)
        elif T + self.get_seq_length(layer_idx) < W:    # :129-132 growing
            self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], k], dim=-2)
            self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], v], dim=-2)
        else:                                           # :134-162 shifting
            keys_to_keep = self.key_cache[layer_idx][:, :, -W + s + T:]
            if using_rope:
                rc, rs = self._rerotation(T, k.dtype, self._cos_cache[:W], self._sin_cache[:W])
                keys_to_keep = (keys_to_keep * rc) + (rotate_half(keys_to_keep) * rs)   # :27-33
            sink_keys = self.key_cache[layer_idx][:, :, :s]
            self.key_cache[layer_idx] = torch.cat([sink_keys, keys_to_keep, k], dim=-2)
            sink_values = self.value_cache[layer_idx][:, :, :s]
            values_to_keep = self.value_cache[layer_idx][:, :, -W + s + T:]
            self.value_cache[layer_idx] = torch.cat([sink_values, values_to_keep, v], dim=-2)
        return self.key_cache[layer_idx], self.value_cache[layer_idx]


def make_policy(alt_cache: Optional[str], window_length: int = 2048, num_sink_tokens: int = 32):
    """Mirror of LiveInferForBenchmark._init_cache's selection (test/inference.py:133-155)."""
    if alt_cache == "default_sink":
        return SinkPolicy(window_length, num_sink_tokens)
    if alt_cache == "sliding_window":
        return SlidingPolicy(window_length)
    if alt_cache == "static":
        return StaticPolicy(window_length)
    if alt_cache in (None, "none"):
        return GrowingPolicy()
    raise ValueError(f"unknown alt_cache {alt_cache!r}")
