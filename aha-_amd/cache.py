"""Cache objects with the reference's class names, constructor arguments and operator interface
(test/sink_cache.py:8-19,57-80, test/sliding_window_cache.py:5-44, test/static_cache.py:5-36).  Each is a
handle on a preallocated ring KV buffer inside the runtime (aha_stream).  The fused LM step updates
the cache itself (qkv_finish writes K/V straight into the ring, sink_rerotate re-rotates kept keys in
place); `update(key_states, value_states, layer_idx, cache_kwargs)` is the same operation exposed at the
reference's operator level (aha_cache_update), for callers that drive the cache from their own
attention code the way transformers does."""
from __future__ import annotations

from typing import Optional


class _BoundCache:
    _alt = None
    window_length = 2048
    num_sink_tokens = 0

    def __init__(self):
        self.stream = None
        self.attn_semantics = "trailing"

    def bind(self, rt):
        if self.stream is None:
            self.stream = rt.open_stream(self._alt, self.window_length, self.num_sink_tokens,
                                         attn_semantics=self.attn_semantics)
        return self

    def get_seq_length(self, layer_idx: Optional[int] = 0) -> int:
        return 0 if self.stream is None else self.stream.get_seq_length()

    def update(self, key_states, value_states, layer_idx: int, cache_kwargs=None):
        """Cache.update of the reference: append this layer's new (already rotated) keys / values, bf16 [1, kv_heads, T,
        head_dim], evicting / re-rotating as the policy says, and return the (K, V) the attention must see, [1, kv_heads, L,
        head_dim].  As in the reference the layer_idx == 0 call advances the bookkeeping (seen tokens, eviction) and the other
        layers of the step follow in order.

        `cache_kwargs` (test/sink_cache.py:100-103): the re-rotation coefficients come from the runtime's own RoPE table (bit-identical
        to what SinkCache accumulates from the "cos" / "sin" it is handed, test/sink_cache.py:35-55,109-121), so the tensors are only
        CHECKED, on the layer-0 call, against that table at the positions get_seq_length() + arange(T) - the only positions the
        reference's callers pass.  Where the reference would behave differently from this ring the call is refused, loudly, before
        anything changes: SinkCache without cos / sin (the reference then shifts the kept keys WITHOUT re-rotating them), a
        `partial_rotation_size`, or cos / sin of other positions.  SlidingWindowCache / TrulyStaticCache ignore the kwargs, as the
        reference's do."""
        if self.stream is None:
            raise RuntimeError("cache is not bound to a runtime: call cache.bind(runtime) (or hand it to LiveLlavaModel) first")
        if key_states.dim() == 4 and key_states.shape[0] != 1:
            raise ValueError("one stream per cache object: batch dimension must be 1")
        self._check_kwargs(key_states, int(layer_idx), cache_kwargs)
        return self.stream.rt.cache_update(self.stream, int(layer_idx), key_states, value_states)

    def _check_kwargs(self, key_states, layer_idx: int, cache_kwargs) -> None:
        pass

    def get_max_length(self) -> Optional[int]:
        return self.window_length

    def get_max_cache_shape(self) -> Optional[int]:
        return self.window_length

    @property
    def _seen_tokens(self) -> int:
        return 0 if self.stream is None else self.stream.seen_tokens

    def reset(self):
        if self.stream is not None:
            self.stream.reset()


class SinkCache(_BoundCache):
    _alt = "default_sink"
    is_sliding = True

    def __init__(self, window_length: int, num_sink_tokens: int) -> None:
        super().__init__()
        self.window_length, self.num_sink_tokens = window_length, num_sink_tokens

    def _check_kwargs(self, key_states, layer_idx: int, cache_kwargs) -> None:
        kw = cache_kwargs or {}
        if kw.get("partial_rotation_size") is not None:
            raise NotImplementedError("SinkCache.update: partial_rotation_size is not supported (Qwen2 rotates the whole head_dim)")
        cos, sin = kw.get("cos"), kw.get("sin")
        if cos is None or sin is None:
            raise ValueError("SinkCache.update needs cache_kwargs['cos'] and ['sin'] (test/sink_cache.py:100-103): without them the "
                             "reference shifts the kept keys without re-rotating them, which this ring does not do")
        if layer_idx == 0:
            self.stream.rt.check_rope_rows(cos, sin, self.stream.get_seq_length(), key_states.shape[-2])


class SlidingWindowCache(_BoundCache):
    _alt = "sliding_window"

    def __init__(self, window_length: int) -> None:
        super().__init__()
        self.window_length = window_length


class TrulyStaticCache(_BoundCache):
    _alt = "static"

    def __init__(self, window_size: int) -> None:
        super().__init__()
        self.window_size = self.window_length = window_size


class DynamicCache(_BoundCache):
    """What `past_key_values=None` turns into (test/inference.py:154-155)."""
    _alt = None

    def get_max_length(self):
        return None

    def get_max_cache_shape(self):
        return None
