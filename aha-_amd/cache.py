"""Cache objects with the reference's class names and constructor arguments
(test/sink_cache.py:8-19, test/sliding_window_cache.py:5-15, test/static_cache.py:5-16).  Each is a
handle on a preallocated ring KV buffer inside the runtime (aha_stream); `update()` is not exposed
because the cache update happens inside the fused LM step (qkv_finish writes K/V straight into
the ring, sink_rerotate re-rotates kept keys in place)."""
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
