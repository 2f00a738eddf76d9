"""Stand-in for the reference's chat-template tokenizer (models/tokenization_live.py:34-134).

Tokenisation is outside the hot path: the driver only needs id tensors
for the system prompt, the 'stream' role markers and user queries.  With no network there is no
Qwen2 vocabulary here, so this class reproduces the template STRUCTURE of the reference
(``<|im_start|>role\\ncontent<|im_end|>`` turns, the stream / stream-generation prompts, the
optional leading ``<|im_end|>`` that closes a stream turn) over a deterministic crc32 word hash.
Any object with the same two methods (a real HF tokenizer carrying the reference's template) can
be passed to the drivers instead.
"""
from __future__ import annotations

import re
import zlib
from typing import List

import torch


class SyntheticChatTokenizer:
    def __init__(self, vocab_size: int, n_special: int = 8):
        assert vocab_size > n_special + 8
        self.vocab_size, self.n_special = vocab_size, n_special
        self.im_start, self.im_end, self.nl = 0, 1, 2
        self.eos_token_id = self.im_end
        self.roles = {"system": 3, "user": 4, "assistant": 5, "stream": 6}

    def _words(self, text: str) -> List[int]:
        span = self.vocab_size - self.n_special
        return [self.n_special + zlib.crc32(w.encode()) % span for w in re.findall(r"\w+|[^\w\s]", text)]

    def apply_chat_template(self, conversation, add_stream_query_prompt=False, add_stream_prompt=False,
                            add_stream_generation_prompt=False, return_tensors="pt", **_):
        ids: List[int] = []
        if add_stream_query_prompt:            # a query arriving mid-stream first closes the stream turn
            ids += [self.im_end]
        for turn in conversation:
            if not turn:
                continue
            if ids:
                ids += [self.nl]
            ids += [self.im_start, self.roles[turn["role"]], self.nl] + self._words(turn["content"]) + [self.im_end]
        if add_stream_prompt:                  # "\n<|im_start|>stream\n": 4 tokens like the reference's
            ids += [self.nl, self.im_start, self.roles["stream"], self.nl]
        if add_stream_generation_prompt:       # "<|im_end|>\n<|im_start|>assistant\n"
            ids += [self.im_end, self.nl, self.im_start, self.roles["assistant"], self.nl]
        return torch.tensor([ids], dtype=torch.long)

    def decode(self, ids, skip_special_tokens=True, **_):
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        return " ".join(f"<{i}>" for i in ids if not (skip_special_tokens and i < self.n_special))
