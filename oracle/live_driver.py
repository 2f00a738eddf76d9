"""ORACLE (test infrastructure, never shipped or measured as the product).

Restatement of the reference's stream driver on top of the oracle model:

  LiveInferForBenchmark            test/inference.py:38-348
    reset :112-130, _init_cache :133-155, input_video_stream :169-187,
    input_query_stream :189-192, _encode_frame :196-229, _encode_query :231-262,
    _generate_response :264-281, inference :283-335
  fast_greedy_generate             models/modeling_live.py:64-90
  round_numbers / truncate_sig     test/inference.py:359-375

Tokenisation is outside the hot path: prompts arrive as id
tensors.
"""
from __future__ import annotations

import collections
from typing import Optional

import torch

from .cache_policies import make_policy
from .qwen2_live import OracleLM, frame_scores
from .vision_tower import OracleVision


def truncate_sig(x, sig=3):
    if x == 0:
        return 0
    return float(f"{x:.{sig}g}")


def round_numbers(data, n):
    if isinstance(data, list):
        return [round_numbers(d, n) for d in data]
    if isinstance(data, dict):
        return {k: round_numbers(v, n) for k, v in data.items()}
    if isinstance(data, float):
        if abs(data) <= 10 ** (-n):
            return truncate_sig(data, n)
        return round(data, n)
    return data


class OracleLiveInfer:
    def __init__(self, cfg, weights, *, dtype=torch.bfloat16, alt_cache="default_sink",
                 window_length=2048, num_sink_tokens=32, frame_fps=1.0,
                 start_ids=None, stream_prompt_ids=None, stream_generation_ids=None,
                 score_heads="relevance_score,informative_score", stream_end_prob_threshold=None,
                 stream_end_score_sum_threshold=None, running_list_length=20,
                 remove_assistant_turns=False, eos_token_id=0, max_new_tokens=200,
                 repetition_penalty=None, attn_semantics="trailing"):
        self.cfg = cfg
        self.dtype = dtype
        self.lm = OracleLM(cfg.lm, weights, dtype, attn_semantics=attn_semantics)
        self.vision = OracleVision(cfg, weights, dtype)
        self.alt_cache, self.window_length, self.num_sink_tokens = alt_cache, window_length, num_sink_tokens
        self.frame_num_tokens = cfg.frame_num_tokens
        self.hidden_size = cfg.lm.hidden_size
        self._start_ids = start_ids
        self._added_stream_prompt_ids = stream_prompt_ids
        self._added_stream_generation_ids = stream_generation_ids
        self.score_heads = score_heads.split(",")
        self.stream_end_prob_threshold = stream_end_prob_threshold
        self.stream_end_score_sum_threshold = stream_end_score_sum_threshold
        self.running_list_length = running_list_length
        self.remove_assistant_turns = remove_assistant_turns
        self.eos_token_id = eos_token_id
        self.max_new_tokens = max_new_tokens
        self.repetition_penalty = repetition_penalty
        self.set_fps(frame_fps)
        self.reset()

    def set_fps(self, fps=None, frame_interval=None):
        assert (fps is None) != (frame_interval is None)
        if fps is not None:
            self.frame_fps, self.frame_interval = fps, 1 / fps
        else:
            self.frame_interval, self.frame_fps = frame_interval, 1 / frame_interval

    def reset(self):
        self.query_queue = collections.deque()
        self.frame_embeds_queue = collections.deque()
        self.video_time = 0
        self.frame_idx = 0
        self.last_role = "system"
        self.last_ids = torch.zeros((1, 0), dtype=torch.long)
        self.past_key_values = make_policy(self.alt_cache, self.window_length, self.num_sink_tokens)
        self.debug_data_list = []
        self.generated_token_ids = []
        self.init_vision_time = False
        self.stream_end_prob_list = []
        self.stream_end_score_sum = 0

    def input_video_stream(self, frames_u8):
        bs = 32
        for b in range(0, len(frames_u8), bs):
            emb = self.vision.visual_embed(frames_u8[b:b + bs]).split(self.frame_num_tokens)
            self.frame_embeds_queue.extend([((r + b) / self.frame_fps, f) for r, f in enumerate(emb)])

    def input_query_stream(self, conversation):
        """turns: {'role':'user','time':t,'ids': LongTensor[1,n]} (content already tokenised)."""
        for turn in conversation:
            if turn["role"] == "user":
                self.query_queue.append((turn["time"], turn["ids"]))

    def _encode_frame(self):
        if not self.frame_embeds_queue:
            return None, None
        _, frame_embeds = self.frame_embeds_queue.popleft()
        if not self.init_vision_time:
            self.last_ids = self._start_ids
            self.init_vision_time = True
        elif self.last_role == "assistant" and not self.remove_assistant_turns:
            self.last_ids = torch.cat([self.last_ids, self._added_stream_prompt_ids], dim=1)
        else:
            self.last_ids = torch.zeros((1, 0), dtype=torch.long)
        inputs_embeds = torch.cat([
            self.lm.embed_tokens(self.last_ids).view(1, -1, self.hidden_size),
            frame_embeds.view(1, -1, self.hidden_size)], dim=1)
        out = self.lm.step(inputs_embeds, self.past_key_values)
        self.frame_idx += 1
        s = frame_scores(out)[0]
        self.last_role = "stream"
        return {"informative_score": s[0].item(), "relevance_score": s[1].item()}, s[2].item()

    def _encode_query(self):
        _, query_ids = self.query_queue.popleft()
        self.last_ids = query_ids
        out = self.lm.step(self.lm.embed_tokens(query_ids), self.past_key_values, want_logits=True)
        self.last_ids = out["logits"][:, -1:].argmax(dim=-1)
        self.last_role = "user"

    def _generate_response(self):
        self.last_ids = self._added_stream_generation_ids
        inputs_embeds = self.lm.embed_tokens(self.last_ids)
        output_ids = []
        for _ in range(self.max_new_tokens):
            out = self.lm.step(inputs_embeds, self.past_key_values, want_logits=True)
            logits = out["logits"][:, -1, :]
            if self.repetition_penalty is not None and self.generated_token_ids:
                idx = torch.tensor(self.generated_token_ids)[None]
                sc = torch.gather(logits, 1, idx)
                sc = torch.where(sc < 0, sc * self.repetition_penalty, sc / self.repetition_penalty)
                logits = logits.scatter(1, idx, sc)
            tok = logits.argmax(dim=-1, keepdim=True)
            if self.repetition_penalty is not None and tok.item() != self.eos_token_id:
                self.generated_token_ids.append(tok.item())
            output_ids.append(tok.item())
            if tok.item() == self.eos_token_id:
                break
            inputs_embeds = self.lm.embed_tokens(tok)
        if not self.remove_assistant_turns:
            self.last_ids = torch.tensor([[output_ids[-1]]])
        else:
            self.last_ids = torch.zeros((1, 0), dtype=torch.long)
        self.last_role = "assistant"
        return output_ids

    def inference(self):
        responses = []
        while self.frame_embeds_queue:
            if self.query_queue and self.video_time >= self.query_queue[0][0]:
                self._encode_query()
            video_scores, unc = self._encode_frame()
            self.debug_data_list.append(dict(time=self.video_time, **video_scores, uncertainty_score=unc))
            need_response = False
            s = sum(v for k, v in video_scores.items() if k in self.score_heads)
            self.stream_end_prob_list.append(s)
            self.stream_end_score_sum += s
            if isinstance(self.running_list_length, int) and self.running_list_length > 0:
                self.stream_end_prob_list = self.stream_end_prob_list[-self.running_list_length:]
            if self.stream_end_score_sum_threshold is not None and self.stream_end_score_sum > self.stream_end_score_sum_threshold:
                need_response = True
                self.stream_end_score_sum = 0
            if self.stream_end_prob_threshold is not None and s > self.stream_end_prob_threshold:
                need_response = True
            if need_response:
                responses.append({"time": self.video_time, "content": self._generate_response(), "role": "assistant"})
            self.video_time += 1 / self.frame_fps
        return responses
