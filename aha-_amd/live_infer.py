"""Stream drivers with the reference's API surface, on top of the HIP runtime.

  LiveInferForBenchmark  <- test/inference.py:38-348
  LiveInferForDemo       <- test/live_infer_for_video.py:80-228
  round_numbers          <- test/inference.py:359-375 (same results, own formulation)

Same method names, argument meaning, attributes and `debug_data_list` schema
(`{time, informative_score, relevance_score, uncertainty_score}`), so code written against the
reference's drivers runs against these.  Differences, all on purpose:
  * the model is an `aha_amd.runtime.Runtime` (weights already on the GPU) instead of
    `build_model_and_tokenizer` (needs network); pass `runtime=` or `cfg=`+`weights=`;
  * frames enter as uint8 and the preprocess is fused into the patch gather on the GPU;
  * frame embeddings stay in HBM (the reference bounces every embedding through host memory,
    test/inference.py:185,213) and each frame costs ONE 12-byte D2H copy instead of three `.item()`s;
  * `inference()` returns the response list (the reference's `return` is commented out, :348).
"""
from __future__ import annotations

import collections
import math
from dataclasses import asdict
from typing import List, Optional

import numpy as np
import torch

from .arguments import LiveTestArguments
from .config import LiveConfig
from .runtime import Runtime, Stream
from .tokenization import SyntheticChatTokenizer


def round_numbers(data, n):
    """Rounding of the debug rows (same results as the reference helper, test/inference.py:359-375): floats are rounded
    to n decimals, except magnitudes <= 10**-n, which keep n significant digits instead of collapsing to 0 (an exact
    zero becomes the int 0); lists and dicts are walked, everything else passes through."""
    if isinstance(data, dict):
        return {key: round_numbers(val, n) for key, val in data.items()}
    if isinstance(data, list):
        return [round_numbers(item, n) for item in data]
    if not isinstance(data, float):
        return data
    if abs(data) > 10.0 ** (-n):
        return round(data, n)
    return float(format(data, f".{n}g")) if data else 0


def truncate_sig(x, sig=3):
    """`sig` significant digits of x (kept for callers of the reference's helper name)."""
    return float(format(x, f".{sig}g")) if x else 0


# per-video driver state, re-created by reset() (attribute names are the reference's: test/inference.py:112-130)
_STREAM_STATE = dict(video_time=0, frame_idx=0, last_role="system", video_tensor=None, first_query_processed=False,
                     init_vision_time=False, num_frames_no_reply=0, stream_end_score_sum=0, consecutive_n_frames=0)


class LiveInferForBenchmark:
    def __init__(self, args: LiveTestArguments, peft_model_id=None, sink_cache=False, alt_cache="default_sink", *,
                 runtime: Optional[Runtime] = None, cfg: Optional[LiveConfig] = None, weights=None, tokenizer=None,
                 device: str = "cuda:0", window_length: int = 2048, num_sink_tokens: int = 32,
                 attn_semantics: str = "trailing") -> None:
        assert not (args.bf16 and args.fp16), "only one of --bf16 true and --fp16 true can be set"
        self.sink_cache = sink_cache
        self.alt_cache = alt_cache
        self.peft_model_id = "aha_weights" if not peft_model_id else peft_model_id
        self.torch_dtype = torch.bfloat16
        if runtime is None:
            if cfg is None or weights is None:
                raise ValueError("pass runtime= or cfg= and weights= (there is no checkpoint download offline)")
            runtime = Runtime(cfg, weights, device=device)
        self.model = runtime                       # plays the role of self.model in the reference
        self.rt = runtime
        self.cfg = runtime.cfg
        self.device = runtime.device
        self.tokenizer = tokenizer or SyntheticChatTokenizer(self.cfg.lm.vocab_size)
        self._window_length, self._num_sink_tokens, self._attn_semantics = window_length, num_sink_tokens, attn_semantics

        # the reference's driver attributes, by name (test/inference.py:39-99): geometry from the model config, the trigger and
        # waiting knobs straight from `args`
        self.hidden_size = self.cfg.lm.hidden_size
        self.frame_resolution, self.frame_num_tokens = self.cfg.frame_resolution, self.cfg.frame_num_tokens
        if args.frame_fps > 0:
            self.set_fps(args.frame_fps)
        for knob in ("uncertainty_wait_threshold", "max_wait_frames", "system_prompt", "stream_end_prob_threshold",
                     "response_min_interval_frames", "threshold_z", "first_n_frames_no_generate", "running_list_length",
                     "stream_end_score_sum_threshold", "consecutive_n_frames_threshold"):
            setattr(self, knob, getattr(args, knob))
        self.score_heads = args.score_heads.split(",")
        self.max_new_tokens = 200                                  # the reference's output buffer is [1,200] (test/inference.py:73)
        triggers = dict(stream_end_prob_threshold=self.stream_end_prob_threshold, threshold_z=self.threshold_z,
                        stream_end_score_sum_threshold=self.stream_end_score_sum_threshold)
        if sum(v is not None for v in triggers.values()) != 1:
            raise ValueError("only one of --stream_end_prob_threshold, --threshold_z and --stream_end_score_sum_threshold can be set. "
                             f"However, they are: {triggers['stream_end_prob_threshold']}, {triggers['threshold_z']}, "
                             f"{triggers['stream_end_score_sum_threshold']}")
        if self.threshold_z is not None and self.first_n_frames_no_generate is None:
            raise ValueError("--first_n_frames_no_generate must be set when --threshold_z is set")
        self.remove_assistant_turns = args.remove_assistant_turns
        self.eos_token_id = getattr(self.tokenizer, "eos_token_id", 0)
        self._start_ids = self.tokenizer.apply_chat_template(
            [{"role": "system", "content": self.system_prompt}], return_tensors="pt").to(self.device)
        self._added_stream_prompt_ids = self.tokenizer.apply_chat_template(
            [{}], add_stream_prompt=True, return_tensors="pt").to(self.device)
        self._added_stream_generation_ids = self.tokenizer.apply_chat_template(
            [{}], add_stream_generation_prompt=True, return_tensors="pt").to(self.device)
        self.repetition_penalty = args.repetition_penalty
        self.past_key_values: Optional[Stream] = None
        self.reset()

    # ---- test/inference.py:101-110 ---------------------------------------------------------------
    def set_fps(self, fps=None, frame_interval=None):
        """Exactly one of `fps` / `frame_interval` (seconds); the other is derived."""
        assert (fps is None) != (frame_interval is None)
        self.frame_fps = fps if fps is not None else 1 / frame_interval
        self.frame_interval = frame_interval if frame_interval is not None else 1 / fps

    # ---- test/inference.py:112-130 ---------------------------------------------------------------
    def reset(self):
        for name, value in _STREAM_STATE.items():
            setattr(self, name, value)
        self.query_queue, self.frame_embeds_queue = collections.deque(), collections.deque()
        self.debug_data_list, self.generated_token_ids, self.stream_end_prob_list = [], [], []
        self.last_ids = self._no_ids()
        self._init_cache(self._window_length, self._num_sink_tokens)

    def _no_ids(self):
        return torch.zeros((1, 0), device=self.device, dtype=torch.long)

    # ---- test/inference.py:133-155 ---------------------------------------------------------------
    def _init_cache(self, window_length=2048, num_sink_tokens=32, instruction_ids=None):
        if instruction_ids is None:
            instruction_ids = self._start_ids
        num_instruction_tokens = instruction_ids.shape[1]
        local_window_length = window_length + num_sink_tokens - num_instruction_tokens
        if self.sink_cache:
            spec = ("default_sink", local_window_length, num_instruction_tokens)
        elif self.alt_cache and self.alt_cache == "default_sink":
            spec = ("default_sink", window_length, num_sink_tokens)
        elif self.alt_cache and self.alt_cache == "sliding_window":
            spec = ("sliding_window", window_length, 0)
        elif self.alt_cache and self.alt_cache == "static":
            spec = ("static", window_length, 0)
        else:
            spec = (None, window_length, 0)
        old = self.past_key_values
        if old is not None and getattr(old, "_spec", None) == spec and old.handle is not None:
            old.reset()                              # same policy: reuse the preallocated KV buffers
            return
        if old is not None:
            old.close()
        self.past_key_values = self.rt.open_stream(spec[0], spec[1], spec[2], attn_semantics=self._attn_semantics)
        self.past_key_values._spec = spec

    # ---- test/inference.py:169-187 ---------------------------------------------------------------
    @torch.no_grad()
    def input_video_stream(self, video_frames: torch.Tensor):
        """video_frames: uint8 [T,3,S,S] RGB (what load_video_for_testing returns)."""
        assert video_frames.dtype == torch.uint8
        video_frames = video_frames.to(self.device)
        batch_size = 32
        for batch_i in range(0, math.ceil(len(video_frames) / batch_size)):
            batch = video_frames[batch_i * batch_size: batch_i * batch_size + batch_size]
            frame_embeds = self.rt.visual_embed(batch).split(self.frame_num_tokens)
            self.frame_embeds_queue.extend(
                [((r + batch_i * batch_size) / self.frame_fps, f) for r, f in enumerate(frame_embeds)])

    # ---- test/inference.py:189-192 ---------------------------------------------------------------
    def input_query_stream(self, conversation):
        for turn in conversation:
            if turn["role"] == "user":
                self.query_queue.append((turn["time"], turn["content"]))

    # ---- test/inference.py:196-229 ---------------------------------------------------------------
    def _encode_frame(self):
        """returns: ({informative_score, relevance_score}, uncertainty_score)"""
        if not self.frame_embeds_queue:
            return None, None
        video_time, frame_embeds = self.frame_embeds_queue.popleft()
        if not self.init_vision_time:
            self.last_ids = self._start_ids
            self.init_vision_time = True
        elif self.last_role == "assistant" and not self.remove_assistant_turns:
            self.last_ids = torch.cat([self.last_ids, self._added_stream_prompt_ids], dim=1)
        else:
            self.last_ids = self._no_ids()
        inputs_embeds = torch.cat([
            self.rt.embed_tokens(self.last_ids).view(1, -1, self.hidden_size),
            frame_embeds.view(1, -1, self.hidden_size)], dim=1)
        scores = self.rt.lm_step([self.past_key_values], inputs_embeds)
        self.frame_idx += 1
        self.num_frames_no_reply += 1
        s = scores[0].tolist()                       # one 12-byte D2H copy per frame
        self.last_role = "stream"
        return {"informative_score": s[0], "relevance_score": s[1]}, s[2]

    # ---- test/inference.py:231-262 ---------------------------------------------------------------
    def _encode_query(self):
        query_time, query = self.query_queue.popleft()
        query_ids = self.tokenizer.apply_chat_template(
            [{"role": "user", "content": query}], add_stream_query_prompt=self.last_role == "stream",
            add_stream_prompt=True, return_tensors="pt").to(self.device)
        self.last_ids = query_ids
        inputs_embeds = self.rt.embed_tokens(self.last_ids).view(1, -1, self.hidden_size)
        self.rt.lm_step([self.past_key_values], inputs_embeds)
        _, am = self.rt.logits_last(1, want_logits=False)
        self.last_ids = am.view(1, 1)
        self.last_role = "user"

    # ---- models/modeling_live.py:64-90 (fast_greedy_generate) + test/inference.py:264-281 ----------
    generation_chunk: Optional[int] = None        # tokens per aha_generate_greedy call (None: the whole response in one call)
    between_chunks = None                         # callable run between chunks, e.g. to step other streams of a server

    def _generate_response(self):
        """fast_greedy_generate runs inside the runtime (aha_generate_greedy): argmax -> embedding -> next single-token step
        stay on the device, the host only reads back each 8-byte token id to stop at EOS.  With `generation_chunk` set the
        response is produced in chunks of that many tokens (same ids), `between_chunks()` running in between."""
        self.last_ids = self._added_stream_generation_ids
        chunked = dict(chunk=self.generation_chunk, between_chunks=self.between_chunks) if self.generation_chunk else {}
        output_ids = self.rt.generate_greedy(self.past_key_values, self.last_ids, self.max_new_tokens, self.eos_token_id,
                                             self.repetition_penalty, self.generated_token_ids, **chunked)
        self.last_ids = self._no_ids() if self.remove_assistant_turns else torch.tensor([[output_ids[-1]]], device=self.device)
        self.num_frames_no_reply = 0
        self.last_role = "assistant"
        return self.tokenizer.decode(output_ids, skip_special_tokens=True, clean_up_tokenization_spaces=True)

    # ---- the response trigger, one rule for both drivers (test/inference.py:311-326, test/live_infer_for_video.py:151-165) ------
    def _response_due(self, video_scores) -> bool:
        """Fold one frame's scores into the running statistics and say whether a response is due: the sum over `score_heads`
        either exceeds the per-frame threshold, or pushes the running sum over its threshold (which then restarts at 0)."""
        frame_score = sum(v for k, v in video_scores.items() if k in self.score_heads)
        self.stream_end_prob_list.append(frame_score)
        keep = self.running_list_length
        if isinstance(keep, int) and keep > 0:
            del self.stream_end_prob_list[:-keep]
        self.stream_end_score_sum += frame_score
        due = False
        if self.stream_end_score_sum_threshold is not None and self.stream_end_score_sum > self.stream_end_score_sum_threshold:
            self.stream_end_score_sum = 0
            due = True
        if self.stream_end_prob_threshold is not None and frame_score > self.stream_end_prob_threshold:
            due = True
        return due

    def _respond(self):
        response = self._generate_response()
        self.num_frames_no_reply = 0
        self.consecutive_n_frames = 0
        return response

    # ---- test/inference.py:283-335 ---------------------------------------------------------------
    def _encode_frames_static_batched(self, frames_per_step: int, last_token_only: bool = False):
        """TrulyStaticCache only: once the cache is frozen a frame's scores do not depend on any other
        frame (test/static_cache.py:26-36), so `frames_per_step` queued frames are scored with ONE pass over
        the weights (the frozen stream is listed once per frame).  Bit-identical to scoring them one by
        one; used by inference() only while no query / prompt prefix has to be interleaved."""
        g = min(frames_per_step, len(self.frame_embeds_queue))
        embeds = torch.stack([self.frame_embeds_queue.popleft()[1] for _ in range(g)]).view(g, -1, self.hidden_size)
        if last_token_only:
            # Under the frozen static cache a new token attends ONLY to the prefix (never to the other
            # tokens of its own frame), and the drivers read the heads at position -1 only
            # (test/inference.py:222-227): the other T-1 tokens of a frame cannot influence its scores.
            # Feed just the last token, at the RoPE position it would have had.  Opt-in, bit-identical.
            T = embeds.shape[1]
            self.past_key_values.set_position_offset(T - 1)
            try:
                scores = self.rt.lm_step([self.past_key_values] * g, embeds[:, -1:].contiguous()).tolist()
            finally:
                self.past_key_values.set_position_offset(0)
        else:
            scores = self.rt.lm_step([self.past_key_values] * g, embeds).tolist()
        self.frame_idx += g
        self.num_frames_no_reply += g
        self.last_role = "stream"
        self.last_ids = self._no_ids()
        return scores

    def _can_batch_static(self, frames_per_step: int) -> bool:
        if not (frames_per_step > 1 and self.alt_cache == "static" and not self.sink_cache and len(self.frame_embeds_queue) > 1):
            return False
        frozen = self.init_vision_time and self.last_role == "stream" and self.past_key_values.get_seq_length() > 0
        horizon = self.video_time + (frames_per_step - 1) / self.frame_fps
        return frozen and not (self.query_queue and self.query_queue[0][0] <= horizon)

    @torch.no_grad()
    def inference(self, verbose=False, total=None, frames_per_step: int = 1, static_last_token_only: bool = False):
        """The benchmark loop (test/inference.py:283-348): per queued frame - a user query that has become due is encoded
        first, the frame is scored, the debug row is recorded, a response is generated when the trigger fires, and the clock
        advances by one frame interval.  Returns the conversation (queries + responses) ordered by time."""
        turns = [{"time": t, "content": q, "role": "user"} for t, q in self.query_queue]
        pending = collections.deque()        # scores already computed by a static batched step
        while self.frame_embeds_queue or pending:
            if self.query_queue and self.video_time >= self.query_queue[0][0]:
                self._encode_query()
            if not pending and self._can_batch_static(frames_per_step):
                pending.extend(self._encode_frames_static_batched(frames_per_step, static_last_token_only))
            if pending:
                info, rel, uncertainty_score = pending.popleft()
                video_scores = {"informative_score": info, "relevance_score": rel}
            else:
                video_scores, uncertainty_score = self._encode_frame()
            self.debug_data_list.append(dict(time=self.video_time, **video_scores, uncertainty_score=uncertainty_score))
            if self._response_due(video_scores):
                if pending:
                    raise RuntimeError("a response was triggered inside a static batched step: use frames_per_step=1 with response thresholds")
                turns.append({"time": self.video_time, "content": self._respond(), "role": "assistant"})
            self.video_time += 1 / self.frame_fps
            if verbose and self.frame_idx % 50 == 0:
                print(f"frame {self.frame_idx}" + (f"/{total}" if total else "") + f" {self.video_time:.2f}s", flush=True)
        return sorted(turns, key=lambda x: x["time"])


def sample_frame_indices(input_fps, frame_count, output_fps, max_num_frames=None, floor_total=False):
    """Which decoded frames the reference's loaders keep (load_video_for_testing, test/inference.py:509-571;
    floor_total=True: load_video, test/live_infer_for_video.py:42-43,74-75).  Decoding itself (cv2.VideoCapture) is
    outside this package: feed the kept frames to `frames_to_canvases`.  Returns (indices, output_fps, duration)."""
    video_duration = frame_count / input_fps
    output_fps = output_fps if output_fps > 0 else max_num_frames / video_duration
    total = math.floor(video_duration * output_fps) if floor_total else math.ceil(video_duration * output_fps)
    frame_sec = [i / output_fps for i in range(total)]
    keep, cur_time, frame_index = [], 0, 0
    for true_index in range(int(frame_count)):
        if frame_index < len(frame_sec) and cur_time >= frame_sec[frame_index]:
            keep.append(true_index)
            frame_index += 1
        if max_num_frames and len(keep) >= max_num_frames:
            break
        cur_time += 1 / input_fps
    return keep, output_fps, video_duration


def frames_to_canvases(rt, frames_hwc_u8, *, bgr=True, method=None) -> torch.Tensor:
    """Decoded frames (uint8 [h,w,3] tensors / arrays, B,G,R as cv2 delivers them unless bgr=False) -> uint8
    [N,3,S,S] on the device, what `input_video_stream` takes: the resize + pad + BGR2RGB + CHW body of
    load_video_for_testing (test/inference.py:538-562) run by aha_frame_ingest on the GPU."""
    method = rt.RESIZE_CV2_LINEAR if method is None else method
    S = rt.cfg.vision.image_size
    out = torch.empty((len(frames_hwc_u8), 3, S, S), dtype=torch.uint8, device=rt.device)
    for i, f in enumerate(frames_hwc_u8):
        f = f if torch.is_tensor(f) else torch.from_numpy(np.ascontiguousarray(f))
        rt.frame_ingest(f.to(rt.device, non_blocking=True), bgr=bgr, method=method, out=out[i])
    return out


class LiveInferForDemo(LiveInferForBenchmark):
    def __init__(self, args, peft_model_id=None, query=None, **kw):
        super().__init__(args, peft_model_id, **kw)
        self.system_prompt = "A multimodal AI assistant is helping users with some activities. \
        Below is their conversation, interleaved with the list of video frames received by the assistant."

    # ---- test/live_infer_for_video.py:98-128 -----------------------------------------------------------
    def load_one_frame(self, frame_path=None, frame_object=None):
        """frame_object: a PIL image / uint8 HxWx3 RGB array or tensor; frame_path: an image file (decoded with PIL).
        PIL-bicubic resize + centred pad to frame_resolution on the GPU (aha_frame_ingest, bit-exact with the
        reference's Image.resize + ImageOps.expand), encode ONE frame, queue its embedding."""
        assert frame_path is not None or frame_object is not None
        if frame_object is None:
            from PIL import Image            # optional dependency, only for file input
            frame_object = Image.open(frame_path).convert("RGB")
        if not torch.is_tensor(frame_object):
            frame_object = torch.from_numpy(np.ascontiguousarray(np.array(frame_object)))
        canvas = self.rt.frame_ingest(frame_object.to(self.device), bgr=False, method=self.rt.RESIZE_PIL_BICUBIC)
        frame_embeds = self.rt.visual_embed(canvas[None]).split(self.frame_num_tokens)
        self.frame_embeds_queue.append((self.video_time, frame_embeds[0]))

    # ---- test/live_infer_for_video.py:135-176 ----------------------------------------------------------
    def input_one_frame(self):
        video_scores, uncertainty_scores = self._encode_frame()
        ret = dict(frame_idx=self.frame_idx, time=round(self.video_time, 1), uncertainty_score=uncertainty_scores, **video_scores)
        ret["response"] = self._respond() if self._response_due(video_scores) else None
        self.video_time += 1 / self.frame_fps
        return ret

    # ---- test/live_infer_for_video.py:179-192 ----------------------------------------------------------
    def input_video(self, video_frames, query, fps=None):
        """video_frames: uint8 [T,3,S,S] already decoded/resized (decoding with cv2 is frame ingest,
        SURVEY.md 8f item 3).  Returns (results, model_response_list) like the reference."""
        conversation = [{"role": "system", "content": self.system_prompt}, {"role": "user", "content": query, "time": 0}]
        self.set_fps(fps=fps or self.frame_fps)
        self.input_video_stream(video_frames)
        self.input_query_stream(conversation)
        model_response_list = self.inference(verbose=False, total=len(video_frames))
        results = round_numbers(self.debug_data_list, 3)
        return results, model_response_list

    # ---- test/live_infer_for_video.py:195-228 ----------------------------------------------------------
    def find_ticks(self, scores, fps, min_separation=10, prominence=0.02, thresh=False, verbose=False):
        """Savitzky-Golay(15,3) smoothing + find_peaks(height=mean+0.5*std, prominence, distance)."""
        import numpy as np
        from scipy.signal import find_peaks, savgol_filter
        if not isinstance(scores, np.ndarray):
            scores = np.array(scores)
        smoothed = savgol_filter(scores, window_length=15, polyorder=3)
        if not thresh:
            thresh = smoothed.mean() + 0.5 * smoothed.std()
        min_separation = 10                       # the reference overrides its own argument (:214)
        distance = int(min_separation * fps)
        peaks, _ = find_peaks(smoothed, height=thresh, prominence=prominence, distance=distance)
        peak_times = peaks / fps
        if verbose:
            print("Detected spikes at:", peak_times)
        return list(peak_times)
