"""Inference-relevant fields of the reference's argument dataclasses, same names and defaults
(models/arguments_live.py:5-75).  Training-only fields of TrainingArguments are not carried."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class LiveTestArguments:
    # LiveTrainingArguments (models/arguments_live.py:5-36)
    llm_pretrained: str = "lmms-lab/llava-onevision-qwen2-7b-ov"
    vision_pretrained: str = "google/siglip-large-patch16-384"
    lora_pretrained: Optional[str] = None
    frame_fps: float = 2
    frame_token_cls: bool = False
    frame_token_pooled: List[int] = field(default_factory=lambda: [7, 7])
    frame_num_tokens: int = 49
    video_pooling_stride: int = 4
    frame_resolution: int = 384
    v_placeholder: str = "<image>"
    max_num_frames: int = 100
    attn_implementation: str = "flash_attention_2"
    bf16: bool = True
    fp16: bool = False
    quantization: bool = False
    # LiveTestArguments (models/arguments_live.py:40-72)
    system_prompt: str = (
        "A multimodal AI assistant is helping users with some activities."
        " Below is their conversation, interleaved with the list of video frames received by the assistant.")
    grounding_mode: bool = False
    repetition_penalty: Optional[float] = None
    stream_end_prob_threshold: Optional[float] = None
    response_min_interval_frames: Optional[int] = None
    threshold_z: Optional[float] = None
    first_n_frames_no_generate: int = 0
    consecutive_n_frames_threshold: int = 1
    running_list_length: int = 20
    stream_end_score_sum_threshold: Optional[float] = None
    remove_assistant_turns: bool = False
    score_heads: str = "relevance_score,informative_score"
    uncertainty_wait_threshold: float = 0.0
    max_wait_frames: int = 3
    no_query: bool = False
