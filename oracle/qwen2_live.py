"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement, in plain torch ops, of the LM step of the reference's per-frame path:

  VideoHeadLiveLlavaQwenForCausalLM.forward, inference branch
      models/live_llava/video_head_live_llava_qwen.py:156-188,317-330
  which runs transformers' Qwen2Model (un-vendored third party, pinned ==4.49.0 in
  requirements.txt:57).  Arithmetic restated from the local copy
      transformers/models/qwen2/modeling_qwen2.py:35-298
  (RMSNorm :236-254, RoPE :51-131, attention :173-233, MLP :35-49, layer :257-298) with
  the rounding points of running those torch ops in the working dtype.

4.49-vs-local deltas handled here: the cache receives ``cache_kwargs``
with cos/sin; positions = ``cache.get_seq_length()`` + arange(T); attention mask:

  attn_semantics="trailing" (default, the parity target of SURVEY.md 8c):
      returned keys that precede the T new ones are all visible, the trailing TxT block
      is causal; when the policy returns no new keys (StaticPolicy after its first call)
      every returned key is visible.
  attn_semantics="hf449_sdpa": transformers-4.49 sdpa mask arithmetic: key j visible to
      new token i iff j <= L_before + i (mask built for L_before+T+1 columns then sliced to
      the returned key length).  Equals "trailing" while the cache grows; once a sink or
      sliding window is full it lets a new token see later tokens of its own chunk.

  attn_semantics="fa2": flash-attn-2, the reference's default attn_implementation
      (models/arguments_live.py:30): causal mask aligned to the bottom-right corner of the
      [T, Lk] score matrix, key j visible to new token i iff j <= i + (Lk - T).  Equal to
      "trailing" whenever the policy returns the new keys; for a frozen StaticPolicy (prefix
      only) the first T - Lk new tokens see no key at all and flash-attn returns 0 for them.
      The last token sees every prefix key under both rules, and under a frozen static cache a
      token's hidden state depends on no other new token, so the scores the drivers read at
      position -1 are identical under "trailing" and "fa2".

Pinned by tests/test_oracle_models.py against local transformers Qwen2Model + DynamicCache
(live, and through tests/golden/qwen2_tiny_steps.npz).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .cache_policies import GrowingPolicy, StaticPolicy, rotate_half


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    # modeling_qwen2.py:247-251
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(dt)


def rope_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float, dtype) -> tuple:
    # modeling_qwen2.py:87-102; fp32 table, cast to the working dtype before use
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = position_ids[:, :, None].to(torch.float32) * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


class OracleLM:
    def __init__(self, lm_cfg, weights: Dict[str, torch.Tensor], dtype=torch.bfloat16,
                 attn_semantics: str = "trailing", attn_impl: str = "sdpa") -> None:
        self.c = lm_cfg
        self.dtype = dtype
        self.w = {k: v.to(dtype) for k, v in weights.items()
                  if k.startswith(("model.", "lm_head", "informative_head", "relevance_head",
                                   "uncertainty_head"))}
        assert attn_semantics in ("trailing", "hf449_sdpa", "fa2")
        self.attn_semantics = attn_semantics
        self.attn_impl = attn_impl
        # PEFT LoRA adapters, UNMERGED as the reference runs them (PeftModel.from_pretrained, models/modeling_live.py:171-179;
        # peft==0.12.0 lora.Linear.forward: result = base(x) + lora_B(lora_A(x)) * scaling, in the working dtype):
        # {weight name: (A [r,in], B [out,r], scaling)}
        self.lora: Dict[str, tuple] = {}

    def attach_lora(self, adapters: Dict[str, tuple]) -> None:
        self.lora = {k: (a.to(self.dtype), b.to(self.dtype), float(s)) for k, (a, b, s) in adapters.items()}

    def _lin(self, x, name, bias=None):
        y = F.linear(x, self.w[name], bias)
        if name in self.lora:
            a, b, s = self.lora[name]
            y = y + F.linear(F.linear(x, a), b) * s
        return y

    # -- embeddings ---------------------------------------------------------------
    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        return F.embedding(ids, self.w["model.embed_tokens.weight"])

    # -- one decoder layer ----------------------------------------------------------
    def _attention(self, i, x, cos, sin, cache, L_before, static_frozen, tr=None):
        c = self.c
        B, T, _ = x.shape
        p = f"model.layers.{i}.self_attn."
        q = self._lin(x, p + "q_proj.weight", self.w[p + "q_proj.bias"])
        k = self._lin(x, p + "k_proj.weight", self.w[p + "k_proj.bias"])
        v = self._lin(x, p + "v_proj.weight", self.w[p + "v_proj.bias"])
        q = q.view(B, T, c.num_attention_heads, c.head_dim).transpose(1, 2)
        k = k.view(B, T, c.num_key_value_heads, c.head_dim).transpose(1, 2)
        v = v.view(B, T, c.num_key_value_heads, c.head_dim).transpose(1, 2)
        cu, su = cos.unsqueeze(1), sin.unsqueeze(1)
        q = (q * cu) + (rotate_half(q) * su)            # modeling_qwen2.py:128-129
        k = (k * cu) + (rotate_half(k) * su)
        if tr is not None:
            tr.update(q=q, k=k, v=v)                    # post-RoPE q/k and v, [B, heads, T, D]
        K, V = cache.update(k, v, i, {"sin": sin, "cos": cos})
        Lk = K.shape[-2]
        # visibility: key j visible to new token t iff j <= off + t
        if static_frozen:
            off = Lk - T if self.attn_semantics == "fa2" else Lk     # fa2: bottom-right aligned; else everything visible
        elif self.attn_semantics in ("trailing", "fa2"):
            off = Lk - T
        else:
            off = L_before
        jj = torch.arange(Lk)[None, :]
        tt = torch.arange(T)[:, None]
        mask = jj <= (off + tt)                         # [T, Lk] bool, True = attend
        g = c.num_attention_heads // c.num_key_value_heads
        Kr = K[:, :, None].expand(B, c.num_key_value_heads, g, Lk, c.head_dim).reshape(B, -1, Lk, c.head_dim)
        Vr = V[:, :, None].expand(B, c.num_key_value_heads, g, Lk, c.head_dim).reshape(B, -1, Lk, c.head_dim)
        scale = c.head_dim ** -0.5
        if self.attn_impl == "sdpa":
            o = F.scaled_dot_product_attention(q, Kr, Vr, attn_mask=mask[None, None], scale=scale)
        else:                                           # eager: modeling_qwen2.py:147-170
            aw = torch.matmul(q, Kr.transpose(2, 3)) * scale
            aw = aw + torch.where(mask, 0.0, torch.finfo(aw.dtype).min)[None, None].to(aw.dtype)
            aw = F.softmax(aw, dim=-1, dtype=torch.float32).to(q.dtype)
            o = torch.matmul(aw, Vr)
        dead = ~mask.any(dim=-1)                        # rows that see no key (fa2 + frozen static): flash-attn gives 0, not NaN
        if dead.any():
            o = torch.where(dead[None, None, :, None], torch.zeros_like(o), o)
        o = o.transpose(1, 2).reshape(B, T, -1)
        if tr is not None:
            tr.update(attn_out=o, K=K, V=V)
        return self._lin(o, p + "o_proj.weight")

    def _mlp(self, i, x, tr=None):
        p = f"model.layers.{i}.mlp."
        g = self._lin(x, p + "gate_proj.weight")
        u = self._lin(x, p + "up_proj.weight")
        a = F.silu(g) * u
        if tr is not None:
            tr.update(act=a)
        return self._lin(a, p + "down_proj.weight")

    # -- the step -----------------------------------------------------------------------
    @torch.no_grad()
    def step(self, inputs_embeds: torch.Tensor, cache, want_logits: bool = False, trace: Optional[list] = None) -> dict:
        """inputs_embeds [B,T,H] in the working dtype; ``cache`` a policy from
        oracle.cache_policies (mutated in place).  Returns the fields of
        VideoHeadCausalLMOutputWithPast the driver consumes
        (video_head_live_llava_qwen.py:317-330)."""
        c = self.c
        if cache is None:
            cache = GrowingPolicy()
        h = inputs_embeds.to(self.dtype)
        B, T, _ = h.shape
        L_before = cache.get_seq_length()
        static_frozen = isinstance(cache, StaticPolicy) and L_before > 0
        pos = (L_before + torch.arange(T))[None, :].expand(B, T)
        cos, sin = rope_cos_sin(pos, c.head_dim, c.rope_theta, self.dtype)
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            tr = {"x_in": h} if trace is not None else None   # per-layer tensors for the teacher-forced parity tests
            r = h
            x = rms_norm(h, self.w[p + "input_layernorm.weight"], c.rms_norm_eps)
            h = r + self._attention(i, x, cos, sin, cache, L_before, static_frozen, tr)
            r = h
            x = rms_norm(h, self.w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
            if tr is not None:
                tr.update(h_mid=h, x_mid=x)
            h = r + self._mlp(i, x, tr)
            if tr is not None:
                tr["h_out"] = h
                trace.append(tr)
        h = rms_norm(h, self.w["model.norm.weight"], c.rms_norm_eps)
        out = {"hidden": h, "past_key_values": cache}
        # video_head_live_llava_qwen.py:185-188
        out["informative_logits"] = F.linear(h, self.w["informative_head.weight"]).float()
        out["relevance_logits"] = torch.sigmoid(F.linear(h, self.w["relevance_head.weight"]).float())
        out["uncertainty"] = F.linear(h, self.w["uncertainty_head.weight"]).float()
        if want_logits:
            out["logits"] = F.linear(h, self.w["lm_head.weight"]).float()        # :175
        return out


def frame_scores(out: dict) -> torch.Tensor:
    """The three floats _encode_frame reads from the last token (test/inference.py:222-227):
    softmax(informative)[1], relevance (already sigmoid), exp(log-variance).  -> fp32 [B,3]"""
    info = out["informative_logits"][:, -1].softmax(dim=-1)[:, 1]
    rel = out["relevance_logits"][:, -1, 0]
    unc = torch.exp(out["uncertainty"][:, -1, 0])
    return torch.stack([info, rel, unc], dim=-1)
