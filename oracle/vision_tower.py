"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the vision half of the reference's per-frame path:

  image_processor.preprocess            test/inference.py:176 (impl in the un-vendored
                                        LLaVA-NeXT submodule: x/255 then (x-.5)/.5)
  vision_encode -> vision_tower         models/live_llava/video_head_live_llava_qwen.py:113-115
  connector (mm_projector, mlp2x_gelu)  video_head_live_llava_qwen.py:107-108
  post_projector_pooling                video_head_live_llava_qwen.py:117-136
  visual_embed                          models/modeling_live.py:31-37
  _siglip_vision_encode (dead-code spec) models/vision_live.py:11-31
  _clip_vision_encode (dead-code spec)   models/vision_live.py:34-54  (CLIP tower restated from the local
                                        transformers copy, models/clip/modeling_clip.py: CLIPVisionEmbeddings,
                                        CLIPEncoderLayer, CLIPVisionTransformer; pinned in tests/test_oracle_models.py)

LLaVA-NeXT is absent from /root/reference (empty submodule dir), so the tower arithmetic
is restated from the published SigLIP architecture as implemented by the local
transformers copy, transformers/models/siglip/modeling_siglip.py:116-358 (embeddings
:116-186, attention :251-308, MLP :312-324, layer :327-358): the llava tower returns the
last executed encoder layer's hidden state with no post-layernorm and no pooling head.

Pinned by tests/test_oracle_models.py against local transformers SiglipVisionModel.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F


def preprocess(frames_u8: torch.Tensor, dtype) -> torch.Tensor:
    """uint8 [N,3,S,S] -> working dtype; rescale 1/255 then normalize mean .5 std .5 in fp32
    (vision_live.py:11-13 states the same constants)."""
    x = frames_u8.to(torch.float32) * 0.00392156862745098
    x = (x - 0.5) / 0.5
    return x.to(dtype)


class OracleVision:
    def __init__(self, cfg, weights: Dict[str, torch.Tensor], dtype=torch.bfloat16) -> None:
        """cfg: aha_amd.config.LiveConfig-like (needs .vision, .lm.hidden_size, pooling knobs)."""
        self.cfg = cfg
        self.v = cfg.vision
        self.dtype = dtype
        self.w = {k: t.to(dtype) for k, t in weights.items() if k.startswith(("vision.", "mm_projector."))}

    @torch.no_grad()
    def tower(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[N,3,S,S] working dtype -> [N,Np,Dv] (modeling_siglip.py:176-186, 327-358)."""
        v, w = self.v, self.w
        x = F.conv2d(pixel_values, w["vision.embeddings.patch_embedding.weight"],
                     w["vision.embeddings.patch_embedding.bias"], stride=v.patch_size)
        x = x.flatten(2).transpose(1, 2)
        x = x + w["vision.embeddings.position_embedding.weight"][None]
        for i in range(v.num_hidden_layers):
            x = self.encoder_layer(x, i)
        return x

    @torch.no_grad()
    def layer_inputs(self, pixel_values: torch.Tensor):
        """The hidden state entering every encoder layer, plus the tower output (teacher-forcing taps of the per-layer parity tests)."""
        v, w = self.v, self.w
        x = F.conv2d(pixel_values, w["vision.embeddings.patch_embedding.weight"],
                     w["vision.embeddings.patch_embedding.bias"], stride=v.patch_size)
        x = x.flatten(2).transpose(1, 2) + w["vision.embeddings.position_embedding.weight"][None]
        taps = []
        for i in range(v.num_hidden_layers):
            taps.append(x)
            x = self.encoder_layer(x, i)
        return taps, x

    @torch.no_grad()
    def encoder_layer(self, x: torch.Tensor, i: int) -> torch.Tensor:
        """SiglipEncoderLayer i (modeling_siglip.py:327-358): [N,Np,Dv] -> [N,Np,Dv]."""
        v, w = self.v, self.w
        N, Np, Dv = x.shape
        nh, hd = v.num_attention_heads, v.head_dim
        p = f"vision.encoder.layers.{i}."
        r = x
        h = F.layer_norm(x, (Dv,), w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], v.layer_norm_eps)
        q = F.linear(h, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"])
        k = F.linear(h, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"])
        vv = F.linear(h, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"])
        q = q.view(N, Np, nh, hd).transpose(1, 2)
        k = k.view(N, Np, nh, hd).transpose(1, 2)
        vv = vv.view(N, Np, nh, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, vv, scale=hd ** -0.5)
        o = o.transpose(1, 2).reshape(N, Np, Dv)
        o = F.linear(o, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
        x = r + o
        r = x
        h = F.layer_norm(x, (Dv,), w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], v.layer_norm_eps)
        h = F.linear(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])
        h = F.gelu(h, approximate="tanh")
        h = F.linear(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
        return r + h

    @torch.no_grad()
    def connector(self, feats: torch.Tensor) -> torch.Tensor:
        """mlp2x_gelu: Linear(Dv->H), GELU (exact erf), Linear(H->H)."""
        w = self.w
        h = F.linear(feats, w["mm_projector.0.weight"], w["mm_projector.0.bias"])
        h = F.gelu(h)
        return F.linear(h, w["mm_projector.2.weight"], w["mm_projector.2.bias"])

    @torch.no_grad()
    def post_projector_pooling(self, image_feature: torch.Tensor) -> torch.Tensor:
        """video_head_live_llava_qwen.py:117-136."""
        stride = self.cfg.video_pooling_stride
        g = self.v.grid
        n, _, d = image_feature.shape
        x = image_feature.view(n, g, g, -1).permute(0, 3, 1, 2).contiguous()
        mode = self.cfg.mm_spatial_pool_mode
        if mode == "average":
            x = F.avg_pool2d(x, stride)
        elif mode == "max":
            x = F.max_pool2d(x, stride)
        elif mode == "bilinear":
            hh, ww = x.shape[2:]
            x = F.interpolate(x, size=[math.ceil(hh / stride), math.ceil(ww / stride)], mode="bilinear")
        else:
            raise ValueError(f"Unexpected mm_spatial_pool_mode: {mode}")
        x = x.permute(0, 2, 3, 1)
        return x.reshape(n, -1, d).contiguous()

    @torch.no_grad()
    def visual_embed(self, frames_u8: torch.Tensor) -> torch.Tensor:
        """uint8 frames -> [N*Tf, H] (modeling_live.py:31-37 after test/inference.py:176)."""
        x = preprocess(frames_u8, self.dtype)
        x = self.tower(x)
        x = self.connector(x)
        x = self.post_projector_pooling(x)
        return x.reshape(-1, x.shape[-1])


def siglip_pooling_head(ov: "OracleVision", last_hidden_state: torch.Tensor) -> torch.Tensor:
    """SiglipVisionTransformer.head = SiglipMultiheadAttentionPoolingHead (transformers modeling_siglip.py; `pooler_output`
    of the vision model, read by models/vision_live.py:27 when frame_token_cls is set): a learned probe attends over the
    post-layernormed tokens (nn.MultiheadAttention, batch_first), then x + mlp(layernorm(x)).  [N,Np,Dv] -> [N,Dv].
    Pinned against the local transformers class in tests/test_oracle_models.py."""
    v, w = ov.v, ov.w
    n, _, d = last_hidden_state.shape
    nh, hd = v.num_attention_heads, v.head_dim
    wi, bi = w["vision.head.attention.in_proj_weight"], w["vision.head.attention.in_proj_bias"]
    q = F.linear(w["vision.head.probe"].reshape(1, 1, d).expand(n, 1, d), wi[:d], bi[:d])
    k = F.linear(last_hidden_state, wi[d:2 * d], bi[d:2 * d])
    vv = F.linear(last_hidden_state, wi[2 * d:], bi[2 * d:])
    q = q.view(n, 1, nh, hd).transpose(1, 2)
    k = k.view(n, -1, nh, hd).transpose(1, 2)
    vv = vv.view(n, -1, nh, hd).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, vv, scale=hd ** -0.5).transpose(1, 2).reshape(n, 1, d)
    x = F.linear(o, w["vision.head.attention.out_proj.weight"], w["vision.head.attention.out_proj.bias"])
    h = F.layer_norm(x, (d,), w["vision.head.layernorm.weight"], w["vision.head.layernorm.bias"], v.layer_norm_eps)
    h = F.gelu(F.linear(h, w["vision.head.mlp.fc1.weight"], w["vision.head.mlp.fc1.bias"]), approximate="tanh")
    x = x + F.linear(h, w["vision.head.mlp.fc2.weight"], w["vision.head.mlp.fc2.bias"])
    return x[:, 0]


def vision_live_encode(ov: "OracleVision", frames_u8: torch.Tensor, post_ln_w: torch.Tensor, post_ln_b: torch.Tensor,
                       frame_token_pooled=(7, 7), frame_token_cls: bool = False) -> torch.Tensor:
    """models/vision_live.py:11-31 (_siglip_vision_encode) then LiveMixin.visual_embed's connector
    (models/modeling_live.py:31-37; that model class has no post_projector_pooling):
      normalize(frames * 1/255, .5, .5) -> vision_model(frames).last_hidden_state (= tower + post_layernorm,
      transformers modeling_siglip.py:622-644) -> adaptive_avg_pool2d over the patch grid (frame_token_pooled), and with
      frame_token_cls the pooling head's output in front of it (vision_live.py:26-31) -> connector.
    Returns [N*(cls + ph*pw), H]; frame_token_pooled = None with frame_token_cls gives the class token alone [N, H]."""
    dt, v = ov.dtype, ov.v
    x = ov.tower(preprocess(frames_u8, dt))
    x = F.layer_norm(x, (v.hidden_size,), post_ln_w.to(dt), post_ln_b.to(dt), v.layer_norm_eps)
    n, _, d = x.shape
    toks = []
    if frame_token_cls:
        toks.append(siglip_pooling_head(ov, x)[:, None])
    if frame_token_pooled:
        s = int(math.sqrt(x.shape[1]))
        sp = F.adaptive_avg_pool2d(x.reshape(n, s, s, d).permute(0, 3, 1, 2), tuple(frame_token_pooled))
        toks.append(sp.flatten(2, 3).permute(0, 2, 1))
    y = ov.connector(torch.cat(toks, dim=1))
    return y.reshape(-1, y.shape[-1])


# ---- CLIP (models/vision_live.py:34-54) -----------------------------------------------------------------
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # transformers.utils.constants
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_clip(frames_u8: torch.Tensor, dtype) -> torch.Tensor:
    """normalize(frames * 1/255, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD) in fp32 (vision_live.py:34-36), then the working dtype"""
    x = frames_u8.to(torch.float32) * 0.00392156862745098
    mean = torch.tensor(OPENAI_CLIP_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(OPENAI_CLIP_STD, dtype=torch.float32).view(1, 3, 1, 1)
    return ((x - mean) / std).to(dtype)


class OracleCLIPVision(OracleVision):
    """CLIPVisionTransformer up to `last_hidden_state` (the encoder output, WITHOUT post_layernorm, which transformers
    applies to the pooled class token only): bias-free patch conv, class embedding prepended, learned positions,
    pre_layrnorm, pre-LN encoder layers with quick_gelu (x * sigmoid(1.702 x)).  Token 0 is the class token."""

    @torch.no_grad()
    def tower(self, pixel_values: torch.Tensor) -> torch.Tensor:
        v, w = self.v, self.w
        x = F.conv2d(pixel_values, w["vision.embeddings.patch_embedding.weight"], None, stride=v.patch_size)
        x = x.flatten(2).transpose(1, 2)
        cls = w["vision.embeddings.class_embedding"].expand(x.shape[0], 1, -1)
        x = torch.cat([cls, x], dim=1) + w["vision.embeddings.position_embedding.weight"][None]
        N, T, Dv = x.shape
        x = F.layer_norm(x, (Dv,), w["vision.pre_layrnorm.weight"], w["vision.pre_layrnorm.bias"], v.layer_norm_eps)
        nh, hd = v.num_attention_heads, v.head_dim
        for i in range(v.num_hidden_layers):
            p = f"vision.encoder.layers.{i}."
            r = x
            h = F.layer_norm(x, (Dv,), w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], v.layer_norm_eps)
            q = F.linear(h, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"]).view(N, T, nh, hd).transpose(1, 2)
            k = F.linear(h, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"]).view(N, T, nh, hd).transpose(1, 2)
            vv = F.linear(h, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"]).view(N, T, nh, hd).transpose(1, 2)
            o = F.scaled_dot_product_attention(q, k, vv, scale=hd ** -0.5).transpose(1, 2).reshape(N, T, Dv)
            x = r + F.linear(o, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
            r = x
            h = F.layer_norm(x, (Dv,), w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], v.layer_norm_eps)
            h = F.linear(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])
            h = h * torch.sigmoid(1.702 * h)                       # QuickGELUActivation
            x = r + F.linear(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
        return x


def clip_visual_embed(ov: "OracleCLIPVision", frames_u8: torch.Tensor) -> torch.Tensor:
    """LiveMixin.visual_embed with a CLIP tower the way LLaVA wires one (select_feature = 'patch': class token dropped,
    CLIPImageProcessor's OpenAI mean/std): tower -> patch features -> connector -> post_projector_pooling."""
    x = ov.tower(preprocess_clip(frames_u8, ov.dtype))[:, 1:]
    x = ov.post_projector_pooling(ov.connector(x))
    return x.reshape(-1, x.shape[-1])


def clip_live_encode(ov: "OracleCLIPVision", frames_u8: torch.Tensor, frame_token_pooled=(7, 7), frame_token_cls: bool = False) -> torch.Tensor:
    """models/vision_live.py:34-54 (_clip_vision_encode) then LiveMixin.visual_embed's connector:
    normalize with the OpenAI CLIP constants -> last_hidden_state -> drop the class token -> adaptive_avg_pool2d over the
    patch grid -> connector.  Returns [N*ph*pw, H].  frame_token_cls without pooling returns the class token alone,
    last_hidden_state[:, 0] (vision_live.py:50-53) -> [N, H]; with pooling the reference's torch.cat of a 2-D and a 3-D
    tensor (vision_live.py:54) raises, and so does this."""
    x = ov.tower(preprocess_clip(frames_u8, ov.dtype))
    if frame_token_cls:
        if frame_token_pooled:
            raise RuntimeError("_clip_vision_encode: torch.cat of [N, D] and [N, P, D] (models/vision_live.py:54)")
        return ov.connector(x[:, 0])
    n, t, d = x.shape
    s = int(math.sqrt(t))                                          # the reference takes sqrt of Np + 1 and truncates
    sp = F.adaptive_avg_pool2d(x[:, 1:].reshape(n, s, s, d).permute(0, 3, 1, 2), tuple(frame_token_pooled))
    y = ov.connector(sp.flatten(2, 3).permute(0, 2, 1))
    return y.reshape(-1, y.shape[-1])
