"""Python host over the C ABI: device memory and streams come from PyTorch-ROCm (plumbing), every
FLOP of the hot path runs in libaha_amd.so.  Nothing here imports ``oracle``."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, List, Optional, Sequence

import torch

from . import lib as _l
from .config import LiveConfig

_POOL_MODE = {"bilinear": 0, "average": 1, "max": 2}
_POLICY = {None: _l.CACHE_NONE, "none": _l.CACHE_NONE, "default_sink": _l.CACHE_SINK,
           "sliding_window": _l.CACHE_SLIDING, "static": _l.CACHE_STATIC}


class AhaError(RuntimeError):
    pass


def _cur_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def rope_table(n_pos: int, head_dim: int, theta: float):
    """cos/sin exactly as Qwen2RotaryEmbedding.forward produces them (fp32 math, cast to bf16;
    transformers modeling_qwen2.py:87-102): emb = cat(freqs, freqs)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(torch.bfloat16), emb.sin().to(torch.bfloat16)


def rerotation_table(cos: torch.Tensor, sin: torch.Tensor, window: int, n_sink: int, T: int):
    """SinkCache._get_rerotation_cos_sin (test/sink_cache.py:35-55) on rows [0, window) of the
    bf16 RoPE table: rotate a kept key back by T positions."""
    c = cos[:window].to(torch.float32)
    s = sin[:window].to(torch.float32)
    oc, sc = c[n_sink + T:], c[n_sink:-T]
    os_, ss = s[n_sink + T:], s[n_sink:-T]
    rc = oc * sc + os_ * ss
    rs = -os_ * sc + oc * ss
    return rc.to(torch.bfloat16).contiguous(), rs.to(torch.bfloat16).contiguous()


class Stream:
    """One video stream's KV state = the reference's Cache object (test/*_cache.py)."""

    def __init__(self, rt: "Runtime", alt_cache: Optional[str], window_length: int, num_sink_tokens: int,
                 capacity: int, attn_semantics: str = "trailing"):
        self.rt = rt
        self.policy = _POLICY[alt_cache]
        self.window_length, self.num_sink_tokens = window_length, num_sink_tokens
        h = C.c_void_p()
        rt._chk(rt.lib.aha_stream_open(rt.ctx, self.policy, window_length, num_sink_tokens, capacity, C.byref(h)))
        self.handle = h
        rt._streams.add(self)
        if attn_semantics != "trailing":            # "hf449_sdpa": transformers-4.49 sdpa mask arithmetic; "fa2": flash-attn-2 alignment
            rt._chk(rt.lib.aha_stream_set_attn_semantics(h, {"hf449_sdpa": _l.ATTN_HF449_SDPA, "fa2": _l.ATTN_FA2}[attn_semantics]))

    def reset(self):
        self.rt._chk(self.rt.lib.aha_stream_reset(self.handle))

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.rt.lib.aha_stream_seq_length(self.handle)

    def set_position_offset(self, offset: int):
        self.rt._chk(self.rt.lib.aha_stream_set_position_offset(self.handle, int(offset)))

    @property
    def seen_tokens(self) -> int:
        return self.rt.lib.aha_stream_seen_tokens(self.handle)

    def export_kv(self, layer: int, want_v: bool = False) -> torch.Tensor:
        n = self.get_seq_length()
        d = self.rt.desc
        out = torch.empty((d.kv_heads, n, d.head_dim), dtype=torch.bfloat16, device=self.rt.device)
        if n == 0:
            return out
        self.rt._chk(self.rt.lib.aha_stream_export_kv(self.rt.ctx, self.handle, layer, int(want_v), out.data_ptr(), _cur_stream()))
        return out

    def close(self):
        # aha_stream_destroy does not touch the context, so a stream may outlive its Runtime without leaking its K/V buffers
        if self.handle is not None:
            self.rt.lib.aha_stream_destroy(self.handle)
        self.handle = None
        self.rt._streams.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Runtime:
    """aha_ctx owner: weights, tables, vision encode, LM step."""

    def __init__(self, cfg: LiveConfig, weights: Dict[str, torch.Tensor], *, device: str = "cuda:0",
                 max_step_tokens: int = 512, max_vit_frames: int = 32, max_positions: Optional[int] = None):
        if not torch.cuda.is_available():
            raise AhaError("aha_amd needs a GPU: the product path has no CPU fallback")
        self.lib = _l.get()
        self._streams = weakref.WeakSet()            # open Streams: closed with the Runtime (their K/V buffers are multi-GB at 7B)
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        v, lm = cfg.vision, cfg.lm
        self.desc = _l.ModelDesc(
            image_size=v.image_size, patch_size=v.patch_size, v_hidden=v.hidden_size, v_layers=v.num_hidden_layers,
            v_heads=v.num_attention_heads, v_inter=v.intermediate_size, v_ln_eps=v.layer_norm_eps,
            hidden=lm.hidden_size, layers=lm.num_hidden_layers, heads=lm.num_attention_heads,
            kv_heads=lm.num_key_value_heads, head_dim=lm.head_dim, inter=lm.intermediate_size, vocab=lm.vocab_size,
            rope_theta=lm.rope_theta, rms_eps=lm.rms_norm_eps,
            max_positions=max_positions or lm.max_position_embeddings, pool_stride=cfg.video_pooling_stride,
            pool_mode=_POOL_MODE[cfg.mm_spatial_pool_mode], max_step_tokens=max_step_tokens, max_vit_frames=max_vit_frames,
            v_kind={"siglip": 0, "clip": 1}[v.kind])
        ctx = C.c_void_p()
        rc = self.lib.aha_ctx_create(C.byref(self.desc), self.device.index or 0, C.byref(ctx))
        self.ctx = ctx if ctx.value else None
        self.frame_num_tokens = cfg.frame_num_tokens
        self.hidden_size = lm.hidden_size
        try:
            self._chk(rc)
            self._load(weights)
            cos, sin = rope_table(self.desc.max_positions, lm.head_dim, lm.rope_theta)
            self._rope_cpu = (cos, sin)
            cd, sd = cos.to(self.device), sin.to(self.device)
            self._chk(self.lib.aha_ctx_set_rope_table(self.ctx, cd.data_ptr(), sd.data_ptr(), cos.shape[0], _cur_stream()))
        except Exception:
            self.close()                 # a half-built context must not leak its device allocations
            raise

    # -- plumbing -------------------------------------------------------------------------------
    def _chk(self, rc: int):
        if rc != 0:
            msg = self.lib.aha_last_error(self.ctx)
            raise AhaError(f"aha_amd error {rc}: {msg.decode() if msg else ''}")

    def _load(self, weights: Dict[str, torch.Tensor]):
        keep, views = [], []
        for name, t in weights.items():
            t = t.to(device=self.device, dtype=torch.bfloat16).contiguous()
            keep.append(t)
            tv = _l.TensorView()
            tv.name = name.encode()
            tv.data = t.data_ptr()
            tv.ndim = t.dim()
            for i, s in enumerate(t.shape[:4]):
                tv.shape[i] = s
            if t.dim() == 4:                         # conv weight [Dv,3,P,P] is used as [Dv, 3*P*P]
                tv.shape[0], tv.shape[1], tv.ndim = t.shape[0], t.shape[1] * t.shape[2] * t.shape[3], 2
            views.append(tv)
        arr = (_l.TensorView * len(views))(*views)
        self._chk(self.lib.aha_ctx_load_weights(self.ctx, arr, len(views), _cur_stream()))
        del keep

    def set_tuning(self, key: str, value: int):
        self._chk(self.lib.aha_ctx_set_tuning(self.ctx, key.encode(), int(value)))

    def close(self):
        for s in list(getattr(self, "_streams", ())):
            s.close()
        if self.ctx is not None:
            self.lib.aha_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- streams ---------------------------------------------------------------------------------
    def open_stream(self, alt_cache: Optional[str] = "default_sink", window_length: int = 2048, num_sink_tokens: int = 32,
                    capacity: Optional[int] = None, attn_semantics: str = "trailing") -> Stream:
        return Stream(self, alt_cache, window_length, num_sink_tokens, capacity or self.desc.max_positions, attn_semantics)

    def set_rerotation_table(self, s: Stream, T: int):
        """Optional: install a host-computed SinkCache re-rotation table for (window, n_sink, T).  aha_lm_step builds the same
        table on the device, asynchronously, the first time a combination evicts, so the step path never calls this; it
        exists for callers that want to supply the table themselves (and for the bit-exactness test of the device build)."""
        if s.policy != _l.CACHE_SINK:
            return
        W, k = s.window_length, s.num_sink_tokens
        if W - k - T <= 0 or self.lib.aha_ctx_has_rerotation_table(self.ctx, W, k, T):
            return
        rc, rs = rerotation_table(self._rope_cpu[0], self._rope_cpu[1], W, k, T)
        rc, rs = rc.to(self.device), rs.to(self.device)
        self._chk(self.lib.aha_ctx_set_rerotation_table(self.ctx, W, k, T, rc.data_ptr(), rs.data_ptr(), _cur_stream()))

    # -- the hot path ------------------------------------------------------------------------------
    def visual_embed(self, frames_u8: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8 [N,3,S,S] on device -> bf16 [N*Tf, H] (LiveMixin.visual_embed, modeling_live.py:31-37,
        with image_processor.preprocess fused in).  Runs on the CURRENT torch stream; pass a
        preallocated `out` when encoding on a side stream (overlap with LM steps)."""
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.dim() == 4
        frames_u8 = frames_u8.contiguous()
        n = frames_u8.shape[0]
        if out is None:
            out = torch.empty((n * self.frame_num_tokens, self.hidden_size), dtype=torch.bfloat16, device=self.device)
        assert out.is_contiguous() and out.numel() == n * self.frame_num_tokens * self.hidden_size
        step = self.desc.max_vit_frames
        for i in range(0, n, step):
            m = min(step, n - i)
            self._chk(self.lib.aha_vit_encode(self.ctx, frames_u8[i:i + m].data_ptr(), m,
                                              out[i * self.frame_num_tokens:].data_ptr(), _cur_stream()))
        return out

    RESIZE_PIL_BICUBIC, RESIZE_CV2_LINEAR = 0, 1

    def frame_ingest(self, frame_hwc_u8: torch.Tensor, *, bgr: bool = False, method: int = 0,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One decoded frame, uint8 [h,w,3] on device (B,G,R order when `bgr`) -> uint8 [3,S,S] RGB canvas:
        aspect-preserving resize + centred zero pad, bit-exact with the reference's resampler for that path
        (method RESIZE_PIL_BICUBIC: load_one_frame, test/live_infer_for_video.py:98-121; RESIZE_CV2_LINEAR:
        load_video_for_testing, test/inference.py:538-562)."""
        assert frame_hwc_u8.dtype == torch.uint8 and frame_hwc_u8.is_cuda and frame_hwc_u8.dim() == 3 and frame_hwc_u8.shape[2] == 3
        frame_hwc_u8 = frame_hwc_u8.contiguous()
        S = self.cfg.vision.image_size
        if out is None:
            out = torch.empty((3, S, S), dtype=torch.uint8, device=self.device)
        assert out.is_contiguous() and out.shape == (3, S, S) and out.dtype == torch.uint8
        h, w, _ = frame_hwc_u8.shape
        self._chk(self.lib.aha_frame_ingest(self.ctx, frame_hwc_u8.data_ptr(), h, w, int(bool(bgr)), int(method),
                                            out.data_ptr(), _cur_stream()))
        return out

    def vision_live_embed(self, frames_u8: torch.Tensor, pooled: int = 7, cls: bool = False) -> torch.Tensor:
        """models/vision_live.py:11-31 / :34-54 contract + connector: tower -> last_hidden_state (SigLIP: post_layernorm) ->
        adaptive average pool to pooled x pooled, with `cls` (frame_token_cls) the class token in front of it (SigLIP: the
        attention-pooling head's pooler_output; pooled = 0: the class token alone) -> mm_projector.
        uint8 [N,3,S,S] -> bf16 [N*(cls + pooled*pooled), H]."""
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.dim() == 4
        frames_u8 = frames_u8.contiguous()
        n, tok = frames_u8.shape[0], (1 if cls else 0) + pooled * pooled
        out = torch.empty((n * tok, self.hidden_size), dtype=torch.bfloat16, device=self.device)
        step = self.desc.max_vit_frames
        for i in range(0, n, step):
            m = min(step, n - i)
            self._chk(self.lib.aha_vit_encode_live(self.ctx, frames_u8[i:i + m].data_ptr(), m, int(pooled), int(bool(cls)),
                                                   out[i * tok:].data_ptr(), _cur_stream()))
        return out

    def tower_output(self, n_frames: int) -> torch.Tensor:
        """bf16 [n*Tt, Dv] tower output of the last encode (test tap; copies).  Tt = Np, or Np + 1 for a CLIP tower, whose class
        token is the LAST row of each frame's block (the oracle / transformers keep it first)."""
        rows, dv = n_frames * (self.cfg.vision.num_patches + (1 if self.cfg.vision.kind == "clip" else 0)), self.cfg.vision.hidden_size
        out = torch.empty((rows, dv), dtype=torch.bfloat16, device=self.device)
        self._chk(self.lib.aha_vit_last_tower_output(self.ctx, n_frames, out.data_ptr(), _cur_stream()))
        return out

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        ids = ids.to(device=self.device, dtype=torch.long).contiguous().view(-1)
        out = torch.empty((ids.numel(), self.hidden_size), dtype=torch.bfloat16, device=self.device)
        if ids.numel():
            self._chk(self.lib.aha_embed_tokens(self.ctx, ids.data_ptr(), ids.numel(), out.data_ptr(), _cur_stream()))
        return out

    def lm_step(self, streams: Sequence[Stream], embeds: torch.Tensor, *, want_raw: bool = False,
                want_hidden: bool = False, out: Optional[torch.Tensor] = None):
        """embeds bf16 [B,T,H] -> scores fp32 [B,3] (+ raw head logits [B,4], last hidden [B,H]).  `out`: a contiguous fp32 [B,3]
        device tensor the scores are written into (a row of a per-step score table: no allocation, no copy kernel)."""
        assert embeds.is_cuda and embeds.dtype == torch.bfloat16 and embeds.dim() == 3
        embeds = embeds.contiguous()
        B, T, _ = embeds.shape
        assert B == len(streams)
        if out is not None:
            assert out.is_cuda and out.dtype == torch.float32 and out.shape == (B, 3) and out.is_contiguous()
        scores = out if out is not None else torch.empty((B, 3), dtype=torch.float32, device=self.device)
        raw = torch.empty((B, 4), dtype=torch.float32, device=self.device) if want_raw else None
        hid = torch.empty((B, self.hidden_size), dtype=torch.bfloat16, device=self.device) if want_hidden else None
        arr = (C.c_void_p * B)(*[s.handle for s in streams])
        self._chk(self.lib.aha_lm_step(self.ctx, arr, B, embeds.data_ptr(), T, scores.data_ptr(),
                                       raw.data_ptr() if raw is not None else None,
                                       hid.data_ptr() if hid is not None else None, _cur_stream()))
        out = [scores]
        if want_raw:
            out.append(raw)
        if want_hidden:
            out.append(hid)
        return out[0] if len(out) == 1 else tuple(out)

    def heads_all(self, B: int, T: int) -> torch.Tensor:
        raw = torch.empty((B * T, 4), dtype=torch.float32, device=self.device)
        self._chk(self.lib.aha_lm_heads_all(self.ctx, raw.data_ptr(), _cur_stream()))
        return raw.view(B, T, 4)

    def last_hidden_all(self, B: int, T: int) -> torch.Tensor:
        out = torch.empty((B * T, self.hidden_size), dtype=torch.bfloat16, device=self.device)
        self._chk(self.lib.aha_lm_last_hidden_all(self.ctx, out.data_ptr(), _cur_stream()))
        return out.view(B, T, self.hidden_size)

    def logits_last(self, B: int, want_logits: bool = True):
        lg = torch.empty((B, self.cfg.lm.vocab_size), dtype=torch.float32, device=self.device) if want_logits else None
        am = torch.empty((B,), dtype=torch.long, device=self.device)
        self._chk(self.lib.aha_lm_logits_last(self.ctx, lg.data_ptr() if lg is not None else None, am.data_ptr(), _cur_stream()))
        return lg, am

    def logits_all(self, B: int, T: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """lm_head over every position of the last step: fp32 [B,T,V] (video_head_live_llava_qwen.py:175).  `out`: a contiguous fp32
        device tensor of B*T*V elements to write into."""
        if out is not None:
            assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == B * T * self.cfg.lm.vocab_size
        lg = out.view(B * T, self.cfg.lm.vocab_size) if out is not None else \
            torch.empty((B * T, self.cfg.lm.vocab_size), dtype=torch.float32, device=self.device)
        self._chk(self.lib.aha_lm_logits_all(self.ctx, lg.data_ptr(), _cur_stream()))
        return lg.view(B, T, -1)

    def generate_greedy(self, stream: Stream, first_ids: torch.Tensor, max_new_tokens: int, eos_token_id: int,
                        repetition_penalty: Optional[float] = None, generated_token_ids: Optional[list] = None, *,
                        chunk: Optional[int] = None, between_chunks=None, on_token=None) -> List[int]:
        """fast_greedy_generate (models/modeling_live.py:64-90): returns the new token ids (the last one is EOS unless the
        limit was hit) and, with a repetition penalty, appends the non-EOS ones to `generated_token_ids` like the reference.
        chunk: produce the response `chunk` tokens per aha_generate_greedy call (same ids as one call), running
        `between_chunks()` in between - other streams can step while a response is being written.
        on_token(token_id, index) -> truthy stops after that token (aha_generate_greedy_cb); it is called on the host as soon
        as the id is visible and may enqueue steps of other streams."""
        ids = first_ids.to(device=self.device, dtype=torch.long).contiguous().view(-1)
        pen = float(repetition_penalty) if repetition_penalty is not None else 0.0
        hist, hlen = None, C.c_int(0)
        if pen > 0:
            prev = list(generated_token_ids or [])
            hist = torch.zeros((len(prev) + max_new_tokens,), dtype=torch.long, device=self.device)
            if prev:
                hist[:len(prev)] = torch.tensor(prev, dtype=torch.long)
            hlen = C.c_int(len(prev))
        stopped = []
        cb = None
        if on_token is not None:
            def _cb(_user, tok, idx, _base=[0]):
                stop = 1 if on_token(int(tok), _base[0] + idx) else 0
                if stop:
                    stopped.append(True)
                return stop
            cb = _l.TOKEN_CB(_cb)
        toks: List[int] = []
        step = max_new_tokens if not chunk or chunk <= 0 else int(chunk)
        while len(toks) < max_new_tokens:
            k = min(step, max_new_tokens - len(toks))
            out = (C.c_int64 * k)()
            n = C.c_int(0)
            if cb is not None:
                _cb.__defaults__[0][0] = len(toks)
            self._chk(self.lib.aha_generate_greedy_cb(
                self.ctx, stream.handle, ids.data_ptr(), ids.numel(), k, int(eos_token_id), pen,
                hist.data_ptr() if hist is not None else None, hist.numel() if hist is not None else 0, C.byref(hlen), out, C.byref(n),
                C.cast(cb, C.c_void_p) if cb is not None else None, None, _cur_stream()))
            toks.extend(int(out[i]) for i in range(n.value))
            if n.value < k or toks[-1] == eos_token_id or stopped:
                break
            if len(toks) < max_new_tokens:
                ids = torch.tensor([toks[-1]], dtype=torch.long, device=self.device)    # the next chunk continues from the last id
                if between_chunks is not None:
                    between_chunks()
        if pen > 0 and generated_token_ids is not None:
            generated_token_ids.extend(t for t in toks if t != eos_token_id)
        return toks

    # -- operator level (include/aha_amd.h "operator level"): the step's kernels on caller tensors ------------------
    def linear(self, w: torch.Tensor, w_up: Optional[torch.Tensor] = None) -> "Linear":
        return Linear(self, w, w_up)

    def linear_tile(self, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = 0,
                    residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        x, w = x.contiguous(), w.contiguous()
        M, K = x.shape
        N = w.shape[0]
        out = torch.empty((M, N), dtype=torch.bfloat16, device=self.device)
        self._chk(self.lib.aha_linear_tile_forward(self.ctx, x.data_ptr(), K, M, w.data_ptr(), K, N, K,
                                                   bias.data_ptr() if bias is not None else None, int(act),
                                                   residual.data_ptr() if residual is not None else None, N, out.data_ptr(), N, _cur_stream()))
        return out

    def rmsnorm(self, x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
        x = x.contiguous()
        M, H = x.shape
        out = torch.empty_like(x)
        self._chk(self.lib.aha_rmsnorm_forward(self.ctx, x.data_ptr(), H, w.data_ptr(), out.data_ptr(), H, M, H, float(eps), _cur_stream()))
        return out

    def resid_rmsnorm(self, partial: torch.Tensor, h: torch.Tensor, w: torch.Tensor, eps: float):
        """partial fp32 [S,M,H]; h bf16 [M,H] (updated in place: h += bf16(sum_s partial)); returns the normalised rows."""
        S, M, H = partial.shape
        assert h.is_contiguous() and partial.is_contiguous()
        xn = torch.empty_like(h)
        self._chk(self.lib.aha_resid_rmsnorm_forward(self.ctx, partial.data_ptr(), S, h.data_ptr(), w.data_ptr(), xn.data_ptr(), M, H,
                                                     float(eps), _cur_stream()))
        return xn

    def heads(self, hidden: torch.Tensor):
        hidden = hidden.contiguous()
        n = hidden.shape[0]
        sc = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        raw = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        self._chk(self.lib.aha_heads_forward(self.ctx, hidden.data_ptr(), hidden.shape[1], n, sc.data_ptr(), raw.data_ptr(), _cur_stream()))
        return sc, raw

    def check_rope_rows(self, cos: torch.Tensor, sin: torch.Tensor, pos0: int, T: int):
        """cos / sin handed to Cache.update ([1,T,D] or [T,D]; a 2-D FULL table is the reference's backward-compatibility form,
        test/sink_cache.py:113-115) must be rows pos0 .. pos0+T-1 of the runtime's RoPE table: the ring re-rotates with that table."""
        tc, ts = self._rope_cpu
        for name, got, tab in (("cos", cos, tc), ("sin", sin, ts)):
            g = got.detach().to("cpu", torch.bfloat16)
            g = g[0] if g.dim() == 3 else g
            if g.dim() != 2 or g.shape[-1] != tab.shape[-1]:
                raise ValueError(f"cache_kwargs['{name}']: expected [1, T, head_dim] or [T, head_dim]")
            if g.shape[0] == T:
                want = tab[pos0:pos0 + T]
            elif g.shape[0] >= pos0 + T:                      # the full-table form
                g, want = g[:pos0 + T], tab[:pos0 + T]
            else:
                raise ValueError(f"cache_kwargs['{name}'] has {g.shape[0]} rows for {T} new tokens")
            if want.shape[0] != g.shape[0] or not torch.equal(g, want):
                raise ValueError(f"cache_kwargs['{name}'] is not the rotary table at positions get_seq_length() + arange(T): the cache "
                                 "re-rotates kept keys with the runtime's own table (set by Runtime from rope_theta)")

    def cache_update(self, stream: Stream, layer_idx: int, k: torch.Tensor, v: torch.Tensor):
        """Cache.update(key_states, value_states, layer_idx, ...) of the reference's cache classes: k, v bf16 [1,Hkv,T,D] (or
        [Hkv,T,D]); returns the (K, V) the reference's update() returns, [1,Hkv,L,D]."""
        d = self.desc
        k, v = k.reshape(d.kv_heads, -1, d.head_dim).contiguous(), v.reshape(d.kv_heads, -1, d.head_dim).contiguous()
        T = k.shape[1]
        self._chk(self.lib.aha_cache_update(self.ctx, stream.handle, int(layer_idx), k.data_ptr(), v.data_ptr(), T, None, None, _cur_stream()))
        return stream.export_kv(layer_idx)[None], stream.export_kv(layer_idx, True)[None]

    def attention(self, streams: Sequence[Stream], q: torch.Tensor, layer: int, causal_off: Optional[Sequence[int]] = None,
                  split_len: int = 0) -> torch.Tensor:
        """q bf16 [B,T,Hq*D] (rotated) over the streams' caches as they are -> bf16 [B,T,Hq*D]."""
        q = q.contiguous()
        B, T, _ = q.shape
        out = torch.empty_like(q)
        arr = (C.c_void_p * B)(*[s.handle for s in streams])
        co = (C.c_int * B)(*causal_off) if causal_off is not None else None
        self._chk(self.lib.aha_attention_forward(self.ctx, arr, B, q.data_ptr(), T, int(layer), co, int(split_len), out.data_ptr(), _cur_stream()))
        return out

    # -- vision operators (include/aha_amd.h "vision operators") -----------------------------------------------------------
    def vit_attention(self, qkv: torch.Tensor, heads: int, head_dim: int) -> torch.Tensor:
        """qkv bf16 [n,T,3*heads*head_dim] (q | k | v per row) -> softmax(q k^T / sqrt(d)) v, bf16 [n,T,heads*head_dim]."""
        qkv = qkv.contiguous()
        n, T, _ = qkv.shape
        out = torch.empty((n, T, heads * head_dim), dtype=torch.bfloat16, device=self.device)
        self._chk(self.lib.aha_vit_attention_forward(self.ctx, qkv.data_ptr(), n, T, int(heads), int(head_dim), out.data_ptr(), _cur_stream()))
        return out

    def vit_layers(self, x: torch.Tensor, n_frames: int, layer_first: int, layer_count: int = 1) -> torch.Tensor:
        """Encoder layers of the vision tower on a caller-supplied hidden state bf16 [n*tokens, Dv]."""
        x = x.contiguous()
        out = torch.empty_like(x)
        self._chk(self.lib.aha_vit_layers_forward(self.ctx, x.data_ptr(), int(n_frames), int(layer_first), int(layer_count), out.data_ptr(), _cur_stream()))
        return out

    def layernorm(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
        x = x.contiguous()
        rows, cols = x.shape
        out = torch.empty_like(x)
        self._chk(self.lib.aha_layernorm_forward(self.ctx, x.data_ptr(), cols, w.data_ptr(), b.data_ptr(), out.data_ptr(), cols, rows, cols,
                                                 float(eps), _cur_stream()))
        return out

    def vit_patchify(self, frames_u8: torch.Tensor) -> torch.Tensor:
        """uint8 [n,3,S,S] -> bf16 [n*Np, Kp] normalised patch vectors (preprocess + conv unfold; zero padding beyond 3*P*P)."""
        frames_u8 = frames_u8.contiguous()
        n = frames_u8.shape[0]
        v = self.cfg.vision
        kp = -(-(3 * v.patch_size * v.patch_size) // 64) * 64
        out = torch.empty((n * v.num_patches, kp), dtype=torch.bfloat16, device=self.device)
        cols = C.c_int(0)
        self._chk(self.lib.aha_vit_patchify_forward(self.ctx, frames_u8.data_ptr(), n, out.data_ptr(), C.byref(cols), _cur_stream()))
        assert cols.value == kp
        return out

    def pool(self, x: torch.Tensor, grid: int, out_grid: int, stride: int, mode: int) -> torch.Tensor:
        """x bf16 [n, rows >= grid^2, C] -> bf16 [n, out_grid^2, C]; mode 0 bilinear, 1 average, 2 max, 3 adaptive average."""
        x = x.contiguous()
        n, rows, ch = x.shape
        out = torch.empty((n, out_grid * out_grid, ch), dtype=torch.bfloat16, device=self.device)
        self._chk(self.lib.aha_pool_forward(self.ctx, x.data_ptr(), n, int(grid), int(out_grid), ch, int(stride), int(mode), rows, out.data_ptr(), _cur_stream()))
        return out

    def pool_gather_rows(self, x: torch.Tensor, grid: int, out_grid: int) -> torch.Tensor:
        x = x.contiguous()
        n, rows, ch = x.shape
        out = torch.empty((n, 4 * out_grid * out_grid, ch), dtype=torch.bfloat16, device=self.device)
        self._chk(self.lib.aha_pool_gather_rows_forward(self.ctx, x.data_ptr(), n, int(grid), int(out_grid), ch, rows, out.data_ptr(), _cur_stream()))
        return out

    def debug_tap(self, which: str, B: int, T: int) -> torch.Tensor:
        d = self.desc
        idx, cols = {"h": (0, d.hidden), "xn": (1, d.hidden), "q_rot": (2, d.heads * d.head_dim), "attn_out": (3, d.heads * d.head_dim),
                     "act": (4, d.inter)}[which]
        out = torch.empty((B * T, cols), dtype=torch.bfloat16, device=self.device)
        self._chk(self.lib.aha_lm_debug_tap(self.ctx, idx, out.data_ptr(), _cur_stream()))
        return out

    def last_step_work(self):
        wb, kb, fl = C.c_double(), C.c_double(), C.c_double()
        self._chk(self.lib.aha_lm_last_step_work(self.ctx, C.byref(wb), C.byref(kb), C.byref(fl)))
        return wb.value, kb.value, fl.value

    def last_gemm_time(self, kind: int = -1):
        ms, n, by = C.c_float(), C.c_int(), C.c_double()
        self._chk(self.lib.aha_lm_last_gemm_time(self.ctx, kind, C.byref(ms), C.byref(n), C.byref(by)))
        return ms.value, n.value, by.value


class Linear:
    """nn.Linear weight in the streaming (weight-read-once) layout; `w_up` makes a gate/up pair for the SwiGLU epilogue."""

    def __init__(self, rt: Runtime, w: torch.Tensor, w_up: Optional[torch.Tensor] = None):
        self.rt, self.N, self.K = rt, w.shape[0], w.shape[1]
        w = w.to(device=rt.device, dtype=torch.bfloat16).contiguous()
        wu = w_up.to(device=rt.device, dtype=torch.bfloat16).contiguous() if w_up is not None else None
        h = C.c_void_p()
        rt._chk(rt.lib.aha_linear_create(rt.ctx, w.data_ptr(), wu.data_ptr() if wu is not None else None, self.N, self.K, C.byref(h), _cur_stream()))
        self.handle, self.pairs = h, wu is not None

    def __call__(self, x: torch.Tensor, epilogue: int = _l.EPI_BF16, split_k: int = 1, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = x.contiguous()
        M = x.shape[0]
        rt = self.rt
        if epilogue == _l.EPI_SPLITK_F32:
            S = rt.lib.aha_linear_split_k(rt.ctx, self.handle, split_k)
            out = torch.empty((S, M, self.N), dtype=torch.float32, device=rt.device)
        elif epilogue == _l.EPI_F32:
            out = torch.empty((M, self.N), dtype=torch.float32, device=rt.device)
        else:
            out = torch.empty((M, self.N), dtype=torch.bfloat16, device=rt.device)
        rt._chk(rt.lib.aha_linear_forward(rt.ctx, self.handle, x.data_ptr(), x.shape[1], M, int(epilogue), int(split_k),
                                          bias.data_ptr() if bias is not None else None, out.data_ptr(), self.N, _cur_stream()))
        return out

    def close(self):
        if self.handle is not None:
            self.rt.lib.aha_linear_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
