"""Test helper: a stand-in for aha_amd.runtime.Runtime whose arithmetic is the ORACLE on CPU, so the
host logic of the drivers (queues, prompts, trigger rules, rounding, sharding) can be tested
without a GPU.  Lives under tests/ - the product never imports it."""
import torch

from oracle import frame_ingest as oracle_ingest
from oracle.cache_policies import make_policy
from oracle.qwen2_live import OracleLM, frame_scores
from oracle.vision_tower import OracleVision


class _FakeStream:
    def __init__(self, alt, W, S):
        self.alt, self.W, self.S = alt, W, S
        self.handle = object()
        self.reset()

    def reset(self):
        self.pol = make_policy(self.alt, self.W, self.S)

    def get_seq_length(self, layer_idx=0):
        return self.pol.get_seq_length()

    def close(self):
        self.handle = None


class OracleBackedRuntime:
    def __init__(self, cfg, weights, dtype=torch.float32):
        self.cfg, self.device = cfg, torch.device("cpu")
        self.hidden_size, self.frame_num_tokens = cfg.lm.hidden_size, cfg.frame_num_tokens
        self.lm, self.vis, self.dtype = OracleLM(cfg.lm, weights, dtype), OracleVision(cfg, weights, dtype), dtype
        self._last = None

    def open_stream(self, alt_cache="default_sink", window_length=2048, num_sink_tokens=32, capacity=None, attn_semantics="trailing"):
        return _FakeStream(alt_cache, window_length, num_sink_tokens)

    RESIZE_PIL_BICUBIC, RESIZE_CV2_LINEAR = 0, 1

    def frame_ingest(self, frame_hwc_u8, *, bgr=False, method=0, out=None):
        f = frame_hwc_u8.cpu().numpy()
        S = self.cfg.vision.image_size
        if method == self.RESIZE_PIL_BICUBIC:
            canvas = oracle_ingest.demo_frame_to_canvas(f[:, :, ::-1] if bgr else f, S)
        else:
            canvas = oracle_ingest.benchmark_frame_to_canvas(f if bgr else f[:, :, ::-1], S)
        canvas = torch.from_numpy(canvas.copy())
        if out is not None:
            out.copy_(canvas)
            return out
        return canvas

    def visual_embed(self, frames_u8):
        return self.vis.visual_embed(frames_u8)

    def embed_tokens(self, ids):
        return self.lm.embed_tokens(ids.view(-1))

    def lm_step(self, streams, embeds, want_raw=False, want_hidden=False):
        outs = [self.lm.step(embeds[b:b + 1].to(self.dtype), s.pol, want_logits=True) for b, s in enumerate(streams)]
        self._last = outs
        return torch.cat([frame_scores(o) for o in outs], 0)

    def logits_last(self, B, want_logits=True):
        lg = torch.cat([o["logits"][:, -1] for o in self._last], 0)
        return (lg if want_logits else None), lg.argmax(-1)

    def generate_greedy(self, stream, first_ids, max_new_tokens, eos_token_id, repetition_penalty=None, generated_token_ids=None,
                        chunk=None, between_chunks=None):
        """fast_greedy_generate (models/modeling_live.py:64-90) on the oracle; `chunk`: the same response produced in several
        calls, each continuing from the previous chunk's last id (what the product's chunked generation does)."""
        if chunk:
            out, ids = [], first_ids
            while len(out) < max_new_tokens:
                part = self.generate_greedy(stream, ids, min(chunk, max_new_tokens - len(out)), eos_token_id, repetition_penalty, generated_token_ids)
                out += part
                if out[-1] == eos_token_id:
                    break
                ids = torch.tensor([[out[-1]]])
                if between_chunks is not None and len(out) < max_new_tokens:
                    between_chunks()
            return out
        emb = self.embed_tokens(first_ids).view(1, -1, self.hidden_size)
        out = []
        for _ in range(max_new_tokens):
            self.lm_step([stream], emb)
            logits, am = self.logits_last(1)
            if repetition_penalty is not None:
                if generated_token_ids:
                    idx = torch.tensor(generated_token_ids)[None]
                    sc = torch.gather(logits, 1, idx)
                    sc = torch.where(sc < 0, sc * repetition_penalty, sc / repetition_penalty)
                    logits = logits.scatter(1, idx, sc)
                tok = int(logits.argmax(-1).item())
                if tok != eos_token_id and generated_token_ids is not None:
                    generated_token_ids.append(tok)
            else:
                tok = int(am.item())
            out.append(tok)
            if tok == eos_token_id:
                break
            emb = self.embed_tokens(torch.tensor([[tok]])).view(1, 1, self.hidden_size)
        return out
