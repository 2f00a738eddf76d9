#!/usr/bin/env python3
"""BASELINE.json configs[2], parity leg (a script, not a pytest case: it takes about two minutes of host CPU): the first N
frames of the long synthetic stream (N > 56 crosses the first evictions and re-rotations at W=2048, sink=32) are scored on
the GPU and replayed through the oracle on the same embeddings.  |HIP - oracle_fp32| is reported against |oracle_bf16 -
oracle_fp32| per score (median / p95 / max) and must stay within max(1e-3, 2 x the band's maximum).
    python tests/long_stream_oracle_prefix.py [--frames 80] [--cache default_sink]
Output of the last run: profiles/r01_long_stream_oracle_prefix.txt."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights, make_token_ids
from aha_amd.runtime import Runtime
from oracle.cache_policies import make_policy
from oracle.qwen2_live import OracleLM, frame_scores

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=80); ap.add_argument("--cache", default="default_sink"); ap.add_argument("--preset", default="bench")
a = ap.parse_args()
cfg = preset(a.preset); tf, H, S = cfg.frame_num_tokens, cfg.lm.hidden_size, cfg.vision.image_size
w_dev = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w_dev, max_step_tokens=128, max_vit_frames=32)
w_cpu = {k: v.cpu() for k, v in w_dev.items() if not k.startswith("vision.")}
del w_dev
torch.cuda.empty_cache()


def frames_batch(i0, n):           # the same counter-based frames as tools/long_stream.py
    g = torch.Generator(device="cuda"); out = []
    for i in range(i0, i0 + n):
        g.manual_seed(i); out.append(torch.randint(0, 256, (3, S, S), generator=g, device="cuda", dtype=torch.uint8))
    return torch.stack(out)


st = rt.open_stream(a.cache, 2048, 32)
rt.lm_step([st], rt.embed_tokens(make_token_ids(20, cfg.lm.vocab_size, seed=101)).view(1, -1, H))
pre_dev = rt.embed_tokens(make_token_ids(35, cfg.lm.vocab_size, seed=100)).view(1, -1, H)
scores, kept = [], []
for i0 in range(0, a.frames, 32):
    n = min(32, a.frames - i0)
    emb = rt.visual_embed(frames_batch(i0, n)).view(n, tf, H)
    kept.append(emb.cpu())
    for j in range(n):
        x = emb[j:j + 1] if i0 + j else torch.cat([pre_dev, emb[:1]], 1)
        scores.append(rt.lm_step([st], x.contiguous()).cpu())
first_scores, emb = torch.cat(scores), torch.cat(kept)

torch.set_num_threads(min(16, torch.get_num_threads()))
ob, o32 = OracleLM(cfg.lm, w_cpu, torch.bfloat16), OracleLM(cfg.lm, w_cpu, torch.float32)
cb, c32 = make_policy(a.cache, 2048, 32), make_policy(a.cache, 2048, 32)
q = ob.embed_tokens(make_token_ids(20, cfg.lm.vocab_size, seed=101)).view(1, -1, H)
pre = ob.embed_tokens(make_token_ids(35, cfg.lm.vocab_size, seed=100)).view(1, -1, H)
ob.step(q, cb); o32.step(q.float(), c32)


def rel(sx):
    return torch.stack([sx[:, 0], sx[:, 1], torch.log(sx[:, 2])], -1)


d32 = band = 0.0
dev_hip, dev_bf = [], []
t0 = time.perf_counter()
for i in range(emb.shape[0]):
    x = emb[i:i + 1] if i else torch.cat([pre, emb[:1]], 1)
    sb, s32 = rel(frame_scores(ob.step(x, cb))), rel(frame_scores(o32.step(x.float(), c32)))
    gs = rel(first_scores[i:i + 1])
    d32, band = max(d32, (gs - s32).abs().max().item()), max(band, (sb - s32).abs().max().item())
    dev_hip.append((gs - s32).abs()[0]); dev_bf.append((sb - s32).abs()[0])
    if i % 16 == 0:
        print(f"oracle frame {i}: seq_len {cb.get_seq_length()} running |hip-fp32| {d32:.2e} band {band:.2e} ({time.perf_counter() - t0:.0f}s)", flush=True)
dh, db = torch.stack(dev_hip), torch.stack(dev_bf)
for c, name in enumerate(("informative (prob.)", "relevance (prob.)", "log uncertainty")):
    print(f"  {name:20s} |HIP - fp32| median {dh[:, c].median():.2e} p95 {dh[:, c].quantile(0.95):.2e} max {dh[:, c].max():.2e}   "
          f"|oracle_bf16 - fp32| median {db[:, c].median():.2e} p95 {db[:, c].quantile(0.95):.2e} max {db[:, c].max():.2e}")
ok = d32 <= max(1e-3, 2.0 * band)
print(f"oracle parity on the first {emb.shape[0]} frames ({a.cache}, W=2048, sink=32, evicting from frame ~56): "
      f"max |HIP - oracle_fp32| = {d32:.3e}, oracle bf16 band = {band:.3e} -> {'PASS' if ok else 'FAIL'}")
sys.exit(0 if ok else 1)
