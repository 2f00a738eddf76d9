#!/usr/bin/env python3
"""BASELINE.json configs[2]: a long synthetic stream through the sliding-window / attention-sink cache at
full model size.  Frames come from a counter-based generator on the device (no 3.4 GB host buffer);
the run is repeated to check bit-reproducibility, and with --oracle-frames N the first N frames of the stream
(N > 56 crosses the first evictions and re-rotations at W=2048, sink=32) are replayed through the oracle (test
infrastructure, CPU) on the same embeddings: |HIP - oracle_fp32| must stay within max(1e-3, 2 * |oracle_bf16 -
oracle_fp32|), the reference's own bf16 noise.  python tools/long_stream.py [--frames 10000] [--oracle-frames 72]"""
import argparse, hashlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights, make_token_ids
from aha_amd.runtime import Runtime

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=10000); ap.add_argument("--cache", default="default_sink")
ap.add_argument("--repeat", type=int, default=2); ap.add_argument("--preset", default="bench")
ap.add_argument("--oracle-frames", type=int, default=0)
a = ap.parse_args()
cfg = preset(a.preset); tf, H, S = cfg.frame_num_tokens, cfg.lm.hidden_size, cfg.vision.image_size
w_dev = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w_dev, max_step_tokens=128, max_vit_frames=32)
w_cpu = {k: v.cpu() for k, v in w_dev.items() if not k.startswith("vision.")} if a.oracle_frames else None
del w_dev
torch.cuda.empty_cache()
kept_emb = []

def frames_batch(i0, n):           # counter-based: frame i depends only on (seed 0, i)
    g = torch.Generator(device="cuda"); out = []
    for i in range(i0, i0 + n):
        g.manual_seed(i); out.append(torch.randint(0, 256, (3, S, S), generator=g, device="cuda", dtype=torch.uint8))
    return torch.stack(out)

digests = []
for rep in range(a.repeat):
    st = rt.open_stream(a.cache, 2048, 32)
    rt.lm_step([st], rt.embed_tokens(make_token_ids(20, cfg.lm.vocab_size, seed=101)).view(1, -1, H))
    pre = rt.embed_tokens(make_token_ids(35, cfg.lm.vocab_size, seed=100)).view(1, -1, H)
    scores = torch.empty((a.frames, 3), device="cuda")
    t0 = time.perf_counter()
    for i0 in range(0, a.frames, 32):
        n = min(32, a.frames - i0)
        emb = rt.visual_embed(frames_batch(i0, n)).view(n, tf, H)
        if rep == 0 and i0 < a.oracle_frames:
            kept_emb.append(emb[:max(0, min(n, a.oracle_frames - i0))].cpu())
        for j in range(n):
            x = emb[j:j + 1] if i0 + j else torch.cat([pre, emb[:1]], 1)
            scores[i0 + j] = rt.lm_step([st], x.contiguous())[0]
        if i0 % 2048 == 0:
            print(f"rep {rep} frame {i0} seq_len {st.get_seq_length()}", flush=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sc = scores.cpu()
    digests.append(hashlib.sha256(sc.numpy().tobytes()).hexdigest())
    if rep == 0:
        first_scores = sc
    print(f"rep {rep}: {a.frames} frames in {dt:.1f}s = {a.frames/dt:.1f} frames/s; finite={bool(torch.isfinite(sc).all())} "
          f"seq_len={st.get_seq_length()} seen={st.seen_tokens} expected_seen={20 + 35 + a.frames * tf} "
          f"score ranges info[{sc[:,0].min():.3f},{sc[:,0].max():.3f}] rel[{sc[:,1].min():.3f},{sc[:,1].max():.3f}] sha256={digests[-1][:16]}")
    st.close()
print("bit-reproducible across runs:", len(set(digests)) == 1)

if a.oracle_frames:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.cache_policies import make_policy
    from oracle.qwen2_live import OracleLM, frame_scores
    torch.set_num_threads(min(16, torch.get_num_threads()))
    emb = torch.cat(kept_emb)[:a.oracle_frames]
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
