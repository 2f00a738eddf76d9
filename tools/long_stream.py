#!/usr/bin/env python3
"""BASELINE.json configs[2]: a long synthetic stream through the sliding-window / attention-sink cache at
full model size.  Frames come from a counter-based generator on the device (no 3.4 GB host buffer);
the run is repeated to check bit-reproducibility.  python tools/long_stream.py [--frames 10000]
(The oracle replay of the stream's first frames lives in tests/long_stream_oracle_prefix.py: only tests/ may touch oracle/.)"""
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
a = ap.parse_args()
cfg = preset(a.preset); tf, H, S = cfg.frame_num_tokens, cfg.lm.hidden_size, cfg.vision.image_size
w_dev = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w_dev, max_step_tokens=128, max_vit_frames=32)
del w_dev
torch.cuda.empty_cache()

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
        for j in range(n):
            x = emb[j:j + 1] if i0 + j else torch.cat([pre, emb[:1]], 1)
            scores[i0 + j] = rt.lm_step([st], x.contiguous())[0]
        if i0 % 2048 == 0:
            print(f"rep {rep} frame {i0} seq_len {st.get_seq_length()}", flush=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sc = scores.cpu()
    digests.append(hashlib.sha256(sc.numpy().tobytes()).hexdigest())
    print(f"rep {rep}: {a.frames} frames in {dt:.1f}s = {a.frames/dt:.1f} frames/s; finite={bool(torch.isfinite(sc).all())} "
          f"seq_len={st.get_seq_length()} seen={st.seen_tokens} expected_seen={20 + 35 + a.frames * tf} "
          f"score ranges info[{sc[:,0].min():.3f},{sc[:,0].max():.3f}] rel[{sc[:,1].min():.3f},{sc[:,1].max():.3f}] sha256={digests[-1][:16]}")
    st.close()
print("bit-reproducible across runs:", len(set(digests)) == 1)
