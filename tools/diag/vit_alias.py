#!/usr/bin/env python3
"""One-frame (and n-frame) vision encode with the tower's weights cache-resident: tuning "vit_alias" = k makes encoder layer l run
layer l % k's weights, so k layers' weights (25 MB each at ViT-L) are re-read from the Infinity Cache / L2 instead of HBM.  The
difference against vit_alias = 0 is what a perfect weight prefetch could buy on the latency path.  Embeddings are wrong on purpose.
    python tools/diag/vit_alias.py [frames]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import LiveConfig, LMConfig
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime
ns = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1]
cfg = LiveConfig(lm=LMConfig(num_hidden_layers=1, vocab_size=1024), name="vit24")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=max(max(ns), 4)); del w
# something to evict the caches between encodes, as the LM step does in the real loop (13 GB of weights per step)
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for n in ns:
    fr = make_frames(n, cfg.vision.image_size, seed=0).cuda()
    for alias in (0, 1, 2, 4, 0, 1):
        rt.set_tuning("vit_alias", alias)
        for flush in (False, True):
            ts = []
            for i in range(12):
                if flush:
                    junk.add_(1)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); rt.visual_embed(fr); e1.record(); e1.synchronize()
                if i >= 4:
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            print(f"{n} frame(s) vit_alias={alias} flush={int(flush)}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}", flush=True)
