#!/usr/bin/env python3
"""Vision encode time by frame count (HIP events, caches flushed between encodes), for A/B runs of two library builds in one gpurun call:
    AHA_AMD_LIB=/path/to/other.so python tools/diag/vit_time.py [frames,...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import LiveConfig, LMConfig
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime
ns = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 7, 8, 32]
cfg = LiveConfig(lm=LMConfig(num_hidden_layers=1, vocab_size=1024), name="vit24")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=max(ns)); del w
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
out = []
for n in ns:
    fr = make_frames(n, cfg.vision.image_size, seed=0).cuda()
    ts = []
    for i in range(12):
        junk.add_(1); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); emb = rt.visual_embed(fr); e1.record(); e1.synchronize()
        if i >= 4:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    out.append(f"{n}f {ts[len(ts)//2]:.3f}ms")
    torch.save(emb.cpu(), f"/tmp/vit_emb_{n}_{os.path.basename(os.environ.get('AHA_AMD_LIB', 'default'))}.pt")
print(os.path.basename(os.environ.get("AHA_AMD_LIB", "libaha_amd.so (default)")), " ".join(out), flush=True)
