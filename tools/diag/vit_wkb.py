#!/usr/bin/env python3
"""Vision encode on the throughput path with the tile-GEMM weights read row-major vs from their k-blocked twins (tuning "tile_wkb"),
interleaved in one process; embeddings must not move by a bit.     python tools/diag/vit_wkb.py [frames,...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import LiveConfig, LMConfig
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime
ns = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 8]
keys = sys.argv[2].split(",") if len(sys.argv) > 2 else ["tile_wkb"]
cfg = LiveConfig(lm=LMConfig(num_hidden_layers=1, vocab_size=1024), name="vit24")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=max(ns)); del w


def med(fr):
    ts = []
    for i in range(12):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rt.visual_embed(fr); e1.record(); e1.synchronize()
        if i >= 4:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for n in ns:
    fr = make_frames(n, cfg.vision.image_size, seed=0).cuda()
    for k in keys:
        rt.set_tuning(k, 0)
    ref = rt.visual_embed(fr).clone()
    for rnd in range(3):
        for on in (0, 1):
            for k in keys:
                rt.set_tuning(k, on)
            same = torch.equal(rt.visual_embed(fr), ref)
            m, lo = med(fr)
            print(f"{n} frames {'+'.join(keys)}={on}: median {m:.3f} ms  min {lo:.3f}  bits {'same' if same else 'DIFFER'}", flush=True)
