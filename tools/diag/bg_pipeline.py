#!/usr/bin/env python3
"""bench.py's Workload (encode of batch k+1 on a second stream underneath the LM steps of batch k) outside bench.py:
serial / overlapped with the default tower / overlapped with the background tower, LM stream at high priority."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aha_amd, bench
from aha_amd.config import preset
from aha_amd.synth import make_weights, make_frames
from aha_amd.runtime import Runtime
cfg = preset("bench")
dev = torch.device("cuda:0")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=320, max_vit_frames=32, max_positions=cfg.lm.max_position_embeddings)
F = 32
frames = make_frames(F, cfg.vision.image_size, seed=0).to(dev)
prefix = torch.arange(100, 135, device=dev); query = torch.arange(200, 220, device=dev)
hi, lo = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
cur = torch.cuda.current_stream()
for name, main, vit, bg in (("serial, current stream", cur, cur, 0), ("serial, priority stream", hi, hi, 0),
                            ("overlap, default tower", hi, lo, 0), ("overlap, background tower", hi, lo, 1),
                            ("overlap, background tower (non-persistent tiles)", hi, lo, 2)):
    wl = bench.Workload(rt, cfg, dev, 1, F, "static", 2048, 32, frames, prefix, query, main, vit, None, tower_bg=bool(bg))
    if bg == 2:
        orig = rt.set_tuning
        def st(k, v, orig=orig): return orig(k, 2 if (k == "tower_bg" and v == 1) else v)
        rt.set_tuning = st
    wl.run(2); torch.cuda.synchronize()
    t = time.perf_counter(); wl.run(6); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 6
    if bg == 2: rt.set_tuning = orig
    print(f"{name:50s}: {dt * 1e3:7.2f} ms per step = {F / dt:6.1f} frames/s", flush=True)
    wl.close()
