#!/usr/bin/env python3
"""Background tower (tuning tower_bg) against the default tower: same bits, time when run alone."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights, make_frames
from aha_amd.runtime import Runtime
cfg = preset("bench")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
for n in (32, 1, 3):
    rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=n)
    fr = make_frames(n, cfg.vision.image_size, seed=1).cuda()
    ref = None
    for bg in (0, 1, 2, 0, 1):
        rt.set_tuning("tower_bg", bg)
        for _ in range(2): out = rt.visual_embed(fr)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): rt.visual_embed(fr)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        if ref is None: ref = out.clone()
        print(f"frames {n:3d} tower_bg={bg}: {dt*1e3:7.2f} ms  same bits as the default tower: {bool(torch.equal(ref, out))}", flush=True)
    rt.set_tuning("tower_bg", 0)
    rt.close()
