#!/usr/bin/env python3
"""Vision encode time per frame by batch size (tile quantisation of the tower GEMMs): python tools/diag/vit_batch.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights, make_frames
from aha_amd.runtime import Runtime
cfg = preset("bench")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
for nmax in (32, 64, 128):
    rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=nmax)
    fr = make_frames(nmax, cfg.vision.image_size, seed=1).cuda()
    for epi in (0, 1, 0, 1):
        rt.set_tuning("tile_epi", epi)
        for _ in range(2): rt.visual_embed(fr)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): rt.visual_embed(fr)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(f"batch {nmax} tile_epi={epi}: {dt*1e3:.2f} ms = {dt*1e3/nmax*32:.2f} ms per 32 frames", flush=True)
    rt.close()
