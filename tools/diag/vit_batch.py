#!/usr/bin/env python3
"""Vision encode time per 32 frames by batch size and tile-kernel selection (tuning tile_p288 / attn_head), interleaved rounds in
one process:  python tools/diag/vit_batch.py [batches]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights, make_frames
from aha_amd.runtime import Runtime
cfg = preset(sys.argv[2] if len(sys.argv) > 2 else "bench")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
batches = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 128, 8]
for nmax in batches:
    rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=nmax)
    fr = make_frames(nmax, cfg.vision.image_size, seed=1).cuda()
    ref = None
    combos = ((0, 0), (1, 0), (0, 1), (1, 1), (0, 0), (1, 1))
    if len(sys.argv) > 3: combos = tuple((int(v), 1) for v in sys.argv[3].split(","))
    for p288, head in combos:
        rt.set_tuning("tile_p288", p288); rt.set_tuning("attn_head", head)
        for _ in range(2): out = rt.visual_embed(fr)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): rt.visual_embed(fr)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        if ref is None: ref = out.clone()
        print(f"batch {nmax:4d} tile_p288={p288} attn_head={head}: {dt*1e3:7.2f} ms = {dt*1e3/nmax*32:6.2f} ms per 32 frames  same bits as first: {bool(torch.equal(ref, out))}", flush=True)
    rt.close()
