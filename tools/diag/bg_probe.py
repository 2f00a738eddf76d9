#!/usr/bin/env python3
"""What delays LM launches underneath a background encode?  LM steps alone, then the same steps while a second stream encodes
32 frames in a loop, for: graph replay on/off, background grid size (256 / 64 / 16 workgroups), stream priorities.
python tools/diag/bg_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights, make_frames
from aha_amd.runtime import Runtime
cfg = preset("bench")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=320, max_vit_frames=32, max_positions=cfg.lm.max_position_embeddings)
H, tf = cfg.lm.hidden_size, cfg.frame_num_tokens
st = rt.open_stream("static", 2048, 32, capacity=cfg.lm.max_position_embeddings)
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(1, 20, H, device="cuda", generator=g) * 0.02).bfloat16()
rt.lm_step([st], q)
emb = (torch.randn(1, tf, H, device="cuda", generator=g) * 0.02).bfloat16()
fr = make_frames(32, cfg.vision.image_size, seed=1).cuda()
out = torch.empty((32 * tf, H), dtype=torch.bfloat16, device="cuda")
NS = 64
def lm_alone(stream):
    with torch.cuda.stream(stream):
        for _ in range(8): rt.lm_step([st], emb)
        stream.synchronize(); t = time.perf_counter()
        for _ in range(NS): rt.lm_step([st], emb)
        stream.synchronize()
    return (time.perf_counter() - t) / NS * 1e3
def overlapped(lm_stream, vit_stream, bg, n_enc):
    """n_enc encodes enqueued on vit_stream, then NS LM steps on lm_stream; returns (ms per LM step, ms until the encodes are done)"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rt.set_tuning("tower_bg", bg)
    with torch.cuda.stream(vit_stream):
        e0.record(vit_stream)
        for _ in range(n_enc): rt.visual_embed(fr, out=out)
        e1.record(vit_stream)
    rt.set_tuning("tower_bg", 0)
    with torch.cuda.stream(lm_stream):
        t = time.perf_counter()
        for _ in range(NS): rt.lm_step([st], emb)
        lm_stream.synchronize()
        lm = (time.perf_counter() - t) / NS * 1e3
    torch.cuda.synchronize()
    return lm, e0.elapsed_time(e1) / n_enc
hi, lo = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
modes = [(1, 0), (0, 0)] if len(sys.argv) < 2 else [(1, int(v)) for v in sys.argv[1].split(",")]
for graph, fuse in modes:
    rt.set_tuning("use_graph", graph); rt.set_tuning("fuse_mlp", fuse % 10); rt.set_tuning("wpb_gateup", 8 if fuse else 5)
    print(f"use_graph={graph} fuse_mlp={fuse}: LM step alone {lm_alone(hi):.3f} ms", flush=True)
    for bg, cus in ((0, 256), (1, 256), (1, 64), (1, 16)):
        rt.set_tuning("bg_cus", cus)
        n_enc = 6 if bg == 0 else 4
        lm, enc = overlapped(hi, lo, bg, n_enc)
        print(f"  tower_bg={bg} bg_cus={cus:3d}: LM step {lm:.3f} ms while encoding; encode {enc:.1f} ms each (LM phase {lm * NS:.0f} ms, encodes {enc * n_enc:.0f} ms)", flush=True)
rt.set_tuning("bg_cus", 256)
