#!/usr/bin/env python3
"""Cost of a second active HIP queue per LM launch, by which pair of streams is used: LM steps on stream i while a 16-workgroup
background encode runs on stream j (the background load is negligible: what is measured is the queue interaction)."""
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
rt.lm_step([st], (torch.randn(1, 20, H, device="cuda", generator=g) * 0.02).bfloat16())
emb = (torch.randn(1, tf, H, device="cuda", generator=g) * 0.02).bfloat16()
fr = make_frames(32, cfg.vision.image_size, seed=1).cuda()
out = torch.empty((32 * tf, H), dtype=torch.bfloat16, device="cuda")
NS = 48
streams = [torch.cuda.Stream(priority=0) for _ in range(6)] + [torch.cuda.Stream(priority=-1) for _ in range(3)]
def lm(stream):
    with torch.cuda.stream(stream):
        t = time.perf_counter()
        for _ in range(NS): rt.lm_step([st], emb)
        stream.synchronize()
    return (time.perf_counter() - t) / NS * 1e3
for s in (streams[0], streams[6]):
    lm(s); print(f"alone on {'hi' if s is streams[6] else 'lo'}-priority stream: {lm(s):.3f} ms", flush=True)
rt.set_tuning("bg_cus", 16)
for i, j in ((0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (6, 0), (6, 1), (6, 2), (6, 3), (7, 0), (8, 1), (1, 0), (2, 0)):
    torch.cuda.synchronize()
    rt.set_tuning("tower_bg", 1)
    with torch.cuda.stream(streams[j]):
        rt.visual_embed(fr, out=out)                     # ~390 ms on 16 workgroups
    rt.set_tuning("tower_bg", 0)
    v = lm(streams[i])
    torch.cuda.synchronize()
    print(f"LM on stream {i} ({'hi' if i >= 6 else 'lo'}), background on stream {j}: LM step {v:.3f} ms", flush=True)
