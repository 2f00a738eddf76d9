#!/usr/bin/env python3
"""Dense (vision tower) attention per 32-frame layer, restaging kernel vs head-resident kernel, interleaved rounds in one process:
python tools/diag/vit_attn_time.py [frames]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = preset("tiny")
rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16), max_step_tokens=64, max_vit_frames=1, max_positions=256)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(n, 576, 3 * 1024, generator=g, device="cuda").bfloat16()
flops = 4.0 * 576 * 576 * 64 * 16 * n
res = {0: [], 2: []}
for rnd in range(5):
    for mode in (0, 2):
        rt.set_tuning("attn_head", mode)
        for _ in range(3):
            rt.vit_attention(qkv, 16, 64)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            rt.vit_attention(qkv, 16, 64)
        b.record(); torch.cuda.synchronize()
        res[mode].append(a.elapsed_time(b) / 20 * 1e3)
for mode, name in ((0, "attn_dense_kernel<64,1> (restaging)"), (2, "attn_head64_kernel<12,3> (head LDS-resident)")):
    v = sorted(res[mode])
    print(f"{name:48s} median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f} us  = {flops / (v[len(v)//2] * 1e-6) / 1e15:.3f} PFLOP/s at {n} frames")
o0 = (rt.set_tuning("attn_head", 0), rt.vit_attention(qkv, 16, 64))[1]
o2 = (rt.set_tuning("attn_head", 2), rt.vit_attention(qkv, 16, 64))[1]
print("bit-identical:", bool(torch.equal(o0, o2)))
