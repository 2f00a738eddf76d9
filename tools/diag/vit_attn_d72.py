#!/usr/bin/env python3
"""so400m-geometry dense attention (729 keys, 16 heads of 72 channels), 96-wide template vs 128-wide padding, interleaved rounds:
python tools/diag/vit_attn_d72.py [frames]"""
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
qkv = torch.randn(n, 729, 3 * 1152, generator=g, device="cuda").bfloat16()
flops = 4.0 * 729 * 729 * 72 * 16 * n
res = {0: [], 1: [], 2: []}
for rnd in range(5):
    for mode in (0, 1, 2):
        rt.set_tuning("attn_d96", min(mode, 1)); rt.set_tuning("attn_tpw", 2 if mode == 2 else 0)
        for _ in range(3):
            rt.vit_attention(qkv, 16, 72)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            rt.vit_attention(qkv, 16, 72)
        b.record(); torch.cuda.synchronize()
        res[mode].append(a.elapsed_time(b) / 20 * 1e3)
for mode, name in ((0, "attn_dense_kernel<128,1> (72 -> 128)"), (1, "attn_dense_kernel<96,1> (72 -> 96)"), (2, "attn_dense_kernel<96,2>")):
    v = sorted(res[mode])
    print(f"{name:40s} median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f} us  = {flops / (v[len(v)//2] * 1e-6) / 1e15:.3f} PFLOP/s at {n} frames")
rt.set_tuning("attn_tpw", 0)
o0 = (rt.set_tuning("attn_d96", 0), rt.vit_attention(qkv, 16, 72))[1]
o1 = (rt.set_tuning("attn_d96", 1), rt.vit_attention(qkv, 16, 72))[1]
print("bit-identical:", bool(torch.equal(o0, o1)))
