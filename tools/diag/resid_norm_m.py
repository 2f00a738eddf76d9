#!/usr/bin/env python3
"""resid_norm_kernel (split-K reduce + residual + RMSNorm) vs row count, S = 8 slabs, H = 3584: where does M = 288 stand between
latency-bound and bandwidth-bound?  Slabs rotate over several buffers (cold: > L2).  python tools/diag/resid_norm_m.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime

cfg = preset("tiny")
rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16))
H, S = 3584, 8
w = torch.ones(H, dtype=torch.bfloat16, device="cuda")
for M in (36, 72, 144, 224, 256, 272, 288, 320, 512, 576):
    bufs = [torch.randn(S, M, H, device="cuda") for _ in range(6)]
    h = torch.randn(M, H, device="cuda").bfloat16()
    for i in range(6):
        rt.resid_rmsnorm(bufs[i % 6], h, w, 1e-6)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 60
    for i in range(n):
        rt.resid_rmsnorm(bufs[i % 6], h, w, 1e-6)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    mb = (S * M * H * 4 + M * H * 6) / 1e6
    print(f"M={M:4d}: {us:6.2f} us per launch ({mb:6.1f} MB -> {mb / us / 1e3 * 1e3:6.2f} TB/s incl. launch gaps)", flush=True)
