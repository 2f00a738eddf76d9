#!/usr/bin/env python3
"""Split-K factors of the weight-streaming GEMMs vs LM step time at B = 1 and B = 8 (SinkCache W=2048, steady state).
python tools/diag/split_sweep.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime

cfg = preset("bench"); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=640, max_vit_frames=8)
del w; torch.cuda.empty_cache()
g = torch.Generator(device="cuda").manual_seed(0)
NAMES = ["qkv", "o", "gate_up", "down", "attn", "rerot"]


def kinds(sts, x):
    out = {}
    for k, name in enumerate(NAMES[:4]):
        rt.set_tuning("time_gemm", 1 << k)
        ms = n = 0
        for i in range(4):
            rt.lm_step(sts, x); torch.cuda.synchronize()
            if i:
                m, c, _ = rt.last_gemm_time(k); ms += m; n += c
        out[name] = ms / max(n, 1) * 1e3
    rt.set_tuning("time_gemm", 0)
    return out


def step_ms(sts, x, n=30):
    for _ in range(5):
        rt.lm_step(sts, x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        rt.lm_step(sts, x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for B in (1, 8):
    sts = [rt.open_stream("default_sink", 2048, 32) for _ in range(B)]
    x = (torch.randn(B, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
    for _ in range(60):
        rt.lm_step(sts, x)
    for sq, so, sd in ((0, 0, 0), (3, 4, 4), (4, 4, 4), (7, 4, 8), (7, 8, 4), (5, 6, 6), (2, 2, 2), (0, 0, 0)):
        rt.set_tuning("split_qkv", sq); rt.set_tuning("split_o", so); rt.set_tuning("split_down", sd)
        k = kinds(sts, x)
        print(f"B={B} split qkv/o/down {sq}/{so}/{sd}: step {step_ms(sts, x):.3f} ms; " + "  ".join(f"{a} {b:.1f}" for a, b in k.items()), flush=True)
    for s in sts:
        s.close()
