#!/usr/bin/env python3
"""Attention key-split length vs time of the attention launch group, SinkCache W=2048 at steady state, B = 1, 2, 4, 8.
python tools/diag/attn_split_sweep.py"""
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


def attn_us(sts, x):
    rt.set_tuning("time_gemm", 1 << 4)
    ms = n = 0
    for i in range(5):
        rt.lm_step(sts, x); torch.cuda.synchronize()
        if i:
            m, c, _ = rt.last_gemm_time(4); ms += m; n += c
    rt.set_tuning("time_gemm", 0)
    return ms / max(n, 1) * 1e3


def step_ms(sts, x, n=20):
    for _ in range(3):
        rt.lm_step(sts, x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        rt.lm_step(sts, x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


LENS = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (0, 192, 256, 320, 384, 512, 1088)
BS = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 4, 8)
for B in BS:
    sts = [rt.open_stream("default_sink", 2048, 32) for _ in range(B)]
    x = (torch.randn(B, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
    for _ in range(60):
        rt.lm_step(sts, x)
    for mode in (1, 2):
        rt.set_tuning("attn_lm", mode)
        row = []
        for sl in LENS:
            rt.set_tuning("attn_split_len", sl)
            row.append(f"{sl}: {attn_us(sts, x):.1f}")
        print(f"B={B} attn_lm={mode}  us per layer by split_len  " + "  ".join(row), flush=True)
    rt.set_tuning("attn_lm", 1); rt.set_tuning("attn_split_len", 0)
    for s in sts:
        s.close()
