#!/usr/bin/env python3
"""Timing ablations of attn_lm_kernel (results are wrong on purpose): which of LDS-DMA refills (1), v_exp (2), S MFMAs (4),
P V MFMAs (8), the block barrier (16), the partial-result stores (32), the Q loads (64) the per-block time is made of.  SinkCache W=2048 steady state.  python tools/diag/attn_ablate.py"""
import os, sys
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


for B in (1, 8):
    sts = [rt.open_stream("default_sink", 2048, 32) for _ in range(B)]
    x = (torch.randn(B, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
    for _ in range(60):
        rt.lm_step(sts, x)
    for sl in (256, 1088):
        rt.set_tuning("attn_split_len", sl)
        row = []
        for abl in (0, 1, 15, 31, 32, 64, 96, 63, 127):
            rt.set_tuning("attn_lm", 2 | (abl << 4))
            row.append(f"{abl}: {attn_us(sts, x):.1f}")
        print(f"B={B} split_len={sl}  us per layer (attn_lm + combine) by ablation mask  " + "  ".join(row), flush=True)
    rt.set_tuning("attn_lm", 1); rt.set_tuning("attn_split_len", 0)
    for s in sts:
        s.close()
