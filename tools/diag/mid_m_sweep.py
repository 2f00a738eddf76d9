#!/usr/bin/env python3
"""Per-kind kernel times of ONE LM step at steady state on SinkCache(W=2048): B=1 and B=8, mid-M kernel on/off, attention
key-split sweep.  python tools/diag/mid_m_sweep.py"""
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


def fill(B):
    sts = [rt.open_stream("default_sink", 2048, 32) for _ in range(B)]
    x = (torch.randn(B, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
    for _ in range(60):
        rt.lm_step(sts, x)
    torch.cuda.synchronize()
    return sts, x


def kinds(sts, x):
    out = {}
    for k, name in enumerate(NAMES):
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
    sts, x = fill(B)
    for use_wl in (0, 1, 0, 1):
        rt.set_tuning("wl_bal", use_wl)
        k = kinds(sts, x)
        print(f"B={B} wl_bal={use_wl}: step {step_ms(sts, x):.3f} ms; us per launch group: " + "  ".join(f"{a} {b:.1f}" for a, b in k.items()), flush=True)
    rt.set_tuning("use_wl", 1)
    for nw in ():
        rt.set_tuning("attn_lm", nw)
        k = kinds(sts, x)
        print(f"B={B} attn_lm={nw}: step {step_ms(sts, x):.3f} ms; attn {k['attn']:.1f} us", flush=True)
    for s in sts:
        s.close()
