#!/usr/bin/env python3
"""The headline's LM step (1 stream, frozen TrulyStaticCache, 20-token prefix, T = 36): step time (graph replay) and the
per-kind launch-group times (direct launches).  python tools/diag/static_step.py"""
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
NAMES = ["qkv", "o", "gate_up", "down", "attn"]
st = rt.open_stream("static", 2048, 0)
rt.lm_step([st], (torch.randn(1, 20, H, generator=g, device="cuda") * 0.05).bfloat16())
x = (torch.randn(1, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
for rep in range(3):
    for _ in range(20):
        rt.lm_step([st], x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200):
        rt.lm_step([st], x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 200 * 1e3
    out = {}
    for k, name in enumerate(NAMES):
        rt.set_tuning("time_gemm", 1 << k)
        tot = n = 0
        for i in range(4):
            rt.lm_step([st], x); torch.cuda.synchronize()
            if i:
                m, c, _ = rt.last_gemm_time(k); tot += m; n += c
        out[name] = tot / max(n, 1) * 1e3
    rt.set_tuning("time_gemm", 0)
    print(f"static step {ms:.3f} ms; us per launch group: " + "  ".join(f"{a} {b:.1f}" for a, b in out.items()), flush=True)
