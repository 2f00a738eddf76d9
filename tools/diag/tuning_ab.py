#!/usr/bin/env python3
"""A/B of LM-step tunings on SinkCache(W=2048, sink=32) at steady state: per-kind launch times by HIP events and the step time.
    python tools/diag/tuning_ab.py B key=value [key=value ...]      e.g.  tuning_ab.py 8 attn_fuse=0 attn_fuse=1 attn_fuse=0 attn_fuse=1
Each key=value is applied in turn (cumulatively) and measured."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = preset("bench"); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=640, max_vit_frames=8)
del w; torch.cuda.empty_cache()
g = torch.Generator(device="cuda").manual_seed(0)
NAMES = ["qkv", "o", "gate_up", "down", "attn", "rerot"]
sts = [rt.open_stream("default_sink", 2048, 32) for _ in range(B)]
x = (torch.randn(B, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
for _ in range(60):
    rt.lm_step(sts, x)
torch.cuda.synchronize()


def kinds():
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


def step_ms(n=30):
    for _ in range(5):
        rt.lm_step(sts, x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        rt.lm_step(sts, x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


last = None
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    rt.set_tuning(k, int(v))
    sc = rt.lm_step(sts, x).cpu()
    kk = kinds()
    print(f"B={B} {k}={v}: step {step_ms():.3f} ms; us per launch group: " + "  ".join(f"{a} {b:.1f}" for a, b in kk.items())
          + f"   scores[0] {sc[0].tolist()}", flush=True)
