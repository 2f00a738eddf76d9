#!/usr/bin/env python3
"""The persistent layer engine (lm_engine.hip) against the launches it replaces, same process: bit comparison of every parity tap and of the
scores, then interleaved timing of the single-stream static step (configs[1]).   python tools/diag/engine_ab.py [rounds] [T]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = preset("bench"); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
T = int(sys.argv[2]) if len(sys.argv) > 2 else tf
levels = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1"])]
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=128, max_vit_frames=8)
del w; torch.cuda.empty_cache()
g = torch.Generator(device="cuda").manual_seed(0)
st = rt.open_stream("static", 2048, 32)
x0 = (torch.randn(1, 20, H, generator=g, device="cuda") * 0.05).bfloat16()
rt.lm_step([st], x0)                                   # the frozen prefix (20 keys)
xs = [(torch.randn(1, T, H, generator=g, device="cuda") * 0.05).bfloat16() for _ in range(4)]

def run(level, x, taps=True):
    rt.set_tuning("engine", level)
    sc, raw, hid = rt.lm_step([st], x, want_raw=True, want_hidden=True)
    out = {"scores": sc.clone(), "raw": raw.clone(), "hid": hid.clone()}
    if taps:
        for name in ("h", "xn", "act"):
            out[name] = rt.debug_tap(name, 1, T).clone()
    torch.cuda.synchronize()
    return out

ok = True
for i, x in enumerate(xs):
    ref = run(0, x)
    for lv in levels[1:]:
        got = run(lv, x)
        for k in ref:
            same = torch.equal(ref[k], got[k])
            if not same:
                d = (ref[k].float() - got[k].float()).abs()
                print(f"input {i} engine={lv} {k}: DIFFERENT  max|d| {d.max().item():.4g}  n {int((d > 0).sum())} of {d.numel()}  nan {int(torch.isnan(got[k].float()).sum())}")
                ok = False
print("bit comparison:", "IDENTICAL" if ok else "MISMATCH", flush=True)

def step_ms(level, n=40):
    rt.set_tuning("engine", level)
    for _ in range(6):
        rt.lm_step([st], xs[0])
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        rt.lm_step([st], xs[0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

for r in range(rounds):
    print("  ".join(f"engine={lv}: {step_ms(lv):.3f} ms" for lv in levels), flush=True)
