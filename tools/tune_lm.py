#!/usr/bin/env python3
"""GPU tuning harness: LM-step time and per-kind GEMM rates for several split-K settings, ViT encode
time for 1 and 32 frames.  python tools/tune_lm.py [--cache static|default_sink] [--streams B]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime

ap = argparse.ArgumentParser()
ap.add_argument("--cache", default="static"); ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--keep-best", action="store_true"); ap.add_argument("--preset", default="bench"); ap.add_argument("--fill", type=int, default=0, help="frames to pre-fill the cache")
ap.add_argument("--sweep", default="wpb_gateup:4,2;wpb_down:4,2;wpb_qkv:4,2;wpb_o:4,2")
a = ap.parse_args()
cfg = preset(a.preset); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size; B = a.streams
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=max(128, B * 96), max_vit_frames=32); del w; torch.cuda.empty_cache()
streams = [rt.open_stream(None if a.cache == "none" else a.cache, 2048, 32, capacity=32768) for _ in range(B)]
x = (torch.randn(B, tf, H, device="cuda") * 0.1).bfloat16()
rt.lm_step(streams, x[:, :20].contiguous())
for _ in range(a.fill): rt.lm_step(streams, x)

def lm_time(n=20):
    for _ in range(3): rt.lm_step(streams, x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): rt.lm_step(streams, x)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

def kinds():
    rt.set_tuning("time_gemm", 15); rt.lm_step(streams, x); torch.cuda.synchronize()
    out = []
    for k in range(4):
        ms, n, by = rt.last_gemm_time(k); out.append((ms / n * 1e3, by / (ms * 1e-3) / 1e12))
    rt.set_tuning("time_gemm", 0); return out

print(f"cache={a.cache} B={B} Lk={streams[0].get_seq_length()}")
base = lm_time(); k = kinds()
print(f"default: lm_step {base:.3f} ms | " + " ".join(f"{n}:{us:.1f}us/{tb:.2f}TB/s" for n, (us, tb) in zip(["qkv", "o", "gu", "down"], k)))
for spec in a.sweep.split(";"):
    key, vals = spec.split(":")
    names = ["qkv", "o", "gateup", "down"]
    i = names.index(key.split("_", 1)[1]) if key.split("_", 1)[1] in names else -1
    for v in vals.split(","):
        rt.set_tuning(key, int(v)); t = lm_time(10); kk = kinds()
        if i >= 0:
            print(f"  {key}={v:>2s}: lm_step {t:.3f} ms  kind {kk[i][0]:.1f} us {kk[i][1]:.2f} TB/s")
        else:
            print(f"  {key}={v:>2s}: lm_step {t:.3f} ms | " + " ".join(f"{n}:{us:.1f}us" for n, (us, tb) in zip(["qkv", "o", "gu", "down"], kk)))
    if a.keep_best: pass
    else: rt.set_tuning(key, {"wpb_qkv": 4, "wpb_o": 4, "wpb_gateup": 5, "wpb_down": 8, "kc_small": 4}.get(key, 0))   # back to the shipped default
fr = make_frames(32, cfg.vision.image_size, seed=0).cuda()
for n in (1, 8, 32):
    for _ in range(2): rt.visual_embed(fr[:n])
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): rt.visual_embed(fr[:n])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"vit {n:2d} frames: {dt*1e3:.2f} ms  ({n/dt:.0f} frames/s, {n*400e9/dt/1e12:.0f} TF/s of 400 GFLOP/frame)")
