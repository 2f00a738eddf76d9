#!/usr/bin/env python3
"""Where a persistent-engine launch (lm_engine.hip) spends its time: per-workgroup wall-clock stamps of the LAST layer's launch of a
single-stream static step, reported in us relative to the earliest loader start (min / median / max over the workgroups).
    python tools/diag/engine_stamps.py [T] [key=value ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime

cfg = preset("bench"); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
T = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else tf
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=128, max_vit_frames=8)
del w; torch.cuda.empty_cache()
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); rt.set_tuning(k, int(v))
g = torch.Generator(device="cuda").manual_seed(0)
st = rt.open_stream("static", 2048, 32)
rt.lm_step([st], (torch.randn(1, 20, H, generator=g, device="cuda") * 0.05).bfloat16())
x = (torch.randn(1, T, H, generator=g, device="cuda") * 0.05).bfloat16()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
buf = torch.zeros((n_cu, 16), dtype=torch.int64, device="cuda")
for _ in range(3):
    rt.lm_step([st], x)
rt._chk(rt.lib.aha_lm_engine_stamps(rt.ctx, buf.data_ptr()))
LEVEL = 1
NAMES = {0: "loader start", 1: "gate/up last slot issued", 2: "down last slot issued", 3: "loader drained", 4: "gate/up X published (rows done seen)",
         5: "down X published (slice done seen)", 6: "gate/up first slot in hand", 7: "gate/up last slot consumed", 8: "activation published",
         9: "down first slot in hand", 10: "down last slot consumed", 12: "row normalised (row workgroups)", 13: "workgroup done", 14: "row: slab loads issued", 15: "row: sums exchanged"}
if any(kv == "engine=2" for kv in sys.argv[1:]):      # lm_stream.hip: the same slots, read as
    NAMES.update({0: "workgroup start", 1: "gate/up prefetch issued, waiting for the rows", 2: "down prefetch issued, waiting for the slice", 6: "gate/up first X chunk staged",
                  7: "gate/up last chunk computed + act stored", 9: "down first X chunk staged", 10: "down last chunk computed + slabs stored"})
    NAMES.pop(3, None)
for rep in range(3):
    buf.zero_()
    rt.lm_step([st], x); torch.cuda.synchronize()
    s = buf.cpu().double()
    t0 = s[:, 0][s[:, 0] > 0].min()
    print(f"--- run {rep} (us after the earliest loader start; 100 MHz clock)")
    for k, name in NAMES.items():
        v = s[:, k]; v = v[v > 0]
        if len(v) == 0: continue
        r = (v - t0) / 100.0
        print(f"  {k:2d} {name:40s} n={len(v):3d}  min {r.min():7.2f}  med {r.median():7.2f}  max {r.max():7.2f}")
    # per group of 8 (blockIdx % 8): when its slice was published and when its down phase ended
    d = (s[:, 13] - t0) / 100.0
    print("  done by group (blockIdx % 8): " + "  ".join(f"{g}: {d[g::8].max():.1f}" for g in range(8)))
rt._chk(rt.lib.aha_lm_engine_stamps(rt.ctx, None))
