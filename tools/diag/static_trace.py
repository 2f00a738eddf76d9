#!/usr/bin/env python3
"""Workload for a kernel trace of the headline's LM step: 1 stream, frozen TrulyStaticCache (20-token prefix), T = 36, graph replay.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/diag/static_trace.py [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
cfg = preset("bench"); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=128, max_vit_frames=8)
del w; torch.cuda.empty_cache()
for kv in sys.argv[2:]:
    k, v = kv.split("="); rt.set_tuning(k, int(v))
g = torch.Generator(device="cuda").manual_seed(0)
st = rt.open_stream("static", 2048, 0)
rt.lm_step([st], (torch.randn(1, 20, H, generator=g, device="cuda") * 0.05).bfloat16())
x = (torch.randn(1, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    rt.lm_step([st], x)
torch.cuda.synchronize()
