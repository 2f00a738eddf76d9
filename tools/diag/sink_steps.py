#!/usr/bin/env python3
"""A few steady-state LM steps on SinkCache(W=2048) for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/diag/sink_steps.py B"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = preset("bench"); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=640, max_vit_frames=8)
del w
rt.set_tuning("use_graph", int(sys.argv[2]) if len(sys.argv) > 2 else 0)      # direct launches by default (per-kernel trace rows); 1: graph replay
g = torch.Generator(device="cuda").manual_seed(0)
sts = [rt.open_stream("default_sink", 2048, 32) for _ in range(B)]
x = (torch.randn(B, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
for _ in range(64):
    rt.lm_step(sts, x)
torch.cuda.synchronize()
