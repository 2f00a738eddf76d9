#!/usr/bin/env python3
"""Ten 32-frame vision encodes at the default tuning, for a kernel trace:
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 $REPO/tools/diag/vit_trace.py [frames] [preset]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights, make_frames
from aha_amd.runtime import Runtime
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = preset(sys.argv[2] if len(sys.argv) > 2 else "bench")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=n)
fr = make_frames(n, cfg.vision.image_size, seed=1).cuda()
if os.environ.get("AHA_VIT_PREFETCH") is not None:
    rt.set_tuning("vit_prefetch", int(os.environ["AHA_VIT_PREFETCH"]))
junk = torch.empty(1 << 29, dtype=torch.uint8, device="cuda") if n <= 4 else None      # latency path: flush the caches between encodes as an LM step would
for _ in range(10):
    if junk is not None:
        junk.add_(1); torch.cuda.synchronize()
    rt.visual_embed(fr)
torch.cuda.synchronize()
rt.close()
