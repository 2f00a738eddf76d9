#!/usr/bin/env python3
"""Single-frame vision encode (the latency path of load_one_frame / the p50 figure): per-kernel trace target.
rocprofv3 --kernel-trace --stats -- python3 tools/diag/vit_one_frame.py [frames]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import LiveConfig, LMConfig, VisionConfig
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = LiveConfig(lm=LMConfig(num_hidden_layers=1, vocab_size=1024), name="vit24")          # the bench tower (24 layers), toy LM
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=max(n, 4)); del w
fr = make_frames(n, cfg.vision.image_size, seed=0).cuda()
for _ in range(5): rt.visual_embed(fr)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): rt.visual_embed(fr)
torch.cuda.synchronize()
print(f"{n} frame(s): {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per encode", flush=True)
