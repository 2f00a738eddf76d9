#!/usr/bin/env python3
"""SinkCache re-rotation kernel at steady state (W=2048, 28 layers): HIP-event time per launch and achieved HBM rate.
python tools/diag/rerot_time.py [streams]"""
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
rt = Runtime(cfg, w, max_step_tokens=640, max_vit_frames=8); del w
g = torch.Generator(device="cuda").manual_seed(0)
sts = [rt.open_stream("default_sink", 2048, 32) for _ in range(B)]
x = (torch.randn(B, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
for _ in range(60): rt.lm_step(sts, x)
torch.cuda.synchronize()
for pg in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]:
    rt.set_tuning("rerot_pg", pg)
    rt.set_tuning("time_gemm", 1 << 5)
    ms = n = by = 0
    for _ in range(8):
        rt.lm_step(sts, x); torch.cuda.synchronize()
        m, c, b = rt.last_gemm_time(5); ms, n, by = ms + m, n + c, by + b
    rt.set_tuning("time_gemm", 0)
    print(f"sink_rerotate_kernel pg={pg}: {ms / n * 1e3:.2f} us per launch, {by / n / 1e6:.1f} MB algorithmic -> {by / (ms * 1e-3) / 1e12:.2f} TB/s ({by / (ms * 1e-3) / 8e12:.3f} of 8 TB/s), B={B}")
