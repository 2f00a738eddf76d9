#!/usr/bin/env python3
"""Long-cache LM attention: two launches (attn_fwd_kernel + attn_combine_kernel) vs one (attn_splitwave_kernel), per layer by HIP
events and per LM step, on SinkCache(W=2048) at steady state, interleaved rounds:  python tools/diag/attn_sw_time.py [streams]"""
import os, sys, time
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
assert sts[0].get_seq_length() == 2048
res = {}
for rnd in range(4):
    for mode in (0, 1):
        rt.set_tuning("attn_sw", mode)
        for _ in range(3): rt.lm_step(sts, x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): rt.lm_step(sts, x)
        torch.cuda.synchronize(); step_ms = (time.perf_counter() - t0) / 10 * 1e3
        rt.set_tuning("time_gemm", 1 << 4)
        ms = n = 0
        for _ in range(3):
            rt.lm_step(sts, x); torch.cuda.synchronize()
            m, c, _b = rt.last_gemm_time(4); ms, n = ms + m, n + c
        rt.set_tuning("time_gemm", 0)
        res.setdefault(mode, []).append((step_ms, ms / n * 1e3))
for mode in (0, 1):
    v = res[mode]
    print(f"attn_sw={mode}: lm_step {min(a for a, _ in v):.3f} ms (min of {len(v)}), attention {sorted(b for _, b in v)[len(v)//2]:.2f} us per layer (median), B={B}")
