#!/usr/bin/env python3
"""Steady-state LM steps on a cache of W=2048 keys, for kernel traces and counter passes:
    rocprofv3 --kernel-trace --stats -- python3 tools/diag/sink_steps.py B [use_graph] [policy] [steps]
policy: default_sink (evicts + re-rotates every step once full), sliding_window (evicts), none (growing cache: long-key attention
without eviction).  Phase markers with wall-clock offsets go to stderr and, with AHA_DUMP_MAPS=<file>, the process's memory map
is written once every library is loaded (so that raw return addresses of a crash in a profiler thread can be resolved)."""
import os, sys, time
T0 = time.time()
def mark(msg):
    print(f"[sink_steps +{time.time() - T0:6.2f}s] {msg}", file=sys.stderr, flush=True)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
use_graph = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # direct launches by default (per-kernel trace rows); 1: graph replay
policy = sys.argv[3] if len(sys.argv) > 3 else "default_sink"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 64
cfg = preset("bench"); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
mark("imports done")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
torch.cuda.synchronize(); mark("weights made")
rt = Runtime(cfg, w, max_step_tokens=640, max_vit_frames=8)
del w
torch.cuda.synchronize(); mark("runtime built")
if os.environ.get("AHA_DUMP_MAPS"):
    with open(os.environ["AHA_DUMP_MAPS"], "w") as f:
        f.write(open("/proc/self/maps").read())
    mark("maps dumped")
rt.set_tuning("use_graph", use_graph)
g = torch.Generator(device="cuda").manual_seed(0)
if policy == "none":
    sts = [rt.open_stream(None, 0, 0, capacity=max(4096, steps * tf + 64)) for _ in range(B)]      # e.g. 600 steps: 21,600 keys
else:
    sts = [rt.open_stream(policy, 2048, 32 if policy == "default_sink" else 0) for _ in range(B)]
x = (torch.randn(B, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
for i in range(steps):
    rt.lm_step(sts, x)
    if i % 8 == 7:
        torch.cuda.synchronize(); mark(f"step {i + 1} done, cache length {sts[0].get_seq_length()}")
torch.cuda.synchronize(); mark("all steps done")
