#!/usr/bin/env python3
"""Vision encode on the latency path with and without weight-prefetch riders (tunings "vit_prefetch", "vit_riders"), interleaved in one process,
the caches flushed by a 1-GiB write between encodes as the LM step flushes them in the real loop.  Also checks that the embeddings are
bit-identical with and without riders.
    python tools/diag/vit_prefetch.py [frames,...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import LiveConfig, LMConfig
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime
ns = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4]
cfg = LiveConfig(lm=LMConfig(num_hidden_layers=1, vocab_size=1024), name="vit24")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=max(max(ns), 4)); del w
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")


def med(fr, flush):
    ts = []
    for i in range(14):
        if flush:
            junk.add_(1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rt.visual_embed(fr); e1.record(); e1.synchronize()
        if i >= 4:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for n in ns:
    fr = make_frames(n, cfg.vision.image_size, seed=0).cuda()
    rt.set_tuning("vit_prefetch", 0)
    ref = rt.visual_embed(fr).clone()
    for rows, riders in ((0, 256), (2400, 256), (2400, 128), (0, 256), (2400, 256)):
        rt.set_tuning("vit_prefetch", rows); rt.set_tuning("vit_riders", riders)
        same = torch.equal(rt.visual_embed(fr), ref)
        print(f"{n} frame(s) vit_prefetch={rows:4d} vit_riders={riders:3d}: {med(fr, True):.3f} ms (caches flushed)  {med(fr, False):.3f} ms (back to back)  bits {'same' if same else 'DIFFER'}", flush=True)
