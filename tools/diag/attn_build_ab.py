#!/usr/bin/env python3
"""Timing + checksums of everything attention.hip serves, for ONE build of the library (AHA_AMD_LIB selects it): run once per build in the
same gpurun call and compare.  Dense attention at 32 / 1 frames, tower encode at 32 / 1 frames, 8-stream SinkCache LM step (attn_lm_kernel +
combine), one stream on a 5,000-key growing cache (attn_fwd_kernel).     python tools/diag/attn_build_ab.py"""
import hashlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime

def sha(t): return hashlib.sha256(t.detach().cpu().contiguous().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
def ev_us(fn, n=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

print("library:", os.environ.get("AHA_AMD_LIB", "default"))
cfg = preset("bench"); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=640, max_vit_frames=32)
del w; torch.cuda.empty_cache()
g = torch.Generator(device="cuda").manual_seed(0)
for n in (32, 1):
    qkv = torch.randn(n, 576, 3 * 1024, generator=g, device="cuda").bfloat16()
    res = sorted(ev_us(lambda: rt.vit_attention(qkv, 16, 64)) for _ in range(5))
    print(f"dense attention {n:2d} frames: median {res[2]:7.1f} us  min {res[0]:7.1f}   sha {sha(rt.vit_attention(qkv, 16, 64))}")
fr = make_frames(32, cfg.vision.image_size, seed=0).cuda()
for n in (32, 1):
    res = sorted(ev_us(lambda: rt.visual_embed(fr[:n]), n=10) for _ in range(5))
    print(f"tower + projector {n:2d} frames: median {res[2] / 1e3:7.3f} ms  min {res[0] / 1e3:7.3f}   sha {sha(rt.visual_embed(fr[:n]))}")
# 8 streams on SinkCache at steady state
B = 8
sts = [rt.open_stream("default_sink", 2048, 32) for _ in range(B)]
x = (torch.randn(B, tf, H, generator=g, device="cuda") * 0.05).bfloat16()
for _ in range(60): rt.lm_step(sts, x)
res = sorted(ev_us(lambda: rt.lm_step(sts, x), n=20) for _ in range(5))
print(f"LM step, 8 streams x 2,048 keys (SinkCache): median {res[2] / 1e3:7.3f} ms  min {res[0] / 1e3:7.3f}   scores sha {sha(rt.lm_step(sts, x).to(torch.bfloat16))}")
for s in sts: s.close()
# one stream, growing cache of ~5,000 keys
st = rt.open_stream(None, capacity=8192)
x1 = x[:1].contiguous()
for _ in range(139): rt.lm_step([st], x1)
res = sorted(ev_us(lambda: rt.lm_step([st], x1), n=5, warm=1) for _ in range(3))
print(f"LM step, 1 stream, growing cache ~{st.get_seq_length()} keys: median {res[1] / 1e3:7.3f} ms   scores sha {sha(rt.lm_step([st], x1).to(torch.bfloat16))}")
