#!/usr/bin/env python3
"""Cache attention over a long growing cache (600 frames = 21,655 keys), one layer, one stream, T = 36: HIP-event
time per call for each kernel choice / key-split length.   python tools/diag/long_attn.py [Lk] [key=value ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import LiveConfig, LMConfig, VisionConfig
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime

Lk = int(sys.argv[1]) if len(sys.argv) > 1 else 21655
cfg = LiveConfig(vision=VisionConfig(image_size=56, patch_size=14, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256),
                 lm=LMConfig(num_hidden_layers=1, vocab_size=512), video_pooling_stride=2, name="op7b_long")
rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16), max_step_tokens=320, max_vit_frames=1, max_positions=32768)
d = rt.desc
g = torch.Generator(device="cuda").manual_seed(1)
B = int(os.environ.get("B", "1"))
sts = []
for b in range(B):
    st = rt.open_stream(None, capacity=32768)
    done = 0
    while done < Lk:
        T = min(512 if False else 36, Lk - done)
        k = torch.randn(d.kv_heads, T, d.head_dim, generator=g, device="cuda").bfloat16()
        v = torch.randn(d.kv_heads, T, d.head_dim, generator=g, device="cuda").bfloat16()
        rt._chk(rt.lib.aha_cache_update(rt.ctx, st.handle, 0, k.data_ptr(), v.data_ptr(), T, None, None, torch.cuda.current_stream().cuda_stream))
        done += T
    sts.append(st)
T = 36
q = (torch.randn(B, T, d.heads * d.head_dim, generator=g, device="cuda")).bfloat16()
kv_bytes = B * Lk * d.kv_heads * d.head_dim * 2 * 2


def timed(split_len, reps=30):
    for _ in range(3):
        out = rt.attention(sts, q, 0, causal_off=[Lk - T] * B, split_len=split_len)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = rt.attention(sts, q, 0, causal_off=[Lk - T] * B, split_len=split_len)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, out


ref = None
for kv in sys.argv[2:] or ["attn_lm=1"]:
    k, v = kv.split("=")
    if k == "split":
        sl = int(v)
    else:
        rt.set_tuning(k, int(v)); sl = 0
    us, out = timed(sl)
    if ref is None:
        ref = out.clone()
    print(f"Lk={Lk} B={B} {kv}: {us:7.1f} us per call (includes ~10 us of host call + descriptor upload) -> {kv_bytes / us / 1e6:.2f} TB/s of K/V; "
          f"max |d| vs first {(out.float() - ref.float()).abs().max().item():.3e}", flush=True)
