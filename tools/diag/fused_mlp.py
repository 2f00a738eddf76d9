"""Fused MLP block (resid_norm + gate/up + down in one launch with grid barriers) vs three launches:
bit-exactness of scores / hidden state and LM step time, full size, static cache, T = 36."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
cfg = preset(sys.argv[1] if len(sys.argv) > 1 else "bench")
rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True), max_step_tokens=128, max_vit_frames=1)
torch.cuda.empty_cache()
H, tf = cfg.lm.hidden_size, cfg.frame_num_tokens
g = torch.Generator().manual_seed(3)
prefix = (torch.randn(1, 55, H, generator=g) * 0.05).bfloat16().cuda()
X = (torch.randn(12, tf, H, generator=g) * 0.05).bfloat16().cuda()
outs = {}
for mode in (0, 1, 2, 0, 1, 2):
    rt.set_tuning("wpb_gateup", 8 if mode else 5); rt.set_tuning("fuse_mlp", mode)
    st = rt.open_stream("default_sink", 2048, 32)
    rt.lm_step([st], prefix)
    sc = []
    for i in range(12):
        s, hid = rt.lm_step([st], X[i:i + 1], want_hidden=True)[::2] if False else (rt.lm_step([st], X[i:i + 1]), None)
        sc.append(s.clone())
    sc = torch.cat(sc).cpu()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        for i in range(12): rt.lm_step([st], X[i:i + 1])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 36
    print(f"fuse_mlp={mode}: {dt * 1e3:.3f} ms per LM step; finite {bool(torch.isfinite(sc).all())}", flush=True)
    outs.setdefault(mode, sc)
    st.close()
print("fused (fences) == unfused bit-exact:", torch.equal(outs[0], outs[1]), "| fused (sc1 hand-offs) == unfused:", torch.equal(outs[0], outs[2]))
