"""HIP-graph replay of frozen-static LM steps: bit-exactness vs direct launches, LM step time, live event timing."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
cfg = preset(sys.argv[1] if len(sys.argv) > 1 else "bench")
policy = sys.argv[2] if len(sys.argv) > 2 else "static"
rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True), max_step_tokens=128, max_vit_frames=1)
torch.cuda.empty_cache()
H, tf = cfg.lm.hidden_size, cfg.frame_num_tokens
g = torch.Generator().manual_seed(3)
prefix = (torch.randn(1, 55, H, generator=g) * 0.05).bfloat16().cuda()
X = (torch.randn(12, tf, H, generator=g) * 0.05).bfloat16().cuda()
outs = {}
for mode in (0, 1, 0, 1):
    rt.set_tuning("use_graph", mode)
    st = rt.open_stream(policy, 2048, 32 if policy == "default_sink" else 0)
    if policy != "static":
        for _ in range(60): rt.lm_step([st], X[0:1])          # fill the window (evicting steady state)
    rt.lm_step([st], prefix)
    sc = torch.cat([rt.lm_step([st], X[i:i + 1]).clone() for i in range(12)]).cpu()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        for i in range(12): rt.lm_step([st], X[i:i + 1])
    host = (time.perf_counter() - t) / 60
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 60
    rt.set_tuning("time_gemm", 1 << 2)
    for i in range(3): rt.lm_step([st], X[i:i + 1])
    torch.cuda.synchronize()
    ms, n, by = rt.last_gemm_time(2)
    rt.set_tuning("time_gemm", 0)
    print(f"use_graph={mode}: {dt * 1e3:.3f} ms per LM step (host enqueue {host * 1e3:.3f} ms); gate/up by events: {ms / max(n, 1) * 1e3:.1f} us x {n}; work {rt.last_step_work()[0] / 1e9:.2f} GB; finite {bool(torch.isfinite(sc).all())}", flush=True)
    outs.setdefault(mode, sc)
    st.close()
print("graph replay == direct launches bit-exact:", torch.equal(outs[0], outs[1]))
