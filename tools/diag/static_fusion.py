import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
cfg = preset("bench")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=256, max_vit_frames=1); del w
H, tf = cfg.lm.hidden_size, 36
g = torch.Generator().manual_seed(1)
prefix = (torch.randn(1, 20, H, generator=g) * 0.1).bfloat16().cuda()
X = (torch.randn(6, tf, H, generator=g) * 0.1).bfloat16().cuda()
res = {}
for fuse in (0, 1):
    rt.set_tuning("fuse_static", fuse)
    st = rt.open_stream("static", 2048, 0); rt.lm_step([st], prefix)
    res[fuse] = torch.cat([rt.lm_step([st], X[i:i+1], want_raw=True)[1] for i in range(6)]).cpu()
    for _ in range(3): rt.lm_step([st], X[:1])
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): rt.lm_step([st], X[:1])
    torch.cuda.synchronize(); print("fuse_static", fuse, "lm_step ms", round((time.perf_counter() - t) / 20 * 1e3, 3))
    res[(fuse, 'b')] = rt.lm_step([st] * 6, X, want_raw=True)[1].cpu()
print("fused == unfused bit-exact:", torch.equal(res[0], res[1]), "| batched fused == seq:", torch.equal(res[(1, 'b')], res[1]), "max diff", (res[0] - res[1]).abs().max().item())
