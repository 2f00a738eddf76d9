import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime
cfg = preset("bench")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=128, max_vit_frames=32); del w
fr = make_frames(32, cfg.vision.image_size, seed=0).cuda()
st = rt.open_stream("static", 2048, 0)
x = (torch.randn(1, 36, cfg.lm.hidden_size, device="cuda") * 0.1).bfloat16()
rt.lm_step([st], x[:, :20].contiguous())
def measure(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    enq = 0.0; t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter(); fn(); enq += time.perf_counter() - a
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"{name:22s} host enqueue {enq/n*1e3:6.3f} ms   wall {tot/n*1e3:6.3f} ms per call")
measure("vit 1 frame", lambda: rt.visual_embed(fr[:1]))
measure("vit 32 frames", lambda: rt.visual_embed(fr), 5)
measure("lm_step T=36", lambda: rt.lm_step([st], x))
