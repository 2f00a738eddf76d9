import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime
cfg = preset("bench")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=32); del w
fr = make_frames(32, cfg.vision.image_size, seed=0).cuda()
outs = {}
for mode in (0, 2, 5, 1):
    rt.set_tuning("tile_dma", mode)
    outs[mode] = rt.visual_embed(fr[:3]).clone()
    for n in (1, 2, 4, 8, 32):
        for _ in range(2): rt.visual_embed(fr[:n])
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): rt.visual_embed(fr[:n])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(f"tile_dma={mode} vit {n:2d} frames: {dt*1e3:.2f} ms ({n*400e9/dt/1e12:.0f} TF/s nominal)")
for m in (2, 5, 1):
    print(f"tile_dma={m} == register-staged bit-exact:", torch.equal(outs[0], outs[m]), "finite", bool(torch.isfinite(outs[m].float()).all()))
