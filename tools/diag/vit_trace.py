"""Run the vision tower alone (bench geometry) for rocprofv3 --kernel-trace: argv = frames, tile_dma mode, reps."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime
n, mode, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = preset(sys.argv[4] if len(sys.argv) > 4 else "bench")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=32); del w
rt.set_tuning("tile_dma", mode)
if len(sys.argv) > 5: rt.set_tuning("attn_tpw", int(sys.argv[5]))
fr = make_frames(n, cfg.vision.image_size, seed=0).cuda()
import time
for _ in range(2): rt.visual_embed(fr)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps): rt.visual_embed(fr)
torch.cuda.synchronize()
print(f"{cfg.name}: {n} frame(s), tile_dma={mode}: {(time.perf_counter() - t) / reps * 1e3:.2f} ms per encode")
