import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import LiveConfig, LMConfig, VisionConfig
from aha_amd.synth import make_frames, make_weights
from aha_amd.runtime import Runtime
cfg = LiveConfig(vision=VisionConfig(num_hidden_layers=4), lm=LMConfig(num_hidden_layers=1, vocab_size=1024), name="vit4")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=32); del w
rt.set_tuning("tile_dma", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
fr = make_frames(32, cfg.vision.image_size, seed=0).cuda()
for _ in range(3): rt.visual_embed(fr)
torch.cuda.synchronize()
