import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import LiveConfig, LMConfig, VisionConfig, preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 28
cfg = LiveConfig(vision=VisionConfig(num_hidden_layers=1), lm=LMConfig(num_hidden_layers=layers, vocab_size=4096))
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=512, max_vit_frames=1, max_positions=4096); del w
H, tf = cfg.lm.hidden_size, 36
g = torch.Generator().manual_seed(1)
prefix = (torch.randn(1, 20, H, generator=g) * 0.1).bfloat16().cuda()
X = (torch.randn(14, tf, H, generator=g) * 0.1).bfloat16().cuda()
st = rt.open_stream("static", 2048, 0); rt.lm_step([st], prefix)
seq = torch.cat([rt.lm_step([st], X[i:i+1], want_raw=True)[1] for i in range(14)]).cpu()
for G in (1, 2, 3, 4, 5, 7, 8, 12, 14):
    out = torch.cat([rt.lm_step([st] * min(G, 14 - i), X[i:i+G].contiguous(), want_raw=True)[1] for i in range(0, 14, G)]).cpu()
    d = (out - seq).abs()
    print(f"G={G:2d} M={G*tf:3d} raw-logit max diff per head {d.max(0).values.tolist()}  worst row {d.max(1).values.argmax().item()}")
