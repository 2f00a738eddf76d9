"""Time aha_frame_ingest (both resamplers) for common source sizes at the bench resolution; prints us/frame and the
effective source-read rate.  Uses a vision-only context (no LM layers worth mentioning)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import LiveConfig, LMConfig, VisionConfig
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
res = int(sys.argv[1]) if len(sys.argv) > 1 else 336
cfg = LiveConfig(vision=VisionConfig(image_size=res, num_hidden_layers=1), lm=LMConfig(num_hidden_layers=1, vocab_size=1024), name="ingest")
w = make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True)
rt = Runtime(cfg, w, max_step_tokens=64, max_vit_frames=1); del w
for h, wd in [(480, 640), (720, 1280), (1080, 1920), (2160, 3840), (res, res)]:
    src = torch.randint(0, 256, (h, wd, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty((3, res, res), dtype=torch.uint8, device="cuda")
    for method, name in ((rt.RESIZE_PIL_BICUBIC, "pil-bicubic"), (rt.RESIZE_CV2_LINEAR, "cv2-linear")):
        for _ in range(3): rt.frame_ingest(src, method=method, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): rt.frame_ingest(src, method=method, out=out)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"{h}x{wd} -> {res}^2 {name:12s}: {us:7.1f} us/frame  ({h * wd * 3 / us / 1e3:.1f} GB/s of source bytes)")
