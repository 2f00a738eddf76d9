#!/usr/bin/env python3
"""BASELINE.json configs[4] / SURVEY.md 8d config 5: a synthetic "TVSum-shaped" set run end to end on the GPU path:

  seeded uint8 videos -> LiveInferForBenchmark (reset / set_fps / input_query_stream / input_video_stream /
  inference, the reference's evaluation flow of test/inference.py:592-711) -> prediction records in the reference's
  JSON schema -> fused score alpha*info + beta*rel - eps*max(0, unc - tau) with the reference's `tvsum` grid-search
  parameters (outputs/grid_search_params.json) -> TVSum metrics (mAP50, mAP15, top-5 mAP, Spearman, Kendall, F1@15)
  and Savitzky-Golay peak picking (find_ticks).

Ground truth is synthetic in the dataset's shape (20 annotators x importance 1..5 per frame, averaged and divided by 5:
test/tvsum/tvsum_utils.py:95-122), weights are seeded random, so the metric VALUES mean nothing; the point is that
every stage of the path runs at full model size, how long it takes, and that the post-processing given these scores is
the ported one (pinned to the reference's functions in tests/test_postproc.py).  Videos shard over ranks like streams
(aha_amd.sharding.streams_of_rank); rank 0 gathers the records.

    python tools/eval_synth_tvsum.py [--videos 50] [--min-frames 80] [--max-frames 640] [--cache default_sink]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/eval_synth_tvsum.py ...
"""
import argparse, json, os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aha_amd  # noqa: E402
from aha_amd.arguments import LiveTestArguments  # noqa: E402
from aha_amd.checkpoint import write_predictions  # noqa: E402
from aha_amd.config import preset  # noqa: E402
from aha_amd.live_infer import LiveInferForBenchmark, LiveInferForDemo, round_numbers  # noqa: E402
from aha_amd.postproc import evaluate_f1, evaluate_tvsum, fuse_scores  # noqa: E402
from aha_amd.runtime import Runtime  # noqa: E402
from aha_amd.sharding import streams_of_rank  # noqa: E402
from aha_amd.synth import make_weights  # noqa: E402

TVSUM_PARAMS = dict(alpha=0.0, beta=-1.0, epsilon=-5.0, uncertainty_threshold=0.04)   # outputs/grid_search_params.json "tvsum"

ap = argparse.ArgumentParser()
ap.add_argument("--videos", type=int, default=50); ap.add_argument("--min-frames", type=int, default=80)
ap.add_argument("--max-frames", type=int, default=640); ap.add_argument("--preset", default="bench")
ap.add_argument("--cache", default="default_sink", choices=["default_sink", "sliding_window", "static", "none"])
ap.add_argument("--frames-per-step", type=int, default=1, help="static cache only: exact frame batching")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "synth_tvsum"))
a = ap.parse_args()

world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")                       # records are Python objects: gathered on the host
torch.cuda.set_device(local)
dev = f"cuda:{local}"
cfg = preset(a.preset)
S = cfg.vision.image_size
rt = Runtime(cfg, make_weights(cfg, device=dev, dtype=torch.bfloat16), device=dev, max_step_tokens=512,   # with lm_head: the query turn argmaxes
             max_vit_frames=32)
torch.cuda.empty_cache()

lengths = np.random.default_rng(0).integers(a.min_frames, a.max_frames + 1, a.videos)


def video(v):                      # counter-based frames: frame i of video v depends only on (v, i)
    g = torch.Generator(device=dev)
    out = torch.empty((int(lengths[v]), 3, S, S), dtype=torch.uint8, device=dev)
    for i in range(out.shape[0]):
        g.manual_seed(1_000_003 * v + i)
        out[i] = torch.randint(0, 256, (3, S, S), generator=g, device=dev, dtype=torch.uint8)
    return out


def ground_truth(v):               # 20 annotators x importance 1..5, mean / 5 (the shape tvsum_utils builds)
    anno = np.random.default_rng(10_000 + v).integers(1, 6, (20, int(lengths[v])))
    return anno.mean(0) / 5.0


args = LiveTestArguments(frame_fps=1, stream_end_prob_threshold=9.0, frame_resolution=S,
                         frame_num_tokens=cfg.frame_num_tokens)       # threshold 9: never triggers a response turn
drv = LiveInferForBenchmark(args, runtime=rt, alt_cache=None if a.cache == "none" else a.cache)
mine = streams_of_rank(a.videos, world, rank)
records, n_frames = [], 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for v in mine:
    frames = video(v)
    drv.reset()
    drv.set_fps(fps=1)
    drv.input_query_stream([{"role": "user", "content": "Which moments of this video are the highlights?", "time": 0}])
    drv.input_video_stream(frames)
    responses = drv.inference(frames_per_step=a.frames_per_step)
    records.append({"video_uuid": f"synth_{v:03d}", "model_response_list": responses, "video_duration": float(frames.shape[0]),
                    "true_frames_list": list(range(frames.shape[0])), "debug_data": round_numbers(drv.debug_data_list, 3)})
    n_frames += frames.shape[0]
torch.cuda.synchronize()
dt = time.perf_counter() - t0

if world > 1:
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((records, n_frames, dt), gathered, dst=0)
    if rank == 0:
        records = sorted((r for part in gathered for r in part[0]), key=lambda r: r["video_uuid"])
        n_frames, dt = sum(p[1] for p in gathered), max(p[2] for p in gathered)
    dist.barrier()
if rank == 0:
    os.makedirs(a.out, exist_ok=True)
    write_predictions(os.path.join(a.out, "predictions.json"), records)
    pred = {r["video_uuid"]: fuse_scores(r["debug_data"], **TVSUM_PARAMS) for r in records}
    gt = {f"synth_{v:03d}": ground_truth(v) for v in range(a.videos)}
    assert all(len(pred[k]) == len(gt[k]) for k in pred), "one fused score per frame"
    m50, m15, top5, spe, ken = evaluate_tvsum(gt, pred)
    f1 = evaluate_f1(gt, pred)
    ticks = {k: LiveInferForDemo.find_ticks(None, np.array([e["relevance_score"] for e in r["debug_data"]]), fps=1)
             for k, r in ((r["video_uuid"], r) for r in records[:5])}
    summary = {"videos": a.videos, "frames": int(n_frames), "n_gpus": world, "cache": a.cache, "frames_per_step": a.frames_per_step,
               "seconds": round(dt, 2), "frames_per_s_end_to_end": round(n_frames / dt, 1),
               "metrics_on_synthetic_gt": {"mAP50": m50, "mAP15": m15, "top5_mAP": top5, "spearman": spe, "kendall": ken, "f1_15": f1},
               "peaks_first_videos_s": {k: [float(x) for x in v] for k, v in ticks.items()},
               "finite": bool(all(np.isfinite(p).all() for p in pred.values()))}
    json.dump(summary, open(os.path.join(a.out, "summary.json"), "w"), indent=1)
    print(json.dumps(summary))
if world > 1:
    dist.destroy_process_group()
