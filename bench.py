#!/usr/bin/env python3
"""bench.py -- frames/sec scored by the per-frame streaming path (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic input for every stream this rank
owns: `--frames` uint8 frames per stream are encoded by the vision tower in one batch (the
reference pre-encodes 32-frame batches, test/inference.py:181-185) and then scored frame by frame
by the LM step against the stream's KV cache (the per-frame loop of test/inference.py:283-335; with the frozen static
cache each step is replayed from a captured HIP graph, bit-identical to direct launches),
ending when the [frames,3] score rows are host-visible.  N=1 workload = BASELINE.json configs[1]:
SigLIP-L/14@336 + Qwen2-7B bf16, single stream, static KV cache.  With N>1 every rank runs its own
independent stream(s) (weak scaling) and the per-step score rows are all-gathered with RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import aha_amd  # noqa: E402,F401
from aha_amd.config import preset  # noqa: E402
from aha_amd.synth import make_frames, make_token_ids, make_weights  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--preset", default="bench")
    p.add_argument("--frames", type=int, default=32, help="frames per stream per step")
    p.add_argument("--streams", type=int, default=1, help="independent streams per GPU (batched LM step)")
    p.add_argument("--cache", default="static", choices=["static", "default_sink", "sliding_window", "none"])
    p.add_argument("--window", type=int, default=2048)
    p.add_argument("--sink", type=int, default=32)
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU-side collective (rehearsal)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--lm-priority", action="store_true", help="run the LM chain on a high-priority HIP stream")
    p.add_argument("--no-overlap", action="store_true", help="encode and score on one stream (no ViT/LM overlap)")
    p.add_argument("--cpu-seconds", type=float, default=20.0)
    p.add_argument("--force-dist", action="store_true",
                   help="initialise torch.distributed even for one rank (exercises the RCCL barrier / all-gather / all-reduce path)")
    p.add_argument("--tile-dma", type=int, default=-1, help="experiment: force a tiled-GEMM variant in the vision tower")
    p.add_argument("--vit-cus", type=int, default=0,
                   help="experiment: restrict the vision stream to this many CUs (HIP CU mask, XCD-balanced)")
    p.add_argument("--lm-cus", type=int, default=-1,
                   help="with --vit-cus: CUs of the LM stream (-1 = the complement of the vision stream's, 0 = all)")
    return p.parse_args()


def cu_masked_stream(first_cu, n_cus, total_cus):
    """A HIP stream whose kernels may only run on CUs [first_cu, first_cu+n_cus).  The driver deals consecutive
    mask bits round-robin over the XCDs, so a contiguous range takes the same share of every XCD."""
    import ctypes
    path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)   # the runtime torch loaded
    hip = ctypes.CDLL(path)
    words = (total_cus + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in range(first_cu, first_cu + n_cus):
        mask[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(st.value)


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (counters cannot be
    collected from inside the process being measured)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
        for k in d["kernels"]:
            if kernel_prefix in k["kernel"]:
                return k["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def host_cores():
    """Cores this process may actually use (cgroup quota / affinity), not the machine's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("AHA_CPU_THREADS", "16"))))


def cpu_baseline(cfg, weights_cpu, frames_u8, prefix_ids, query_ids, cache, window, sink, budget_s):
    """The oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample
    of the same workload: same weights, same frames, same cache policy."""
    from oracle.cache_policies import make_policy
    from oracle.qwen2_live import OracleLM, frame_scores
    from oracle.vision_tower import OracleVision
    cores = host_cores()
    torch.set_num_threads(cores)
    # the reference runs bf16; a CPU without native bf16 GEMM is faster in fp32 - use whichever this host runs faster
    def probe(dt):
        x, y = torch.randn(256, 2048).to(dt), torch.randn(2048, 2048).to(dt)
        x @ y
        t = time.perf_counter()
        for _ in range(3):
            x @ y
        return time.perf_counter() - t
    dt_cpu = torch.bfloat16 if probe(torch.bfloat16) <= probe(torch.float32) else torch.float32
    ov, olm = OracleVision(cfg, weights_cpu, dt_cpu), OracleLM(cfg.lm, weights_cpu, dt_cpu)
    pol = make_policy(None if cache == "none" else cache, window, sink)
    olm.step(olm.embed_tokens(query_ids), pol)                     # untimed: query turn (static prefix)
    tf = cfg.frame_num_tokens
    done, t0 = 0, time.perf_counter()
    for i in range(frames_u8.shape[0]):
        emb = ov.visual_embed(frames_u8[i:i + 1]).view(1, tf, -1)
        if i == 0:
            emb = torch.cat([olm.embed_tokens(prefix_ids), emb], dim=1)
        frame_scores(olm.step(emb, pol))
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{done} frames (ViT 1 frame + LM step each, frame 0 carries the system prompt), oracle {str(dt_cpu).split('.')[-1]} sdpa on {cores} threads, {dt:.1f}s"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    local = local % max(1, torch.cuda.device_count())            # --backend gloo rehearsal: several ranks on one GPU
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when the communicator is created (at the first collective): keep stdout for
        # the ONE JSON line by pointing fd 1 at stderr until the communicator exists.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            torch.cuda.set_device(local)
            if a.backend == "nccl":                               # RCCL over xGMI
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
            else:
                dist.init_process_group("gloo")
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    from aha_amd.sharding import gather_scores
    from aha_amd.runtime import Runtime

    cfg = preset(a.preset)
    tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
    B, F = a.streams, a.frames
    cache = None if a.cache == "none" else a.cache
    n_sys, n_query = 35, 20
    w = make_weights(cfg, device=dev, dtype=torch.bfloat16, skip_lm_head=True)
    rt = Runtime(cfg, w, device=str(dev), max_step_tokens=max(B * (tf + n_sys), 320), max_vit_frames=min(32, B * F),
                 max_positions=cfg.lm.max_position_embeddings)
    if a.tile_dma >= 0:
        rt.set_tuning("tile_dma", a.tile_dma)
    want_cpu = (not a.no_cpu_baseline) and rank == 0 and world == 1
    w_cpu = {k: v.cpu() for k, v in w.items()} if want_cpu else None
    del w
    torch.cuda.empty_cache()

    prefix_ids = make_token_ids(n_sys, cfg.lm.vocab_size, seed=100)
    query_ids = make_token_ids(n_query, cfg.lm.vocab_size, seed=101)
    frames = [make_frames(F, cfg.vision.image_size, seed=1000 * rank + s).to(dev) for s in range(B)]
    frames_all = torch.cat(frames, 0)                              # [B*F,3,S,S] stream-major
    streams = [rt.open_stream(cache, a.window, a.sink, capacity=cfg.lm.max_position_embeddings) for _ in range(B)]
    scores_host = torch.empty((F, B, 3), dtype=torch.float32).pin_memory()
    scores_dev = torch.empty((F, B, 3), dtype=torch.float32, device=dev)
    n_streams_global = B * world                                  # stream g lives on rank g % world

    # stream prologue (untimed): query turn first (test/inference.py:294-298), then system prompt + frame 0
    q = rt.embed_tokens(query_ids).view(1, -1, H).expand(B, -1, -1).contiguous()
    rt.lm_step(streams, q)
    emb0 = rt.visual_embed(frames_all[::F].contiguous()).view(B, tf, H)
    pre = rt.embed_tokens(prefix_ids).view(1, -1, H).expand(B, -1, -1)
    rt.lm_step(streams, torch.cat([pre, emb0], dim=1).contiguous())

    # The vision tower is MFMA-bound, the LM steps are HBM-bound and they use disjoint workspaces, so
    # the tower of batch k+1 runs on a second HIP stream while the LM scores batch k (double-buffered
    # embeddings, events both ways).  Every batch's encode and all of its LM steps are inside the
    # timed region; --no-overlap serialises them on one stream.
    main_stream = torch.cuda.Stream(priority=-1) if a.lm_priority else torch.cuda.current_stream()   # LM chain: short kernels
    vit_stream = torch.cuda.Stream() if not a.no_overlap else main_stream
    if a.vit_cus > 0 and not a.no_overlap:
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        vit_stream = cu_masked_stream(0, a.vit_cus, n_cu)
        if a.lm_cus != 0:
            lm_n = n_cu - a.vit_cus if a.lm_cus < 0 else a.lm_cus
            main_stream = cu_masked_stream(n_cu - lm_n, lm_n, n_cu)
    emb_buf = [torch.empty((B * F * tf, H), dtype=torch.bfloat16, device=dev) for _ in range(2)]
    emb_ready = [torch.cuda.Event() for _ in range(2)]
    emb_free = [torch.cuda.Event() for _ in range(2)]
    for e in emb_free:
        e.record(main_stream)

    def encode(k):
        with torch.cuda.stream(vit_stream):
            vit_stream.wait_event(emb_free[k & 1])                 # the LM is done with this slot
            rt.visual_embed(frames_all, out=emb_buf[k & 1])
            emb_ready[k & 1].record(vit_stream)

    def run(n_steps):
        with torch.cuda.stream(main_stream):
            _run(n_steps)

    def _run(n_steps):
        encode(0)
        for k in range(n_steps):
            if k + 1 < n_steps:
                encode(k + 1)
            main_stream.wait_event(emb_ready[k & 1])
            emb = emb_buf[k & 1].view(B, F, tf, H)
            for i in range(F):
                scores_dev[i] = rt.lm_step(streams, emb[:, i].contiguous())
            emb_free[k & 1].record(main_stream)
            if use_dist:
                # one collective per step on [F, B, 3] score rows -> [F, B*world, 3] in global stream order
                loc = scores_dev if a.backend == "nccl" else scores_dev.cpu()
                run.last_global = gather_scores(loc, n_streams_global)
            scores_host.copy_(scores_dev, non_blocking=True)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run(a.warmup)
    sync()
    t0 = time.perf_counter()
    run(a.steps)
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert run.last_global.shape == (F, n_streams_global, 3) and torch.isfinite(run.last_global).all()
        dt = t.item()
    assert torch.isfinite(scores_host).all()

    # Dominant kernel: the gate/up weight-streaming GEMM (fused SwiGLU).  In the timed region the LM steps are replayed from
    # a HIP graph and share the GPU with the next batch's vision tower, so the kernel is timed with HIP events (on its launch
    # stream, around each of its 28 launches) on identical LM steps issued right after the region, with nothing else in
    # flight: same process, same buffers, same stream state.  Average over 4 steps (112 launches).
    emb_last = emb_buf[(a.steps - 1) & 1].view(B, F, tf, H)
    rt.set_tuning("time_gemm", 1 << 2)
    g_ms = g_bytes = 0.0
    g_n = 0
    for i in range(6):
        rt.lm_step(streams, emb_last[:, i % F].contiguous())
        torch.cuda.synchronize()
        if i >= 2:                                                    # first two: direct launch, then graph capture
            ms, n, by = rt.last_gemm_time(2)
            g_ms, g_n, g_bytes = g_ms + ms, g_n + n, g_bytes + by
    rt.set_tuning("time_gemm", 0)
    wb, kvb, fl = rt.last_step_work()

    # p50 per-frame latency: ViT(1 frame) + LM step + score D2H, events on the launch stream
    lat = []
    one = frames_all[:B].contiguous()
    for i in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e = rt.visual_embed(one).view(B, tf, H)
        s = rt.lm_step(streams, e)
        scores_host[0].copy_(s, non_blocking=True)
        e1.record()
        e1.synchronize()
        if i >= 8:
            lat.append(e0.elapsed_time(e1))
    lat.sort()

    # secondary datum (NOT `value`): TrulyStaticCache frames are independent once the cache is frozen
    # (test/static_cache.py:26-36; tests/test_gpu_parity.py proves it bit-exactly), so G frames of one stream can
    # share one pass over the weights by listing the frozen stream G times in a single aha_lm_step.
    static_batched = None
    if a.cache == "static" and B == 1:
        G = max(1, 320 // tf)                                          # rows one fused gate/up pass holds (gemm_ws: 20 row tiles)
        def step_batched():
            emb = rt.visual_embed(frames_all).view(F, tf, H)
            for i in range(0, F, G):
                g = min(G, F - i)
                scores_dev[i:i + g, 0] = rt.lm_step(streams * g, emb[i:i + g].contiguous())
            scores_host.copy_(scores_dev, non_blocking=True)
        ref = scores_host.clone()
        step_batched()
        sync()
        max_dev = (scores_host - ref).abs().max().item()             # vs the sequential pass on the same frames
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step_batched()
        sync()
        dtb = time.perf_counter() - t1
        static_batched = {"frames_per_lm_step": G, "frames_per_s": F * a.steps / dtb, "ms_per_step": dtb / a.steps * 1e3,
                          "max_abs_score_diff_vs_sequential": max_dev}    # 0.0: bit-identical
        # second observation (also NOT `value`): under the frozen static cache a new token attends only to the prefix,
        # so the scores read at position -1 do not depend on the other tf-1 tokens of the frame; feeding only each
        # frame's last token (at its RoPE position) is bit-identical and makes the step vision-bound.
        def step_last_token():
            emb = rt.visual_embed(frames_all).view(F, tf, H)
            streams[0].set_position_offset(tf - 1)
            for i in range(0, F, 16):                                 # at most 16 streams per aha_lm_step
                g = min(16, F - i)
                scores_dev[i:i + g, 0] = rt.lm_step(streams * g, emb[i:i + g, -1:].contiguous())
            streams[0].set_position_offset(0)
            scores_host.copy_(scores_dev, non_blocking=True)
        step_last_token()
        sync()
        max_dev2 = (scores_host - ref).abs().max().item()
        t2 = time.perf_counter()
        for _ in range(a.steps):
            step_last_token()
        sync()
        dtl = time.perf_counter() - t2
        static_batched["last_token_only"] = {"frames_per_s": F * a.steps / dtl, "ms_per_step": dtl / a.steps * 1e3,
                                             "max_abs_score_diff_vs_sequential": max_dev2}

    # per-kind GEMM breakdown of one LM step (diagnostic, outside the timed region)
    rt.set_tuning("time_gemm", 15)
    rt.lm_step(streams, rt.visual_embed(one).view(B, tf, H))
    torch.cuda.synchronize()
    kinds = {}
    for k, name in enumerate(["qkv", "o_proj", "gate_up_swiglu", "down_proj"]):
        ms, n, by = rt.last_gemm_time(k)
        kinds[name] = {"ms": round(ms, 4), "launches": n, "GBps": round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
    rt.set_tuning("time_gemm", 0)

    if rank == 0:
        total_frames = F * B * world * a.steps
        achieved = (g_bytes / g_n) / ((g_ms / g_n) * 1e-3) / 1e9 if g_n else None
        out = {
            "metric": "frames/sec scored (whole node)", "value": total_frames / dt, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"configs[1]: {cfg.name} shapes ({'so400m' if cfg.vision.hidden_size == 1152 else 'ViT-L'}/14@{cfg.vision.image_size} + Qwen2-7B dims), "
                                   f"{B} stream(s)/GPU, {a.cache} KV cache (W={a.window}), {F} frames/stream/step, "
                                   f"Tf={tf} tokens/frame, seeded random weights",
                       "frames_per_step": F * B * world, "streams_per_gpu": B, "cache": a.cache, "vit_lm_overlap": not a.no_overlap,
                       "parallelism": f"stream-sharded x{world}" + (f", {'RCCL' if a.backend == 'nccl' else 'gloo'} all-gather of scores" if world > 1 else "")},
            "p50_frame_latency_ms": lat[len(lat) // 2],
            "roofline": {"bound": "hbm", "kernel": "gemm_ws_kernel<MT,2,KC,SWIGLU> (gate/up projection + SwiGLU)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None,
                         "avg_launch_us": g_ms / g_n * 1e3 if g_n else None, "launches_timed": g_n,
                         "algorithmic_bytes_per_launch": g_bytes / g_n if g_n else None,
                         "traffic": pmc_traffic("gemm_ws_kernel<3, 2,"),
                         "traffic_source": "profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                           "command, gfx950 correction 2*FETCH+WRITE); null when absent"},
            "static_cache_batched_frames": static_batched,
            "lm_step": {"weight_bytes": wb, "kv_bytes": kvb, "flops": fl, "gemm_kinds": kinds},
        }
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(cfg, w_cpu, frames[0][:8].cpu(), prefix_ids, query_ids, a.cache, a.window,
                                               a.sink, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    for s in streams:
        s.close()
    rt.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
