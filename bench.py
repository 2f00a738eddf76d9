#!/usr/bin/env python3
"""bench.py -- frames/sec scored by the per-frame streaming path (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic input for every stream this rank
owns: `--frames` uint8 frames per stream are encoded by the vision tower in one batch (the
reference pre-encodes 32-frame batches, test/inference.py:181-185) and then scored frame by frame
by the LM step against the stream's KV cache (the per-frame loop of test/inference.py:283-335; each step is
replayed from a captured HIP graph, bit-identical to direct launches), ending when the [frames,3] score rows
are host-visible.  N=1 workload = BASELINE.json configs[1]: SigLIP-L/14@336 + Qwen2-7B bf16, single stream,
static KV cache.  With N>1 every rank runs its own independent stream(s) (weak scaling) and the per-step score
rows are all-gathered with RCCL.

`--gpus N` with no WORLD_SIZE in the environment: this process launches the N ranks itself
(`python -m torch.distributed.run --nproc-per-node N ... bench.py ...`) BEFORE touching the GPU and relays rank 0's
JSON line.  Under an external launcher (WORLD_SIZE set) `--gpus` must equal WORLD_SIZE.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: bf16 dense ~2.5 PFLOP/s


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
    p.add_argument("--no-secondary", action="store_true", help="skip the secondary data (sink_w2048, eight_stream_sink, static batching)")
    p.add_argument("--lm-priority", action="store_true", help="run the LM chain on a high-priority HIP stream")
    p.add_argument("--overlap", action="store_true",
                   help="encode batch k+1 on a second HIP stream while the LM scores batch k (round 3: measured 1.8 %% SLOWER than serial with "
                        "the persistent tower kernels, which hold every CU for a whole GEMM; default off)")
    p.add_argument("--no-overlap", action="store_true", help="(default now; kept for old command lines)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget: frames are scored until it is spent (at most --frames)")
    p.add_argument("--force-dist", action="store_true",
                   help="initialise torch.distributed even for one rank (exercises the RCCL barrier / all-gather / all-reduce path)")
    p.add_argument("--abi-allgather", action="store_true", help="(ignored; the C-ABI collective is checked by tools/abi_allgather_check.py in its own process group)")
    p.add_argument("--no-abi-allgather", action="store_true", help="(ignored; kept for old command lines)")
    p.add_argument("--no-configs3", action="store_true", help="with more than one rank: skip the configs[3] datum (8 streams per GPU on SinkCache)")
    p.add_argument("--no-ref-geometry", action="store_true", help="skip the secondary datum on the reference-faithful geometry (so400m/14@384, Tf=49)")
    p.add_argument("--dry-run-collective", action="store_true",
                   help="launcher / rendezvous / all-gather plumbing only, on synthetic score rows: no GPU, no hot path, value = null")
    p.add_argument("--tile-dma", type=int, default=-1, help="experiment: force a tiled-GEMM variant in the vision tower")
    p.add_argument("--vit-cus", type=int, default=0,
                   help="experiment: restrict the vision stream to this many CUs (HIP CU mask, XCD-balanced)")
    p.add_argument("--lm-cus", type=int, default=-1,
                   help="with --vit-cus: CUs of the LM stream (-1 = the complement of the vision stream's, 0 = all)")
    return p.parse_args()


def launch_ranks(a):
    """--gpus N without an external launcher: start N ranks as children (never exec: this process has not touched the GPU and
    does not need to), relay rank 0's JSON line, exit with the launcher's code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
    if p.returncode != 0 or line is None:
        sys.stderr.write(p.stdout[-4000:])
        sys.exit(p.returncode or 1)
    print(line)
    sys.exit(0)


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


def pmc_traffic(kernel_prefix, workload):
    """HBM bytes per launch of a kernel from the newest committed PMC summary (counters cannot be collected from inside the
    process being measured) - only from a counter run of the SAME workload as the timed one: a summary row carries the
    workload it was collected on ("static_1stream", "static_8stream", "sink_1stream_steady", "sink_8stream_steady"); rows of any
    other workload, or a summary that does not say, give None.  `kernel_prefix` may be a tuple: the launches of one timed group
    (cache attention + its combine kernel) - their bytes are added, and every one of them must have a row."""
    prefixes = (kernel_prefix,) if isinstance(kernel_prefix, str) else tuple(kernel_prefix)
    for name in ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        # the round-2 file was collected on static-cache runs only (tools/pmc_round.sh of that round), 1 and 8 streams merged per kernel
        total = 0
        for pre in prefixes:
            hit = None
            for k in d["kernels"]:
                wl = k.get("workload", d.get("workload", "static_1stream+static_8stream"))
                if pre in k["kernel"] and workload in wl.split("+"):
                    hit = k["hbm_bytes_per_launch"]
                    break
            if hit is None:
                total = None
                break
            total += hit
        if total is not None:
            return total, name
    return None, None


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


class Workload:
    """B streams x F frames per step on one GPU: batched vision encode on a second HIP stream (double-buffered embeddings,
    events both ways) while the LM scores the previous batch; every batch's encode and all of its LM steps are timed."""

    def __init__(self, rt, cfg, dev, B, F, cache, window, sink, frames_all, prefix_ids, query_ids, main_stream, vit_stream, gather=None):
        self.rt, self.B, self.F, self.tf, self.H = rt, B, F, cfg.frame_num_tokens, cfg.lm.hidden_size
        self.frames_all, self.main_stream, self.vit_stream, self.gather = frames_all, main_stream, vit_stream, gather
        self.streams = [rt.open_stream(cache, window, sink, capacity=cfg.lm.max_position_embeddings) for _ in range(B)]
        self.scores_host = torch.empty((F, B, 3), dtype=torch.float32).pin_memory()
        self.scores_dev = torch.empty((F, B, 3), dtype=torch.float32, device=dev)
        self.last_global = None
        tf, H = self.tf, self.H
        # stream prologue (untimed): query turn first (test/inference.py:294-298), then system prompt + frame 0
        q = rt.embed_tokens(query_ids).view(1, -1, H).expand(B, -1, -1).contiguous()
        rt.lm_step(self.streams, q)
        emb0 = rt.visual_embed(frames_all[::F].contiguous()).view(B, tf, H)
        pre = rt.embed_tokens(prefix_ids).view(1, -1, H).expand(B, -1, -1)
        rt.lm_step(self.streams, torch.cat([pre, emb0], dim=1).contiguous())
        self.emb_buf = [torch.empty((B * F * tf, H), dtype=torch.bfloat16, device=dev) for _ in range(2)]
        self.emb_ready = [torch.cuda.Event() for _ in range(2)]
        self.emb_free = [torch.cuda.Event() for _ in range(2)]
        for e in self.emb_free:
            e.record(main_stream)

    def encode(self, k):
        with torch.cuda.stream(self.vit_stream):
            self.vit_stream.wait_event(self.emb_free[k & 1])           # the LM is done with this slot
            self.rt.visual_embed(self.frames_all, out=self.emb_buf[k & 1])
            self.emb_ready[k & 1].record(self.vit_stream)

    def run(self, n_steps):
        B, F, tf, H = self.B, self.F, self.tf, self.H
        pending = None
        with torch.cuda.stream(self.main_stream):
            self.encode(0)
            for k in range(n_steps):
                if k + 1 < n_steps:
                    self.encode(k + 1)
                self.main_stream.wait_event(self.emb_ready[k & 1])
                emb = self.emb_buf[k & 1].view(B, F, tf, H)
                for i in range(F):
                    self.rt.lm_step(self.streams, emb[:, i].contiguous(), out=self.scores_dev[i])
                self.emb_free[k & 1].record(self.main_stream)
                if self.gather is not None:
                    # one collective per step on the [F, B, 3] score rows, started here and waited for a step later: it runs
                    # underneath the next step's LM launches; the last one is resolved before run() returns
                    if pending is not None:
                        self.last_global = pending.result()
                    pending = self.gather(self.scores_dev)
                self.scores_host.copy_(self.scores_dev, non_blocking=True)
            if pending is not None:
                self.last_global = pending.result()

    def last_emb(self, n_steps):
        return self.emb_buf[(n_steps - 1) & 1].view(self.B, self.F, self.tf, self.H)

    def close(self):
        for s in self.streams:
            s.close()


def timed_kind(rt, wl, emb, kind, n_steps=4, skip=1):
    """HIP-event time of one launch kind on LM steps issued with nothing else in flight (direct launches: timed steps bypass
    graph replay so the events are live).  Returns (ms, launch groups, algorithmic bytes)."""
    rt.set_tuning("time_gemm", 1 << kind)
    ms = by = 0.0
    n = 0
    F = emb.shape[1]
    for i in range(n_steps + skip):
        rt.lm_step(wl.streams, emb[:, i % F].contiguous())
        torch.cuda.synchronize()
        if i >= skip:
            m, c, b = rt.last_gemm_time(kind)
            ms, n, by = ms + m, n + c, by + b
    rt.set_tuning("time_gemm", 0)
    return ms, n, by


def roofline_hbm(kernel, ms, n, by, traffic_prefix=None, workload=None):
    ach = (by / n) / ((ms / n) * 1e-3) / 1e9 if n and ms > 0 else None
    traffic, src = pmc_traffic(traffic_prefix, workload) if traffic_prefix and workload else (None, None)
    return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS if ach else None, "avg_launch_us": ms / n * 1e3 if n else None, "launches_timed": n,
            "algorithmic_bytes_per_launch": by / n if n else None, "traffic": traffic, "traffic_workload": workload if traffic is not None else None,
            "traffic_source": (f"profiles/{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 correction 2*FETCH+WRITE)" if src else None)}


def main():
    a = parse()
    a.no_overlap = not a.overlap
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)                                          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: under an external launcher pass --gpus equal to the number of ranks")
    dist = None
    use_dist = world > 1 or a.force_dist or a.dry_run_collective
    on_gpu = not a.dry_run_collective
    if on_gpu:
        local = local % max(1, torch.cuda.device_count())        # --backend gloo rehearsal: several ranks on one GPU
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
            if on_gpu:
                torch.cuda.set_device(local)
            if a.backend == "nccl" and on_gpu:                    # RCCL over xGMI
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
            else:
                dist.init_process_group("gloo")
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)

    import aha_amd  # noqa: F401
    from aha_amd.config import preset
    from aha_amd.sharding import gather_scores, gather_scores_async
    from aha_amd.synth import make_frames, make_token_ids, make_weights

    B, F = a.streams, a.frames
    n_streams_global = B * world                                  # stream g lives on rank g % world

    def ranks_seen():
        t = torch.tensor([rank], device=(f"cuda:{local}" if (on_gpu and a.backend == "nccl") else "cpu"))
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return len({int(x.item()) for x in out})

    if a.dry_run_collective:
        # launcher + rendezvous + collective plumbing on synthetic rows; nothing of the hot path runs and nothing is measured
        loc = torch.full((F, B, 3), float(rank))
        glob = gather_scores(loc, n_streams_global)
        ok = glob.shape == (F, n_streams_global, 3) and all(float(glob[0, g, 0]) == g % world for g in range(n_streams_global))
        seen = ranks_seen()
        dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "frames/sec scored (whole node)", "value": None, "unit": "frames/s", "n_gpus": world, "dry_run": True,
                              "ranks_seen": seen, "gather_ok": bool(ok), "backend": "gloo", "steps": a.steps, "warmup": a.warmup}))
        dist.destroy_process_group()
        return

    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    from aha_amd.runtime import Runtime

    cfg = preset(a.preset)
    tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
    cache = None if a.cache == "none" else a.cache
    n_sys, n_query = 35, 20
    base_cfg = B == 1 and a.cache == "static" and a.preset == "bench"
    secondary = (not a.no_secondary) and world == 1 and base_cfg
    # more than one rank: the one multi-GPU configuration BASELINE.json names is configs[3] (8 streams per GPU, 64 over the node) -
    # measured on every rank after the headline region, same barrier + max-over-ranks timing, the score rows all-gathered with RCCL
    configs3 = (not a.no_secondary) and (not a.no_configs3) and world > 1 and base_cfg
    B2 = 8                                                        # configs[3]: 64 streams over 8 GPUs
    w = make_weights(cfg, device=dev, dtype=torch.bfloat16, skip_lm_head=not secondary)   # lm_head only for the all-position-logits datum
    # vision batches: 32 frames for the headline stream (the reference pre-encodes 32 at a time, test/inference.py:181); the
    # 8-stream datum encodes its 256 frames per step in batches of 128 (better tile quantisation of the tower's N = 1024 GEMMs:
    # 18.2 vs 19.3 ms per 32 frames; a frame's embedding does not depend on the batch it is encoded in - bit-exact, tested)
    rt = Runtime(cfg, w, device=str(dev), max_step_tokens=max((B2 if (secondary or configs3) else B) * (tf + n_sys), 320),
                 max_vit_frames=128 if (secondary or configs3) else 32, max_positions=cfg.lm.max_position_embeddings)
    if a.tile_dma >= 0:
        rt.set_tuning("tile_dma", a.tile_dma)
    want_cpu = (not a.no_cpu_baseline) and rank == 0             # rank 0 only, at every world size (timed after the ranks have parted)
    w_cpu = {k: v.cpu() for k, v in w.items()} if want_cpu else None
    del w
    torch.cuda.empty_cache()

    prefix_ids = make_token_ids(n_sys, cfg.lm.vocab_size, seed=100)
    query_ids = make_token_ids(n_query, cfg.lm.vocab_size, seed=101)
    frames = [make_frames(F, cfg.vision.image_size, seed=1000 * rank + s).to(dev) for s in range(B)]
    frames_all = torch.cat(frames, 0)                              # [B*F,3,S,S] stream-major

    # The vision tower is MFMA-bound, the LM steps are HBM-bound and they use disjoint workspaces, so with --overlap the tower
    # of batch k+1 runs on a second HIP stream while the LM scores batch k.  Default: one stream.  Measured zero-sum in every form
    # (persistent tower, background tower that leaves room for LM workgroups on every CU, priorities): a second active queue costs
    # each of the LM step's ~200 launches ~2.6 us, more than the tower's 15 % share returns (profiles/r03_overlap_background_tower.txt).
    main_stream = torch.cuda.Stream(priority=-1) if a.lm_priority else torch.cuda.current_stream()   # LM chain: short kernels
    vit_stream = torch.cuda.Stream() if not a.no_overlap else main_stream
    if a.vit_cus > 0 and not a.no_overlap:
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        vit_stream = cu_masked_stream(0, a.vit_cus, n_cu)
        if a.lm_cus != 0:
            lm_n = n_cu - a.vit_cus if a.lm_cus < 0 else a.lm_cus
            main_stream = cu_masked_stream(n_cu - lm_n, lm_n, n_cu)

    def gather(scores_dev):
        loc = scores_dev if a.backend == "nccl" else scores_dev.cpu()
        return gather_scores_async(loc, n_streams_global)         # handle; .result() -> [F, B*world, 3] in global stream order

    wl = Workload(rt, cfg, dev, B, F, cache, a.window, a.sink, frames_all, prefix_ids, query_ids, main_stream, vit_stream,
                  gather if use_dist else None)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    wl.run(a.warmup)
    sync()
    t0 = time.perf_counter()
    wl.run(a.steps)
    sync()
    dt = time.perf_counter() - t0
    dist_info = None
    if use_dist:
        t = torch.tensor([dt], device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert wl.last_global.shape == (F, n_streams_global, 3) and torch.isfinite(wl.last_global).all()
        dt = t.item()
        # the collective on its own: one all-gather of the [F, B, 3] rows per step (latency-bound: a few hundred bytes per rank)
        for _ in range(5):
            gather(wl.scores_dev).result()
        sync()
        t1 = time.perf_counter()
        for _ in range(50):
            gather(wl.scores_dev).result()
        torch.cuda.synchronize()
        ag_us = (time.perf_counter() - t1) / 50 * 1e6
        dist_info = {"allgather_us": ag_us, "ranks_seen": ranks_seen(), "backend": "RCCL (torch.distributed nccl)" if a.backend == "nccl" else "gloo",
                     "rows_per_rank": F * B, "bytes_per_rank": F * B * 12,
                     "overlap": "step k's all-gather is waited for after step k+1's LM launches are enqueued"}
        assert dist_info["ranks_seen"] == a.gpus
    assert torch.isfinite(wl.scores_host).all()

    # Dominant kernel: the gate/up weight-streaming GEMM (fused SwiGLU), timed with HIP events on its launch stream around each of
    # its 28 launches per step, on identical LM steps issued right after the timed region with nothing else in flight: same
    # process, same buffers, same stream state.  Timed steps are launched directly (not replayed), so the events are live.
    emb_last = wl.last_emb(a.steps)
    g_ms, g_n, g_bytes = timed_kind(rt, wl, emb_last, 2)
    wb, kvb, fl = rt.last_step_work()

    # p50 per-frame latency: ViT(1 frame) + LM step + score D2H, events on the launch stream
    lat = []
    one = frames_all[:B].contiguous()
    for i in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e = rt.visual_embed(one).view(B, tf, H)
        s = rt.lm_step(wl.streams, e)
        wl.scores_host[0].copy_(s, non_blocking=True)
        e1.record()
        e1.synchronize()
        if i >= 8:
            lat.append(e0.elapsed_time(e1))
    lat.sort()

    # the two stages of the step on their own, HIP events on the launch stream (same process, right after the timed region):
    # F LM steps as the timed loop issues them (graph replay), and one batched vision encode of the step's frames
    def ev_ms(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps
    emb_v = emb_last.view(B, F, tf, H)

    # (on a frozen TrulyStaticCache - the headline - these steps change no stream state; with --cache none or a sink / sliding
    # policy they would append 3 F tf keys behind the measurement and can run a growing stream past the RoPE table, so such runs
    # carry no whole-LM-step / whole-step fractions)
    def lm_pass():
        for i in range(F):
            rt.lm_step(wl.streams, emb_v[:, i].contiguous(), out=wl.scores_dev[i])
    lm_ms_per_step = ev_ms(lm_pass, 2) / F if a.cache == "static" else None
    vit_ms = ev_ms(lambda: rt.visual_embed(frames_all), 3)

    # per-kind GEMM breakdown of one LM step (diagnostic, outside the timed region)
    rt.set_tuning("time_gemm", 15)
    rt.lm_step(wl.streams, rt.visual_embed(one).view(B, tf, H))
    torch.cuda.synchronize()
    kinds = {}
    for k, name in enumerate(["qkv", "o_proj", "gate_up_swiglu", "down_proj"]):
        ms, n, by = rt.last_gemm_time(k)
        kinds[name] = {"ms": round(ms, 4), "launches": n, "GBps": round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
    rt.set_tuning("time_gemm", 0)

    # [r6] the two single-launch forms of the layer's MLP half (tuning "engine"; opt-in, bit-identical: tests/test_gpu_layers.py) against the launches:
    # the same F LM steps by HIP events, alternated twice in this process - the driver-run record of profiles/r06_engine_mlp_stamps.txt
    engines_datum = None
    if secondary and a.cache == "static":
        res = {0: [], 1: [], 2: []}
        for _ in range(2):
            for lv in (0, 1, 2):
                rt.set_tuning("engine", lv)
                res[lv].append(ev_ms(lm_pass, 2) / F)
        rt.set_tuning("engine", 0)
        engines_datum = {"what": "one whole LM step (ms, HIP events, graph replay, two alternated samples each): the launches (shipped) / lm_engine.hip "
                                 "(LDS-DMA loader ring: resid_norm + gate/up + down_proj in one launch) / lm_stream.hip (register-streaming gate/up -> down_proj)",
                         "launches_ms": res[0], "engine1_ring_ms": res[1], "engine2_stream_ms": res[2]}

    static_batched = sink_datum = eight_datum = logits_datum = growing_datum = None
    if configs3:
        wl.close()
        eight_datum = eight_stream_datum(rt, cfg, dev, a, B2, prefix_ids, query_ids, main_stream, vit_stream, sync, rank=rank, world=world,
                                         dist=dist, backend=a.backend)
    elif secondary:
        logits_datum = all_position_logits_datum(rt, wl, frames_all, F, tf, H, cfg.lm.vocab_size, dt / a.steps * 1e3, sync)
        static_batched = static_batching_datum(rt, wl, frames_all, F, tf, H, a.steps, sync)
        wl.close()
        sink_datum = sink_w2048_datum(rt, cfg, dev, a, frames_all, prefix_ids, query_ids, main_stream, vit_stream, sync)
        growing_datum = growing_600_datum(rt, cfg, dev, a, frames_all, prefix_ids, query_ids)
        eight_datum = eight_stream_datum(rt, cfg, dev, a, B2, prefix_ids, query_ids, main_stream, vit_stream, sync)
    else:
        wl.close()

    ref_datum = None
    if secondary and not a.no_ref_geometry:
        rt.close()
        rt = None
        torch.cuda.empty_cache()
        ref_datum = ref_geometry_datum(dev, a, prefix_ids, query_ids, main_stream, vit_stream, sync)
    if rt is not None:
        rt.close()

    out = None
    if rank == 0:
        total_frames = F * B * world * a.steps
        rf = roofline_hbm("gemm_ws_kernel<MT,2,KC,SWIGLU> (gate/up projection + SwiGLU)", g_ms, g_n, g_bytes, "gemm_ws_kernel<3, 2,", "static_1stream")
        # The dominant kernel is the BEST-placed part of the step; the fractions a reader should take away are the stages' and the
        # whole step's (SURVEY.md 8d algorithmic work: LM = weights streamed once per step + K/V read; vision + projector = MFMA flops)
        v, P = cfg.vision, cfg.vision.num_patches
        vit_flops = B * F * (2.0 * P * (3 * v.patch_size ** 2 * v.hidden_size + v.num_hidden_layers * (4 * v.hidden_size ** 2 + 2 * v.hidden_size * v.intermediate_size))
                             + 4.0 * P * P * v.hidden_size * v.num_hidden_layers
                             + 2.0 * (4 * tf) * (v.hidden_size * H + H * H))          # projector on the 4*Tf rows bilinear pooling samples
        vit_w_bytes = 2.0 * (3 * v.patch_size ** 2 * v.hidden_size + v.num_hidden_layers * (4 * v.hidden_size ** 2 + 2 * v.hidden_size * v.intermediate_size)
                             + v.hidden_size * H + H * H)
        lm_bytes = wb + kvb
        step_ms = dt / a.steps * 1e3
        if lm_ms_per_step is not None:
            rf["lm_step"] = {"bound": "hbm", "what": f"one whole LM step (28 layers + heads, B={B}, T={tf}): weights + K/V bytes / its time by HIP events",
                             "achieved": lm_bytes / (lm_ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": lm_bytes / (lm_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": lm_ms_per_step, "algorithmic_bytes": lm_bytes}
        rf["vision"] = {"bound": "mfma", "what": f"tower + projector + pool of {B * F} frames: algorithmic flops / its time by HIP events",
                        "achieved": vit_flops / (vit_ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": vit_flops / (vit_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "ms": vit_ms, "algorithmic_flops": vit_flops}
        rf["step"] = {"bound": "hbm", "what": "the whole timed step (vision batch + F LM steps): algorithmic HBM bytes / ms_per_step",
                      "achieved": (F * lm_bytes + vit_w_bytes) / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": (F * lm_bytes + vit_w_bytes) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "lm_share_of_step": F * lm_ms_per_step / step_ms if lm_ms_per_step is not None else None, "vision_share_of_step": vit_ms / step_ms}
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
            "roofline": rf,
            "distributed": dist_info,
            "with_all_position_logits": logits_datum,
            "static_cache_batched_frames": static_batched,
            "sink_w2048": sink_datum,
            "growing_600": growing_datum,
            "eight_stream_sink": eight_datum,
            "ref_so400m_384": ref_datum,
            "layer_engines": engines_datum,
            "lm_step": {"weight_bytes": wb, "kv_bytes": kvb, "flops": fl, "gemm_kinds": kinds},
        }
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(cfg, w_cpu, frames[0][:32].cpu(), prefix_ids, query_ids, a.cache, a.window,
                                               a.sink, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None

    if use_dist:
        # nothing below talks to another rank.  The C ABI's own collective (aha_allgather_scores) is not exercised here: it runs in
        # its own process group (tools/abi_allgather_check.py; tests/test_gpu_configs.py launches it), where a stall is that program's
        # non-zero exit and can never be reported as a successful bench.
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def ref_geometry_datum(dev, a, prefix_ids, query_ids, main_stream, vit_stream, sync):
    """secondary: the same single-stream static-cache step on the REFERENCE-FAITHFUL geometry (models/arguments_live.py:22-24,
    SURVEY.md fact 3): so400m/14@384 tower (26 layers, width 1152, 16 heads x 72, MLP 4304, 729 patches) + Qwen2-7B dims,
    bilinear 27 -> 7 pooling = 49 tokens per frame.  Its own runtime and seeded weights; every encode and LM step timed."""
    from aha_amd.config import preset
    from aha_amd.runtime import Runtime
    from aha_amd.synth import make_frames, make_weights
    cfg = preset("ref")
    F, tf = a.frames, cfg.frame_num_tokens
    w = make_weights(cfg, device=dev, dtype=torch.bfloat16, skip_lm_head=True)
    rt = Runtime(cfg, w, device=str(dev), max_step_tokens=max(tf + 35, 320), max_vit_frames=32, max_positions=cfg.lm.max_position_embeddings)
    del w
    torch.cuda.empty_cache()
    frames = make_frames(F, cfg.vision.image_size, seed=3000).to(dev)
    wl = Workload(rt, cfg, dev, 1, F, "static", a.window, a.sink, frames, prefix_ids, query_ids, main_stream, vit_stream)
    wl.run(2)
    sync()
    steps = 5
    t0 = time.perf_counter()
    wl.run(steps)
    sync()
    dt = time.perf_counter() - t0
    assert torch.isfinite(wl.scores_host).all()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        rt.visual_embed(frames)
    e1.record()
    e1.synchronize()
    out = {"workload": f"configs[1] on the reference-faithful geometry: so400m/14@384 + Qwen2-7B dims, 1 stream, static KV cache, {F} frames/step, Tf={tf} tokens/frame",
           "frames_per_s": F * steps / dt, "ms_per_step": dt / steps * 1e3, "vision_ms_per_32_frames": e0.elapsed_time(e1) / 3 / F * 32}
    wl.close()
    rt.close()
    return out


def all_position_logits_datum(rt, wl, frames_all, F, tf, H, V, headline_ms, sync):
    """secondary (NOT `value`): the reference's forward runs lm_head over EVERY position of every frame step (logits fp32 [1,T,V],
    video_head_live_llava_qwen.py:175) and then uses none of it on a frame step; SURVEY.md 8(d) leaves lm_head out of the algorithmic
    work and the headline loop does not compute it.  This datum does, so that the cost of the reference's literal per-frame work is
    on record: the headline step + aha_lm_logits_all after every LM step (1.09 GB of lm_head weights and 21.9 MB of fp32 logits per frame)."""
    logits = torch.empty((tf, V), dtype=torch.float32, device=frames_all.device)
    streams, scores_dev, scores_host = wl.streams, wl.scores_dev, wl.scores_host

    def step():
        emb = rt.visual_embed(frames_all).view(F, tf, H)
        for i in range(F):
            rt.lm_step(streams, emb[i:i + 1], out=scores_dev[i])
            rt.logits_all(1, tf, out=logits)
        scores_host.copy_(scores_dev, non_blocking=True)
    step()
    sync()
    assert torch.isfinite(logits).all()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    return {"workload": "the headline step + lm_head over all T positions after every frame's LM step (what the reference's forward computes)",
            "frames_per_s": F / dt * 1e3, "ms_per_step": dt, "lm_head_ms_per_frame": (dt - headline_ms) / F}


def static_batching_datum(rt, wl, frames_all, F, tf, H, steps, sync):
    """secondary (NOT `value`): TrulyStaticCache frames are independent once the cache is frozen (test/static_cache.py:26-36;
    tests/test_gpu_parity.py proves it bit-exactly), so G frames of one stream can share one pass over the weights by listing
    the frozen stream G times in a single aha_lm_step; and only each frame's last token can influence its scores."""
    G = max(1, 320 // tf)                                          # rows one fused gate/up pass holds (gemm_ws: 20 row tiles)
    streams, scores_dev, scores_host = wl.streams, wl.scores_dev, wl.scores_host

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
    for _ in range(steps):
        step_batched()
    sync()
    dtb = time.perf_counter() - t1
    out = {"frames_per_lm_step": G, "frames_per_s": F * steps / dtb, "ms_per_step": dtb / steps * 1e3,
           "max_abs_score_diff_vs_sequential": max_dev}            # 0.0: bit-identical

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
    for _ in range(steps):
        step_last_token()
    sync()
    dtl = time.perf_counter() - t2
    out["last_token_only"] = {"frames_per_s": F * steps / dtl, "ms_per_step": dtl / steps * 1e3, "max_abs_score_diff_vs_sequential": max_dev2}
    return out


def sink_w2048_datum(rt, cfg, dev, a, frames_all, prefix_ids, query_ids, main_stream, vit_stream, sync):
    """secondary: the same single-stream step on SinkCache(W=2048, sink=32) at STEADY STATE - cache full, every step evicts,
    re-rotates the 1,980 kept keys of all 28 layers in place and attends over 2,048 keys (BASELINE configs[2] geometry)."""
    F, tf = a.frames, cfg.frame_num_tokens
    wl = Workload(rt, cfg, dev, 1, F, "default_sink", 2048, 32, frames_all[:F], prefix_ids, query_ids, main_stream, vit_stream)
    fill = -(-(2048 // tf + 2) // F)                              # steps until the window is full
    wl.run(fill)
    sync()
    assert wl.streams[0].get_seq_length() == 2048
    steps = 5
    t0 = time.perf_counter()
    wl.run(steps)
    sync()
    dt = time.perf_counter() - t0
    assert torch.isfinite(wl.scores_host).all()
    emb = wl.last_emb(steps)
    a_ms, a_n, a_by = timed_kind(rt, wl, emb, 4)
    r_ms, r_n, r_by = timed_kind(rt, wl, emb, 5)
    out = {"workload": f"1 stream, SinkCache W=2048 sink=32 at steady state (evicting every step), {F} frames/step",
           "frames_per_s": F * steps / dt, "ms_per_step": dt / steps * 1e3,
           "roofline_attention": roofline_hbm("attn_fwd_kernel<128,true> + attn_combine_kernel (one layer: 2,048 keys x 4 KV heads, K and V read once; traffic: both kernels)",
                                              a_ms, a_n, a_by, ("attn_fwd_kernel<128, true>", "attn_combine_kernel"), "sink_1stream_steady"),
           "roofline_rerotation": roofline_hbm("sink_rerotate_kernel<128> (all 28 layers: kept keys read + written in place)", r_ms, r_n, r_by,
                                               "sink_rerotate_kernel", "sink_1stream_steady")}
    wl.close()
    return out


def growing_600_datum(rt, cfg, dev, a, frames_all, prefix_ids, query_ids):
    """secondary: SURVEY.md 8d config 2's "DynamicCache-equivalent for 600 frames" - past_key_values=None in the reference
    (test/inference.py:154-155): the cache only grows, 20 + 35 + 600 x 36 = 21,655 keys at the end, 1.24 GB of K/V read per LM step.
    One stream, every frame scored in order (vision in batches of 32); the attention roofline is taken on the last frames, where K/V -
    not the weights' share of a launch - is the HBM term."""
    F, tf, H = a.frames, cfg.frame_num_tokens, cfg.lm.hidden_size
    n_frames = 600
    st = rt.open_stream(None, capacity=cfg.lm.max_position_embeddings)
    rt.lm_step([st], rt.embed_tokens(query_ids).view(1, -1, H))
    pre = rt.embed_tokens(prefix_ids).view(1, -1, H)
    scores = torch.empty((n_frames, 3), dtype=torch.float32, device=dev)
    host = torch.empty((n_frames, 3), dtype=torch.float32).pin_memory()

    class _W:                                                      # what timed_kind needs
        streams = [st]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tail_ms = None
    for i0 in range(0, n_frames, F):
        n = min(F, n_frames - i0)
        emb = rt.visual_embed(frames_all[:n]).view(n, tf, H)
        last = i0 + n >= n_frames
        if last:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for j in range(n):
            x = emb[j:j + 1] if i0 + j else torch.cat([pre, emb[:1]], 1).contiguous()
            rt.lm_step([st], x, out=scores[i0 + j:i0 + j + 1])
        if last:
            e1.record()
    host.copy_(scores, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tail_ms = e0.elapsed_time(e1) / n
    assert torch.isfinite(host).all() and st.get_seq_length() == 20 + 35 + n_frames * tf
    # attention at the final length (the timed steps append 36 keys each: 21.7k keys)
    emb = rt.visual_embed(frames_all[:F]).view(1, F, tf, H)
    a_ms, a_n, a_by = timed_kind(rt, _W, emb, 4, n_steps=2)
    wb, kvb, fl = rt.last_step_work()
    out = {"workload": f"1 stream, growing cache (past_key_values=None), {n_frames} frames scored in order: {st.get_seq_length()} keys at the end",
           "frames_per_s": n_frames / dt, "ms_total": dt * 1e3, "lm_step_ms_last_batch": tail_ms, "keys_at_end": st.get_seq_length(),
           "kv_bytes_per_step_at_end": kvb, "weight_bytes_per_step": wb,
           "lm_step_hbm_frac_at_end": (wb + kvb) / (tail_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "roofline_attention": roofline_hbm("attn_lm_kernel<128,8> + attn_combine16c_kernel at ~21.7k keys (one layer: K and V of 4 KV heads read once; traffic: both kernels)",
                                              a_ms, a_n, a_by, ("attn_lm_kernel<128, 8>", "attn_combine16c_kernel"), "growing_1stream_tail")}
    st.close()
    return out


def eight_stream_datum(rt, cfg, dev, a, B2, prefix_ids, query_ids, main_stream, vit_stream, sync, rank=0, world=1, dist=None, backend="nccl"):
    """The per-GPU share of BASELINE configs[3] (64 streams over 8 GPUs): 8 streams batched into every LM step (M = 288 rows per
    weight pass) on SinkCache(W=2048, sink=32) at steady state, vision encode of 8 x F frames per step.  One rank: a secondary
    datum.  More than one rank: every rank runs its 8 streams (stream g of the node lives on rank g % world), each step's [F, 8, 3]
    score rows are all-gathered with RCCL underneath the next step, and the time is barrier-bracketed and max-reduced like `value`."""
    from aha_amd.sharding import gather_scores_async
    from aha_amd.synth import make_frames
    F, tf = a.frames, cfg.frame_num_tokens
    frames8 = torch.cat([make_frames(F, cfg.vision.image_size, seed=2000 + 100 * rank + s).to(dev) for s in range(B2)], 0)
    gather = None
    if world > 1:
        def gather(scores_dev):
            return gather_scores_async(scores_dev if backend == "nccl" else scores_dev.cpu(), B2 * world)
    wl = Workload(rt, cfg, dev, B2, F, "default_sink", 2048, 32, frames8, prefix_ids, query_ids, main_stream, vit_stream, gather)
    fill = -(-(2048 // tf + 2) // F)
    wl.run(fill)
    sync()
    assert all(s.get_seq_length() == 2048 for s in wl.streams)
    steps = 3
    t0 = time.perf_counter()
    wl.run(steps)
    sync()
    dt = time.perf_counter() - t0
    ag_us = None
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        assert wl.last_global.shape == (F, B2 * world, 3) and torch.isfinite(wl.last_global).all()
        for _ in range(5):
            gather(wl.scores_dev).result()
        sync()
        t1 = time.perf_counter()
        for _ in range(50):
            gather(wl.scores_dev).result()
        torch.cuda.synchronize()
        ag_us = (time.perf_counter() - t1) / 50 * 1e6
    assert torch.isfinite(wl.scores_host).all()
    emb = wl.last_emb(steps)
    g_ms, g_n, g_by = timed_kind(rt, wl, emb, 2, n_steps=2)
    a_ms, a_n, a_by = timed_kind(rt, wl, emb, 4, n_steps=2)
    wb, kvb, fl = rt.last_step_work()
    M = B2 * tf
    flops = 2.0 * M * (2 * cfg.lm.intermediate_size) * cfg.lm.hidden_size
    tf_s = flops / ((g_ms / g_n) * 1e-3) / 1e12 if g_n else None
    # LM step on its own (graph replay, HIP events) and against both roofs: the step's algorithmic flops and bytes
    emb_v = emb.view(B2, F, tf, cfg.lm.hidden_size)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(F):
        rt.lm_step(wl.streams, emb_v[:, i].contiguous(), out=wl.scores_dev[i])
    e1.record()
    e1.synchronize()
    lm_ms = e0.elapsed_time(e1) / F
    # the GEMM does not read the cache: its counter row from the 8-stream static run is the same launch (same M, same weights)
    rl = roofline_hbm("gate/up + SwiGLU at M = 288 rows", g_ms, g_n, g_by, "gemm_wl_bal18_kernel", "static_8stream")
    rl.update({"mfma_achieved_TFLOPs": tf_s, "mfma_peak_TFLOPs": MFMA_PEAK_TFLOPS, "mfma_frac": tf_s / MFMA_PEAK_TFLOPS if tf_s else None,
               "flops_per_launch": flops, "note": "arithmetic intensity ~288 flop/B sits on the ridge (312): both fractions are reported"})
    out = {"workload": f"configs[3] per-GPU share: {B2} streams/GPU x {world} GPU(s), SinkCache W=2048 sink=32 at steady state, {F} frames/stream/step "
                       f"(M = {M} rows per LM step)" + (f", {'RCCL' if backend == 'nccl' else 'gloo'} all-gather of the score rows" if world > 1 else ""),
           "frames_per_s": world * B2 * F * steps / dt, "ms_per_step": dt / steps * 1e3, "n_gpus": world, "streams_total": B2 * world,
           "allgather_us": ag_us, "lm_step_ms": lm_ms,
           "lm_step_mfma_frac": (fl / (lm_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS) if fl else None,
           "lm_step_hbm_frac": (wb + kvb) / (lm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "roofline_gate_up": rl,
           "roofline_attention": roofline_hbm("attn_lm_kernel<128,8> + attn_combine16_kernel at 8 streams x 2,048 keys (one layer; traffic: both kernels)",
                                              a_ms, a_n, a_by, ("attn_lm_kernel<128, 8>", "attn_combine16_kernel"), "sink_8stream_steady")}
    wl.close()
    return out


if __name__ == "__main__":
    main()
