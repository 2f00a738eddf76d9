#!/usr/bin/env python3
"""The C ABI's own collective on its own: aha_comm_unique_id / aha_comm_init_rank / aha_allgather_scores (RCCL communicator built
from a unique id inside libaha_amd.so, include/aha_amd.h "collective") against torch.distributed's all_gather of the same rows.

A separate PROCESS GROUP on purpose: a collective that stalls must surface as this program's non-zero exit,
never inside bench.py.  Every rank arms a hard deadline that fires while the main thread is blocked INSIDE a C call (RCCL init,
hipStreamSynchronize, a barrier): a daemon watchdog thread - ctypes and torch release the GIL around those calls - names the step
the rank stalled in on stderr and ends the process with os._exit(3).  (A Python signal handler cannot do this: CPython runs
handlers only between bytecodes of the main thread, so SIGALRM is never served while that thread sits in C.)  Nothing is retried.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/abi_allgather_check.py
    python tools/abi_allgather_check.py            # one rank, sets up its own rendezvous

Rank 0 prints ONE JSON line {"ok": true, "ranks": N, "allgather_us": ...}; exit 0 iff every rank saw the right rows.
"""
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STEP = ["start"]


def arm_deadline(seconds, step=STEP, what="abi_allgather_check"):
    """Hard deadline for a process whose main thread may block inside C: after `seconds` a daemon thread writes the step the
    process is in to stderr (os.write: no locks shared with the blocked thread) and ends it with exit code 3.  Returns a
    callable that disarms it.  tests/test_host_logic.py exercises it against a main thread blocked in libc sleep()."""
    done = threading.Event()

    def _watch():
        if done.wait(seconds):
            return
        try:
            os.write(2, f"{what}: rank {os.environ.get('RANK', '0')} stalled in step '{step[0]}' ({seconds} s deadline)\n".encode())
        finally:
            os._exit(3)
    threading.Thread(target=_watch, name="deadline", daemon=True).start()
    return done.set


def main():
    rows = int(os.environ.get("AHA_CHECK_ROWS", "256"))
    disarm = arm_deadline(float(os.environ.get("AHA_CHECK_DEADLINE_S", "180")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("LOCAL_RANK", "0")
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    # RCCL prints a banner on stdout when a communicator is created: keep stdout for the one JSON line
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    STEP[0] = "torch.distributed init (RCCL)"
    dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    dist.barrier()
    import aha_amd  # noqa: F401
    from aha_amd import lib as L
    lib = L.get()
    STEP[0] = "aha_comm_unique_id + broadcast"
    idb = (C.c_ubyte * L.COMM_ID_BYTES)()
    box = [None]
    if rank == 0:
        rc = lib.aha_comm_unique_id(idb, L.COMM_ID_BYTES)
        box[0] = bytes(idb) if rc == 0 else None
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        sys.stderr.write("aha_comm_unique_id failed: " + lib.aha_comm_last_error().decode() + "\n")
        os._exit(4)
    idb = (C.c_ubyte * L.COMM_ID_BYTES).from_buffer_copy(box[0])
    STEP[0] = "aha_comm_init_rank (blocks until every rank has joined)"
    comm = C.c_void_p()
    rc = lib.aha_comm_init_rank(idb, L.COMM_ID_BYTES, world, rank, local, C.byref(comm))
    if rc != 0:
        sys.stderr.write(f"aha_comm_init_rank failed on rank {rank}: " + lib.aha_comm_last_error().decode() + "\n")
        os._exit(5)
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    loc = torch.rand((rows, 3), generator=g, device="cuda", dtype=torch.float32)
    glob = torch.zeros((world, rows, 3), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    STEP[0] = "aha_allgather_scores"
    for _ in range(3):
        rc |= lib.aha_allgather_scores(comm, loc.data_ptr(), rows, glob.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        rc |= lib.aha_allgather_scores(comm, loc.data_ptr(), rows, glob.data_ptr(), st)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    STEP[0] = "torch.distributed all_gather (the comparison)"
    want = [torch.empty_like(loc) for _ in range(world)]
    dist.all_gather(want, loc)
    ok = rc == 0 and torch.equal(glob, torch.stack(want))
    flag = torch.tensor([1 if ok else 0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    lib.aha_comm_destroy(comm)
    STEP[0] = "final barrier"
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(saved, 1)
    os.close(saved)
    all_ok = bool(flag.item())
    if rank == 0:
        print(json.dumps({"ok": all_ok, "ranks": world, "rows_per_rank": rows, "bytes_per_rank": rows * 12, "allgather_us": us}), flush=True)
    disarm()
    sys.exit(0 if all_ok else 1)


if __name__ == "__main__":
    main()
