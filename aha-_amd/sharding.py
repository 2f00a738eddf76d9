"""Multi-GPU: the path shards by video stream (streams never exchange arithmetic).
One process per GPU, full model replica per rank, stream i -> rank i % world.  The only
collective is an all-gather of the per-step score rows (fp32 [frames, streams_local, 3]; a few
hundred bytes per rank), issued with torch.distributed: backend "nccl" is RCCL over xGMI on
ROCm, "gloo" serves the CPU tests.  No reduction anywhere; the result is laid out in GLOBAL
stream order so every rank sees the same [frames, n_streams, 3] tensor.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def streams_of_rank(n_streams: int, world: int, rank: int) -> List[int]:
    """Round-robin partition: global stream ids this rank owns, ascending."""
    return list(range(rank, n_streams, world))


def max_streams_per_rank(n_streams: int, world: int) -> int:
    return (n_streams + world - 1) // world


class ScoreGather:
    """An all-gather of score rows in flight.  `result()` makes the CURRENT stream (nccl) or the host (gloo) wait for it and
    returns fp32 [F, n_streams, 3] in global stream order; until then the caller's stream keeps running (the
    collective of step k overlaps the LM steps of step k+1)."""

    def __init__(self, work, out, local, n_streams, world):
        self.work, self.out, self.local, self.n_streams, self.world = work, out, local, n_streams, world
        self.glob = None

    def result(self) -> torch.Tensor:
        if self.glob is None:
            if self.work is not None:
                self.work.wait()
                self.work = None
            if self.out is None:                              # no process group: the local rows are the global rows
                self.glob = self.local
            else:
                F = self.out.shape[1]
                glob = self.out.new_empty((F, self.n_streams, 3))
                for r in range(self.world):
                    ids = streams_of_rank(self.n_streams, self.world, r)
                    if ids:
                        glob[:, ids] = self.out[r, :, : len(ids)]
                self.glob = glob
            self.out = self.local = None
        return self.glob


def gather_scores_async(local: torch.Tensor, n_streams: int, group: Optional[dist.ProcessGroup] = None) -> ScoreGather:
    """Starts the all-gather of this rank's rows (fp32 [F, S_local, 3], in streams_of_rank order) and returns its handle.  The
    rows are copied into a private padded buffer first, so `local` may be overwritten at once."""
    if not (dist.is_available() and dist.is_initialized()):
        assert local.shape[1] == n_streams
        return ScoreGather(None, None, local, n_streams, 1)
    world = dist.get_world_size(group)
    F = local.shape[0]
    cap = max_streams_per_rank(n_streams, world)
    buf = local.new_zeros((F, cap, 3))
    buf[:, : local.shape[1]] = local                       # ragged ranks are padded to the common size
    out = local.new_empty((world, F, cap, 3))
    work = dist.all_gather_into_tensor(out.view(-1), buf.view(-1).contiguous(), group=group, async_op=True)
    return ScoreGather(work, out, None, n_streams, world)


def gather_scores(local: torch.Tensor, n_streams: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """local: fp32 [F, S_local, 3] rows of this rank's streams (in streams_of_rank order).
    Returns fp32 [F, n_streams, 3] in global stream order on every rank."""
    return gather_scores_async(local, n_streams, group).result()
