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


def gather_scores(local: torch.Tensor, n_streams: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """local: fp32 [F, S_local, 3] rows of this rank's streams (in streams_of_rank order).
    Returns fp32 [F, n_streams, 3] in global stream order on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        assert local.shape[1] == n_streams
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    F = local.shape[0]
    cap = max_streams_per_rank(n_streams, world)
    buf = local.new_zeros((F, cap, 3))
    buf[:, : local.shape[1]] = local                       # ragged ranks are padded to the common size
    out = local.new_empty((world, F, cap, 3))
    dist.all_gather_into_tensor(out.view(-1), buf.view(-1).contiguous(), group=group)
    glob = local.new_empty((F, n_streams, 3))
    for r in range(world):
        ids = streams_of_rank(n_streams, world, r)
        if ids:
            glob[:, ids] = out[r, :, : len(ids)]
    return glob
