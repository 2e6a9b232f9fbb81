"""Multi-GPU multi-start driver: shard independent start points over ranks.

Each start point owns a private travel-time box and only reads the shared
velocity volume and star (serial_new/sweep-tt-multistart.c:158-162: the `s`
loop has no cross-iteration dependence; mpi/backup.c:351-363 already runs one
start per rank), so starts are the unit of distribution: one process per GPU,
no communication while sweeping, and one final gather of the per-start boxes to
rank 0 - the step the reference left as a TODO (mpi/backup.c:381-386).  With
backend "nccl" (RCCL on ROCm) the gather moves device buffers over xGMI.

The solver is injected (`solve_fn`) so the same sharding/gather code runs under
the CPU test tier with backend "gloo".
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np


def shard_starts(nstart: int, world_size: int, rank: int) -> List[int]:
    """Round-robin assignment: start s goes to rank s % world_size."""
    return list(range(rank, nstart, world_size))


def shard_sizes(nstart: int, world_size: int) -> List[int]:
    return [len(range(r, nstart, world_size)) for r in range(world_size)]


def gather_boxes(local, nstart: int, dist=None, dst: int = 0):
    """Gather the per-rank stacks of boxes [n_local, nx, ny, nz] on rank `dst` and
    return them ordered by global start index ([nstart, nx, ny, nz]); other ranks
    return None.  Ranks may hold different numbers of starts; stacks are padded to
    the largest shard for the collective."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = shard_sizes(nstart, world)
    nmax = max(sizes)
    box_shape = tuple(local.shape[1:])
    send = local
    if local.shape[0] < nmax:
        send = torch.zeros((nmax,) + box_shape, dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    send = send.contiguous()
    if rank == dst:
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=recv, dst=dst)
        out = torch.empty((nstart,) + box_shape, dtype=local.dtype, device=local.device)
        for r in range(world):
            for n, s in enumerate(shard_starts(nstart, world, r)):
                out[s] = recv[r][n]
        return out
    dist.gather(send, gather_list=None, dst=dst)
    return None


def solve_sharded(starts: Sequence, solve_fn: Callable, dist=None, dst: int = 0):
    """Solve this rank's shard with `solve_fn(list_of_starts) -> tensor[n_local,...]`
    and gather all boxes on rank `dst`.  Returns (all_boxes_or_None, local_boxes)."""
    starts = np.asarray(starts, dtype=np.int32).reshape(-1, 3)
    if dist is None or not dist.is_initialized():
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(), dist.get_rank()
    mine = shard_starts(len(starts), world, rank)
    local = solve_fn(starts[mine])
    return gather_boxes(local, len(starts), dist, dst), local
