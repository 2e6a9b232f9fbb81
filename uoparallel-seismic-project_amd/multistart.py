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


def start_cost(start, shape) -> float:
    """Estimated cost of one start: the distance (cells) from the start to the farthest
    corner of the grid.  The number of passes a solve needs grows with the longest shortest
    path, i.e. with this distance (starts of the benchmark workload differ 3.5x in the
    sweeps the reference itself needs: 22 to 76)."""
    return float(np.sqrt(sum(max(int(c), int(n) - 1 - int(c)) ** 2 for c, n in zip(start, shape))))


def all_shards(nstart: int, world_size: int, starts=None, shape=None) -> List[List[int]]:
    """Start indices per rank.  Without coordinates: round-robin (start s on rank
    s % world_size).  With them: longest-first by estimated cost onto the least loaded rank
    (LPT), every rank holding at most ceil(nstart / world_size) starts so that memory stays
    balanced; all ranks compute the same assignment from the same inputs."""
    if starts is None or shape is None or world_size <= 1:
        return [list(range(r, nstart, world_size)) for r in range(world_size)]
    cap = -(-nstart // world_size)
    cost = [start_cost(starts[s], shape) for s in range(nstart)]
    order = sorted(range(nstart), key=lambda s: (-cost[s], s))
    shards: List[List[int]] = [[] for _ in range(world_size)]
    load = [0.0] * world_size
    for s in order:
        r = min((r for r in range(world_size) if len(shards[r]) < cap), key=lambda r: (load[r], r))
        shards[r].append(s)
        load[r] += cost[s]
    return [sorted(sh) for sh in shards]


def shard_starts(nstart: int, world_size: int, rank: int, starts=None, shape=None) -> List[int]:
    """This rank's start indices (see all_shards)."""
    return all_shards(nstart, world_size, starts, shape)[rank]


def shard_sizes(nstart: int, world_size: int) -> List[int]:
    return [len(range(r, nstart, world_size)) for r in range(world_size)]


def gather_boxes(local, nstart: int, dist=None, dst: int = 0, shards=None):
    """Gather the per-rank stacks of boxes [n_local, nx, ny, nz] on rank `dst` and
    return them ordered by global start index ([nstart, nx, ny, nz]); other ranks
    return None.  Ranks may hold different numbers of starts; stacks are padded to
    the largest shard for the collective.  `shards`: the assignment in use (all_shards);
    round-robin when omitted."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if shards is None:
        shards = all_shards(nstart, world)
    nmax = max(len(sh) for sh in shards)
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
            for n, s in enumerate(shards[r]):
                out[s] = recv[r][n]
        return out
    dist.gather(send, gather_list=None, dst=dst)
    return None


def solve_sharded(starts: Sequence, solve_fn: Callable, dist=None, dst: int = 0, shape=None):
    """Solve this rank's shard with `solve_fn(list_of_starts) -> tensor[n_local,...]`
    and gather all boxes on rank `dst`.  Returns (all_boxes_or_None, local_boxes).  With
    the grid `shape` the shards are balanced by estimated cost, else dealt round-robin."""
    starts = np.asarray(starts, dtype=np.int32).reshape(-1, 3)
    if dist is None or not dist.is_initialized():
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(), dist.get_rank()
    shards = all_shards(len(starts), world, starts if shape is not None else None, shape)
    local = solve_fn(starts[shards[rank]])
    return gather_boxes(local, len(starts), dist, dst, shards=shards), local


# --------------------------------------------------------------------------
# one start on several GPUs: star split + all-reduce(min)
# --------------------------------------------------------------------------

def star_slices(noffsets: int, nslice: int):
    """Contiguous, nearly equal slices [lo, hi) of the offsets l in [0, noffsets) - the
    range sweepXYZ walks, serial_new/sweep-tt-multistart.c:160,206 (noffsets =
    starsize - 1: the last star entry is never an offset).  Empty slices are kept so
    that every rank has one."""
    base, extra = divmod(max(noffsets, 0), nslice)
    out, lo = [], 0
    for r in range(nslice):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def solve_star_split(box, slice_solve_fn: Callable, dist=None, max_rounds: int = 100000):
    """One start point on all ranks (precedent: cuda/cudasweep-tt-multistart.cu:316-384, the
    reference's star split with a per-sweep reduction).  Every rank holds the same box
    `box` (torch tensor [nx,ny,nz]: INFINITY, start 0 - or any later state) and relaxes
    only ITS slice of the star: `slice_solve_fn(box) -> bool` brings the box, in place, to
    the fixed point of the rank's offsets and says whether anything changed.  After each
    round the boxes are combined with an element-wise all-reduce(min) (RCCL over xGMI for
    backend "nccl"); the loop ends after a round in which no rank changed anything - the
    box is then a fixed point of every slice, i.e. of the whole star, and since the slices'
    edge sets add up to the reference's edge set it is bit-identical to the single-GPU
    result.  Returns the number of rounds.

    min is associative, commutative and idempotent, so the order of the reduction does not
    matter.  This is a correctness path for runs with fewer starts than GPUs, not a fast
    one: shortest paths alternate between offsets of different slices at almost every hop,
    so it needs about as many rounds (each with a reduction of the whole box) as the
    single-GPU solve needs passes."""
    import torch

    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    rounds = 0
    while True:
        rounds += 1
        if rounds > max_rounds:
            raise RuntimeError("solve_star_split: no convergence")
        changed = bool(slice_solve_fn(box))
        if not multi:
            if not changed:
                return rounds
            continue
        flag = torch.tensor([1 if changed else 0], dtype=torch.int32, device=box.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) == 0:
            return rounds
        dist.all_reduce(box, op=dist.ReduceOp.MIN)


def solve_star_split_local(box, slice_solve_fns: Sequence[Callable], max_rounds: int = 100000):
    """The same iteration with the "ranks" played one after the other in this process
    (tests, and hosts that drive several contexts themselves): every round each slice
    solver starts from the round's common box, then the results are min-combined."""
    import torch

    rounds = 0
    while True:
        rounds += 1
        if rounds > max_rounds:
            raise RuntimeError("solve_star_split_local: no convergence")
        results, any_changed = [], False
        for fn in slice_solve_fns:
            mine = box.clone()
            any_changed |= bool(fn(mine))
            results.append(mine)
        if not any_changed:
            return rounds
        combined = results[0]
        for r in results[1:]:
            combined = torch.minimum(combined, r)
        box.copy_(combined)
