"""Multi-GPU multi-start driver: shard independent start points over ranks.

Each start point owns a private travel-time box and only reads the shared
velocity volume and star (serial_new/sweep-tt-multistart.c:158-162: the `s`
loop has no cross-iteration dependence; mpi/backup.c:351-363 already runs one
start per rank), so starts are the unit of distribution: one process per GPU,
no communication while sweeping, and one final gather of the per-start boxes to
rank 0 - the step the reference left as a TODO (mpi/backup.c:381-386).  With
backend "nccl" (RCCL on ROCm) the gather moves device buffers over xGMI, every box
straight into its slot of the result; a result set that does not fit the root's GPU
(1024x1024x512 x 111 starts) is gathered into host memory instead (plan_gather).

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


# What share of the root's free device memory the gathered result set may take before the
# gather goes to host memory instead (the root also holds its own shard, the padded volumes
# of the solver and the velocity).
DEVICE_GATHER_FRACTION = 0.5


def plan_gather(nstart: int, box_bytes: int, free_device_bytes: int | None,
                fraction: float = DEVICE_GATHER_FRACTION) -> dict:
    """Where the final gather of `nstart` boxes of `box_bytes` each goes:
    "device" - rank `dst` receives every remote box straight into its slot of one device
    array (RCCL send / recv over xGMI) - while the whole set fits `fraction` of the root's free
    device memory; "host" beyond that (1024x1024x512 x 111 starts = 238 GB do not fit one
    288 GB GPU next to the solver's own volumes): every rank copies its boxes from its GPU into
    ONE array in host memory.  `free_device_bytes` None (no device: CPU rehearsal) -> "host"."""
    total = int(nstart) * int(box_bytes)
    if free_device_bytes is None:
        return {"path": "host", "bytes": total, "why": "no device memory to gather into"}
    budget = int(fraction * free_device_bytes)
    if total <= budget:
        return {"path": "device", "bytes": total,
                "why": f"{total / 1e9:.2f} GB <= {fraction:.2f} x {free_device_bytes / 1e9:.1f} GB free on the root"}
    return {"path": "host", "bytes": total,
            "why": f"{total / 1e9:.2f} GB > {fraction:.2f} x {free_device_bytes / 1e9:.1f} GB free on the root"}


def _same_host(dist) -> bool:
    """Do all ranks run on one machine (one node: the case the north star names)?"""
    import socket
    names = [None] * dist.get_world_size()
    dist.all_gather_object(names, socket.gethostname(), group=_host_group(dist))
    return len(set(names)) == 1


_cpu_group = None           # (default group it was made for, gloo group)


def _host_group(dist):
    """A process group that can move CPU tensors: the default one unless it is RCCL-only.  The gloo group made
    for an RCCL-only default group is kept for as long as THAT default group lives: after destroy_process_group
    and a new init (bench.py's fall-back to gloo, tests that make a one-rank nccl group) the cached handle would
    belong to the destroyed world."""
    global _cpu_group
    if "gloo" in str(dist.get_backend()):
        return None                                 # (the default group does)
    world = dist.group.WORLD
    if _cpu_group is None or _cpu_group[0] is not world:
        _cpu_group = (world, dist.new_group(backend="gloo"))
    return _cpu_group[1]


def gather_boxes(local, nstart: int, dist=None, dst: int = 0, shards=None, path: str = "device",
                 shm_dir: str = "/dev/shm", tag: str = "ttsweep", loopback: bool = False, group=None):
    """Gather the per-rank stacks of boxes [n_local, nx, ny, nz] on rank `dst`, ordered by
    global start index ([nstart, nx, ny, nz]); other ranks return None.  Ranks may hold
    different numbers of starts (`shards`: the assignment in use, all_shards; round-robin when
    omitted).  This is the step the reference left as a TODO (mpi/backup.c:381-386).

    path "device": grouped point-to-point transfers (batch_isend_irecv: ncclSend / ncclRecv in
        one group under backend "nccl", i.e. RCCL over xGMI; isend / irecv under gloo) of every
        box STRAIGHT INTO its slot of the result - no padding to the largest shard, no staging
        stacks, no reorder copy: the root holds the result set once, next to its own shard.
    path "host": the result lives in host memory.  All ranks on one machine: rank `dst` creates
        one array in shared memory (a file under `shm_dir`), every rank maps it and copies its
        boxes from its GPU straight into their slots (D2H over PCIe, all GPUs at once, no
        inter-process copy); ranks on different machines: the boxes travel as CPU tensors over
        a gloo group.  The returned tensor is a CPU tensor (shared mapping or plain).
    group (path "device"): the process group the device transfers run in - an RCCL group next to a gloo default group
        (bench.py: the default group carries the control traffic, RCCL only the gather); None: the default group.
    loopback (testing aid, path "device"): the root's own boxes travel like everybody else's - a send to
        itself and the matching receive in the same group -, also in a group of ONE rank: the collective
        path then runs on a single GPU (a one-rank `nccl` group is the only RCCL a one-GPU box offers)."""
    import torch

    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not loopback):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if shards is None:
        shards = all_shards(nstart, world)
    box_shape = tuple(local.shape[1:])
    mine = shards[rank]
    if local.shape[0] != len(mine):
        raise ValueError(f"rank {rank} holds {local.shape[0]} boxes, its shard has {len(mine)}")
    local = local.contiguous()

    if path == "device":
        out = None
        ops = []
        if rank == dst:
            out = torch.empty((nstart,) + box_shape, dtype=local.dtype, device=local.device)
            if loopback:
                ops += [dist.P2POp(dist.isend, local[n], dst, group) for n in range(len(mine))]
            else:
                for n, s in enumerate(mine):
                    out[s].copy_(local[n])
            for r in range(world):
                if r != dst or loopback:
                    ops += [dist.P2POp(dist.irecv, out[s], r, group) for s in shards[r]]
        else:
            ops = [dist.P2POp(dist.isend, local[n], dst, group) for n in range(len(mine))]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out

    if path != "host":
        raise ValueError(f"unknown gather path {path!r}")
    numel = int(np.prod(box_shape)) if box_shape else 1
    if _same_host(dist):
        import os
        if local.dtype != torch.float32:
            raise ValueError(f"the shared-memory gather holds float32 boxes, not {local.dtype}")
        name = [None]
        if rank == dst:
            name[0] = os.path.join(shm_dir, f"{tag}_{os.getpid()}_{nstart}x{numel}.f32")
            with open(name[0], "wb") as f:
                f.truncate(max(nstart * numel, 1) * 4)
        ctl = _host_group(dist)                     # (the host path does not depend on RCCL: its hand-shakes run over gloo)
        try:
            dist.broadcast_object_list(name, src=dst, group=ctl)
            out = torch.from_file(name[0], shared=True, size=max(nstart * numel, 1), dtype=torch.float32)
            dist.barrier(group=ctl)                 # every rank has mapped the file:
        finally:
            # ... the name can go - a rank that dies from here on leaves no file (up to 238 GB of host
            # memory for BASELINE config 5) behind; the mappings keep the pages until they are dropped
            if rank == dst and name[0] is not None and os.path.exists(name[0]):
                os.unlink(name[0])
        out = out[: nstart * numel].view((nstart,) + box_shape)
        for n, s in enumerate(mine):
            out[s].copy_(local[n])                  # device -> the shared host array (or host -> host)
        if local.is_cuda:
            torch.cuda.synchronize(local.device)
        dist.barrier(group=ctl)                     # every slot is written
        return out if rank == dst else None
    group = _host_group(dist)
    out = None
    ops = []
    if rank == dst:
        out = torch.empty((nstart,) + box_shape, dtype=local.dtype)
        for n, s in enumerate(mine):
            out[s].copy_(local[n])
        for r in range(world):
            if r != dst:
                ops += [dist.P2POp(dist.irecv, out[s], r, group) for s in shards[r]]
    else:
        host = local.cpu()
        ops = [dist.P2POp(dist.isend, host[n], dst, group) for n in range(len(mine))]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def solve_sharded(starts: Sequence, solve_fn: Callable, dist=None, dst: int = 0, shape=None,
                  path: str = "device"):
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
    return gather_boxes(local, len(starts), dist, dst, shards=shards, path=path), local


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


def solve_star_split(box, slice_solve_fn: Callable, dist=None, max_rounds: int = 100000, collectives: bool = False):
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
    single-GPU solve needs passes.  collectives (testing aid): run the two all-reduces also in a
    group of ONE rank (the only RCCL a one-GPU box offers)."""
    import torch

    multi = dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or collectives)
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
