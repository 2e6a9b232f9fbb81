"""CPU tier: the N > 1 path (start sharding + gather) with world_size 2 and 3 over gloo.
The per-rank solver is injected; here it is the CPU oracle (checker role only), so
the test exercises exactly the sharding / gather code bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, nstart, outdir, path="device", balanced=False, one_host=True):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import ttsweep_pkg
    P = ttsweep_pkg.load()
    shape = (10, 9, 6)
    v = P.inputs.velocity_model(*shape, seed=21)
    offs = P.inputs.read_triples(P.inputs.star_path("3"))
    fs = O.make_star(offs)
    rng = np.random.default_rng(5)
    starts = np.stack([rng.integers(0, n, size=nstart) for n in shape], axis=1).astype(np.int32)

    def solve_fn(my_starts):
        boxes = [O.converge(v, fs, st, order=1)[0] for st in my_starts]
        if not boxes:
            return torch.empty((0,) + shape, dtype=torch.float32)
        return torch.from_numpy(np.stack(boxes))

    if not one_host:            # play "ranks on different machines": CPU tensors over the process group
        P.multistart._same_host = lambda d: False
    allb, local = P.multistart.solve_sharded(starts, solve_fn, dist, dst=0, shape=shape if balanced else None,
                                             path=path)
    assert local.shape[0] == len(P.multistart.shard_starts(nstart, world, rank, starts if balanced else None,
                                                           shape if balanced else None))
    if rank == 0:
        assert not allb.is_cuda and tuple(allb.shape) == (nstart,) + shape
        np.save(os.path.join(outdir, "gathered.npy"), allb.numpy())
        np.save(os.path.join(outdir, "starts.npy"), starts)
    else:
        assert allb is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nstart,path,balanced,one_host", [
    (2, 5, "device", False, True),      # unequal shards (3 + 2), boxes straight into their slots
    (3, 4, "device", True, True),       # cost-balanced shards (not round-robin), 2 + 1 + 1
    (2, 1, "device", False, True),      # a rank without any start
    (2, 5, "host", True, True),         # forced host gather: one shared-memory array all ranks write
    (3, 7, "host", False, True),
    (2, 5, "host", False, False),       # host gather between machines: CPU tensors, point to point
])
def test_sharded_solve_gathers_in_start_order(tmp_path, oracle, pkg, world, nstart, path, balanced, one_host):
    port = free_port()
    mp.spawn(worker, args=(world, port, nstart, str(tmp_path), path, balanced, one_host), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    starts = np.load(tmp_path / "starts.npy")
    assert got.shape[0] == nstart
    v = pkg.inputs.velocity_model(10, 9, 6, seed=21)
    fs = oracle.make_star(pkg.inputs.read_triples(pkg.inputs.star_path("3")))
    for s in range(nstart):
        want, _, _ = oracle.converge(v, fs, starts[s])
        assert np.array_equal(got[s].view(np.uint32), want.view(np.uint32)), s


def group_worker(rank, world, port, port2, nstart, outdir):
    """The process-group set-up of bench.py (round 5): the DEFAULT group carries the control traffic, the device
    transfers of the gather run in a group of their own (there: RCCL beside a gloo default group; here: a second gloo
    group), and every rank learns of a failed probe over the default group before it depends on that group."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ttsweep_pkg
    P = ttsweep_pkg.load()
    data = dist.new_group(backend="gloo")
    # the agreement: one rank's probe "fails" - every rank hears of it
    errs = [None] * world
    dist.all_gather_object(errs, "probe failed on rank 1" if rank == 1 else None)
    assert [e for e in errs if e] == ["probe failed on rank 1"]
    shape = (4, 3, 5)
    shards = P.multistart.all_shards(nstart, world)
    mine = shards[rank]
    local = torch.stack([torch.full(shape, float(s)) for s in mine]) if mine else torch.empty((0,) + shape)
    out = P.multistart.gather_boxes(local, nstart, dist, dst=0, shards=shards, path="device", group=data)
    if rank == 0:
        np.save(os.path.join(outdir, "gathered.npy"), out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()
    # a second world in the same process: a cached helper group of the first one must not be used again
    os.environ["MASTER_PORT"] = str(port2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert P.multistart._host_group(dist) is None
    out = P.multistart.gather_boxes(local, nstart, dist, dst=0, shards=shards, path="host", tag=f"ttsweep_t{port}")
    if rank == 0:
        np.save(os.path.join(outdir, "gathered2.npy"), out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_in_a_group_of_its_own_and_after_a_new_world(tmp_path, pkg):
    world, nstart = 3, 7
    port, port2 = free_port(), free_port()
    mp.spawn(group_worker, args=(world, port, port2, nstart, str(tmp_path)), nprocs=world, join=True)
    for name in ("gathered.npy", "gathered2.npy"):
        got = np.load(tmp_path / name)
        assert got.shape == (nstart, 4, 3, 5)
        for s in range(nstart):
            assert (got[s] == float(s)).all(), (name, s)


def test_shard_assignment_is_a_partition(pkg):
    for nstart in (0, 1, 4, 24, 111):
        for world in (1, 2, 4, 8):
            shards = [pkg.multistart.shard_starts(nstart, world, r) for r in range(world)]
            flat = sorted(s for sh in shards for s in sh)
            assert flat == list(range(nstart))
            assert pkg.multistart.shard_sizes(nstart, world) == [len(sh) for sh in shards]
            assert max(map(len, shards)) - min(map(len, shards)) <= 1


def test_cost_balanced_shards(pkg):
    """With coordinates the starts are dealt longest-first by estimated cost: still a
    partition with balanced counts, the same on every rank, and the heaviest shard is no
    heavier than under round-robin (start-24 on 241x241x51 over 8 ranks: 3 starts each)."""
    M = pkg.multistart
    shape = (241, 241, 51)
    starts = pkg.inputs.read_triples(pkg.inputs.starts_path("24"))
    cost = [M.start_cost(s, shape) for s in starts]
    assert M.start_cost((0, 0, 50), shape) > M.start_cost((120, 120, 50), shape)
    for world in (1, 2, 4, 8, 5):
        shards = M.all_shards(len(starts), world, starts, shape)
        assert sorted(s for sh in shards for s in sh) == list(range(len(starts)))
        assert max(map(len, shards)) <= -(-len(starts) // world)
        assert shards == M.all_shards(len(starts), world, starts, shape)
        assert [M.shard_starts(len(starts), world, r, starts, shape) for r in range(world)] == shards
        heaviest = max(sum(cost[s] for s in sh) for sh in shards)
        rr = max(sum(cost[s] for s in sh) for sh in M.all_shards(len(starts), world))
        assert heaviest <= rr + 1e-9


# --------------------------------------------------------------------------
# one start on several ranks: star split + all-reduce(min)
# --------------------------------------------------------------------------

def split_worker(rank, world, port, star, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import ttsweep_pkg
    P = ttsweep_pkg.load()
    shape = (12, 10, 7)
    v = P.inputs.velocity_model(*shape, seed=33)
    fs = O.make_star(P.inputs.read_triples(P.inputs.star_path(star)))
    start = (3, 8, 6)
    lo, hi = P.multistart.star_slices(len(fs) - 1, world)[rank]

    def slice_solve(box):
        tt = box.numpy()                                    # shares memory: relaxed in place
        if hi <= lo:
            return False
        _, _, stores = O.converge(v, fs, start, starstart=lo, starstop=hi, tt=tt)
        return stores > 0

    box = torch.from_numpy(O.tt_init(shape, start))
    rounds = P.multistart.solve_star_split(box, slice_solve, dist)
    np.save(os.path.join(outdir, f"split_{rank}.npy"), box.numpy())
    if rank == 0:
        np.save(os.path.join(outdir, "rounds.npy"), np.array([rounds]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,star", [(2, "3"), (3, "5")])
def test_star_split_reaches_the_full_star_fixed_point(tmp_path, oracle, pkg, world, star):
    """Every rank relaxes only its slice of the offsets; with an all-reduce(min) per round
    all ranks end with the box the whole star converges to, bit for bit."""
    port = free_port()
    mp.spawn(split_worker, args=(world, port, star, str(tmp_path)), nprocs=world, join=True)
    shape = (12, 10, 7)
    v = pkg.inputs.velocity_model(*shape, seed=33)
    fs = oracle.make_star(pkg.inputs.read_triples(pkg.inputs.star_path(star)))
    want, _, _ = oracle.converge(v, fs, (3, 8, 6))
    for r in range(world):
        got = np.load(tmp_path / f"split_{r}.npy")
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), r
    assert int(np.load(tmp_path / "rounds.npy")[0]) >= 2


def test_star_slices_partition_the_offsets(pkg):
    for n, k in [(817, 8), (97, 3), (5, 8), (0, 2), (1, 1)]:
        sl = pkg.multistart.star_slices(n, k)
        assert len(sl) == k and sl[0][0] == 0 and sl[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        assert max(h - l for l, h in sl) - min(h - l for l, h in sl) <= 1


def test_gather_plan_picks_the_path_from_the_sizes(pkg):
    """Device gather while the result set fits half of the root's free HBM, host beyond
    (1024x1024x512 x 111 starts = 238 GB cannot be gathered on one 288 GB GPU)."""
    M = pkg.multistart
    box241, box1024 = 241 * 241 * 51 * 4, 1024 * 1024 * 512 * 4
    assert M.plan_gather(24, box241, 280 * 10**9)["path"] == "device"
    assert M.plan_gather(8, 512 * 512 * 256 * 4, 280 * 10**9)["path"] == "device"
    big = M.plan_gather(111, box1024, 219 * 10**9)
    assert big["path"] == "host" and big["bytes"] == 111 * box1024
    assert M.plan_gather(14, box1024, 219 * 10**9)["path"] == "device"
    assert M.plan_gather(3, box241, None)["path"] == "host"
