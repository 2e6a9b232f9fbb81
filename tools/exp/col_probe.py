"""Experiment: the TILE kernel's two drivers on the plain 6-neighbour star - column pipelines in ONE launch
(TTSWEEP_OPT_ASYNC = -1 / 1; mode 2: with TTSWEEP_OPT_TILE_IN_PLACE = 0) against one launch per tile hyperplane (0): same boxes bit for bit, time, work.
python tools/exp/col_probe.py nx,ny,nz nstart [reps] [modes e.g. 1,0]
ORDERS=0,2,5: the one-launch modes once per sequence of orderings (TTSWEEP_OPT_TILE_ORDER), digests compared across all"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ttsweep_pkg
P = ttsweep_pkg.load()
if os.environ.get("TTSWEEP_LIB"):       # another build of the same sources (e.g. gpurun_exp/colprof.so: -DTTSWEEP_COL_PROFILE)
    P._lib.use_library(os.environ["TTSWEEP_LIB"])
shape = tuple(int(x) for x in sys.argv[1].split(","))
nstart = int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
modes = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 0]
dev = torch.device("cuda:0")
v = P.inputs.velocity_model_device(*shape, 20160507, dev)
fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("six")))
starts = P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path("111")), *shape)[:nstart]
if os.environ.get("START_K"):     # the starts at another depth (fraction of nz, e.g. 0.5) instead of on the face k = nz - 1
    starts = starts.copy(); starts[:, 2] = int(float(os.environ["START_K"]) * (shape[2] - 1))
cells = shape[0] * shape[1] * shape[2]
digest = {}
orders = [int(x) for x in os.environ.get("ORDERS", "-1").split(",")]     # (-1: the library's default)
for mode, order in [(m, o) for m in modes for o in (orders if m else orders[:1])]:
    with P.TravelTimeSolver(shape, fs) as sol:
        if order >= 0: sol.set_option(P.OPT_TILE_ORDER, order)
        if os.environ.get("QUEUES"): sol.set_option(P.OPT_QUEUES, int(os.environ["QUEUES"]))      # claim sequences
        sol.set_option(P.OPT_ASYNC, 1 if mode == 2 else mode)     # (mode 2: columns, in the library's padded volumes)
        if mode == 2:
            sol.set_option(P.OPT_TILE_IN_PLACE, 0)
        sol.set_option(P.OPT_TIMING, 1)
        sol.set_velocity(v)
        tt = torch.empty((nstart,) + shape, dtype=torch.float32, device=dev)
        for rep in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = sol.solve_device(starts, tt, init=True)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            st = sol.stats()
            gb = 12.0 * st["cells_relaxed"] / 1e9
            print(f"mode {mode} order {order} {shape} x {nstart}: rc {rc} wall {wall:.2f} ms solve {st['solve_ms']:.2f} ms kernel {st['sweep_kernel_ms']:.2f} ms "
                  f"launches {st['launches']} sweeps max {st['sweeps_max']} total {st['sweeps_total']} grid-eq {st['cells_relaxed'] / cells / nstart:.2f} "
                  f"algorithmic {gb / (max(st['sweep_kernel_ms'], 1e-9) / 1e3):.0f} GB/s over kernel, {gb / (st['solve_ms'] / 1e3):.0f} GB/s over solve, "
                  f"fallbacks {st['fallbacks']}", flush=True)
        rc2 = sol.solve_device(starts, tt, init=False)
        print(f"mode {mode}: second solve rc {rc2} fallbacks {sol.stats()['fallbacks']}", flush=True)
        digest[(mode, order)] = [int(tt[s].view(torch.int32).to(torch.int64).sum().item()) for s in range(nstart)]
        if nstart <= 2 and cells <= 600 * 600 * 300:
            print("validate", [sol.validate_device(starts[s], tt[s]) for s in range(nstart)], flush=True)
if len(digest) > 1:
    print("digests equal:", all(d == next(iter(digest.values())) for d in digest.values()))
