#!/usr/bin/env python3
"""Benchmark of the multi-start travel-time sweep (the hot path of
serial_new/sweep-tt-multistart.c) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric config): synthetic 241x241x51 velocity volume
(the real model is absent from the reference tree), the reference's default
818-offset star (docs/818-FS.txt, serial_new/Makefile:15) and the 24 start points
of docs/start-24-241-241-51.txt.  One "step" = one complete multi-start solve:
initialise every travel-time box on the device (all INFINITY, start 0), relax to
convergence, and - for N > 1 - gather all boxes on rank 0 over RCCL.  The starts
are sharded round-robin over the N ranks (strong scaling: the 24 starts are the
fixed total work); inputs are resident in HBM when the timed region starts.

metric: Mcells*sweeps/s = (cells relaxed against the whole forward star, summed
over passes and starts) / wall seconds / 1e6; one full sweep of one start relaxes
`cells` cells, so for a schedule that skips nothing this is cells x sweeps / s.
The GPU schedule skips units whose inputs did not change and holds far units back until
final values can reach them, so it executes fewer cell-relaxations than full sweeps
would; those skipped cells are NOT counted.
Passes of different schedules are the same work per cell but not the same
progress, so ms_per_step (time to the converged solution) is the number to
compare across schedules and against the CPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TLANEOPS = 78.6           # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz (SIMD-32)
BYTES_PER_CELL_SWEEP = 12           # read v, read tt, write tt
LANEOPS_PER_RELAX = 4               # add, mul, add, min


def cpu_baseline(P, v, offs, start, sweeps=4):
    """The CPU restatement of serial_new (oracle/, kind "port") timed on this
    host: `sweeps` reference-order passes of one start from the initial state."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    fs = O.make_star(offs)
    tt = O.tt_init(v.shape, start)
    t0 = time.perf_counter()
    for _ in range(sweeps):
        O.sweep(v, tt, fs, start)
    dt = time.perf_counter() - t0
    return {"value": v.size * sweeps / dt / 1e6, "unit": "Mcells*sweeps/s", "cores": 1,
            "kind": "port", "seconds": round(dt, 2),
            "sample": f"{sweeps} reference-order passes of start 0 on the same grid/star "
                      f"(single thread, gcc -O3; host has {os.cpu_count()} logical cores)"}


def cpu_baseline_cores(shape, star, starts_name, ncores, sweeps=1):
    """The same CPU restatement with one start per core (the strategy of the reference's
    mpi/backup.c:351-363), `ncores` processes side by side, each `sweeps` reference-order
    passes of its own start.  Child processes: they never touch the GPU."""
    import subprocess
    script = os.path.join(ROOT, "oracle", "cpu_sample.py")
    args = [str(n) for n in shape] + [star, starts_name]
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, script] + args + [str(i), str(sweeps)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for i in range(ncores)]
    secs = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode == 0:
            secs.append(float(out.strip().splitlines()[-1]))
    wall = time.perf_counter() - t0
    if len(secs) != ncores:
        return None
    cells = shape[0] * shape[1] * shape[2]
    # rate while all cores sweep together: each finished `sweeps` passes in its own time
    return {"value": sum(cells * sweeps / t for t in secs) / 1e6, "unit": "Mcells*sweeps/s",
            "cores": ncores, "kind": "port", "seconds": round(max(secs), 2),
            "wall_seconds_incl_startup": round(wall, 2),
            "sample": f"{sweeps} reference-order pass(es) of {ncores} different starts, one process per "
                      f"core, side by side (host has {os.cpu_count()} logical cores)"}


def measured_traffic():
    """HBM-side bytes per sweep-kernel launch from the rocprofv3 PMC passes of this same
    command (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 x2 read correction),
    committed under profiles/; bench.py cannot run a profiler around itself."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))["bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", default="241,241,51")
    ap.add_argument("--star", default="818")
    ap.add_argument("--starts", default="24")
    ap.add_argument("--nstarts", type=int, default=0, help="use only the first N start points")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                         "the multi-rank path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    import torch
    import ttsweep_pkg
    P = ttsweep_pkg.load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    default_workload = (args.grid, args.star, args.starts, args.nstarts, args.kernel) == \
        ("241,241,51", "818", "24", 0, 0) and world == 1
    nx, ny, nz = map(int, args.grid.split(","))
    cells = nx * ny * nz
    offs = P.inputs.read_triples(P.inputs.star_path(args.star))
    fs = P.inputs.make_fs(offs)
    starts = P.inputs.read_triples(P.inputs.starts_path(args.starts))
    if args.nstarts:
        starts = starts[:args.nstarts]
    if (nx, ny, nz) != (241, 241, 51):
        starts = P.inputs.scaled_starts(starts, nx, ny, nz)
        v_dev = P.inputs.velocity_model_device(nx, ny, nz, 20160507, dev)
        v_host = None
    else:
        v_host = P.inputs.velocity_model(nx, ny, nz, 20160507)
        v_dev = torch.from_numpy(v_host).to(dev)
    nstart = len(starts)
    mine = P.multistart.shard_starts(nstart, world, rank)
    my_starts = starts[mine]

    sol = P.TravelTimeSolver((nx, ny, nz), fs, device=dev_index)
    if args.kernel:
        sol.set_option(P.OPT_KERNEL, args.kernel)
    sol.set_velocity(v_dev)
    tt = torch.empty((len(mine), nx, ny, nz), dtype=torch.float32, device=dev)

    def step():
        sol.solve_device(my_starts, tt, init=True)
        if dist is not None and args.backend != "nccl":     # rehearsal: gather through the host
            return P.multistart.gather_boxes(tt.cpu(), nstart, dist, dst=0)
        return P.multistart.gather_boxes(tt, nstart, dist, dst=0)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    sweeps_local = 0            # passes launched (a pass relaxes only the units that are due)
    relaxed_local = 0           # cells actually relaxed against the whole star
    for _ in range(args.steps):
        step()
        st_ = sol.stats()
        sweeps_local += st_["sweeps_total"]
        relaxed_local += st_["cells_relaxed"]
    fence()
    dt = time.perf_counter() - t0

    # max over ranks of the elapsed time, sum over ranks of the passes executed
    agg = torch.tensor([dt, float(sweeps_local), float(relaxed_local)], dtype=torch.float64,
                       device=dev if args.backend == "nccl" else "cpu")
    if dist is not None:
        tmax = agg[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ssum = agg[1:].clone()
        dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
        dt, sweeps_all, relaxed_all = float(tmax.item()), float(ssum[0].item()), float(ssum[1].item())
    else:
        sweeps_all, relaxed_all = float(sweeps_local), float(relaxed_local)

    # one extra, instrumented solve: HIP events around every sweep launch on the
    # library's own stream (not part of the timed region)
    sol.set_option(P.OPT_TIMING, 1)
    sol.solve_device(my_starts, tt, init=True)
    st = sol.stats()
    sol.set_option(P.OPT_TIMING, 0)

    if rank == 0:
        kern_s = st["sweep_kernel_ms"] / 1e3
        launches = max(st["launches"], 1)
        alg_bytes = BYTES_PER_CELL_SWEEP * st["cells_relaxed"]
        relax = st["relaxations_per_sweep"] * (st["cells_relaxed"] / cells)
        achieved = alg_bytes / kern_s / 1e9 if kern_s > 0 else 0.0
        lane = LANEOPS_PER_RELAX * relax / kern_s / 1e12 if kern_s > 0 else 0.0
        out = {
            "metric": "Mcells*sweeps/s (241x241x51, 818-offset star, 24 starts, to convergence)",
            "value": relaxed_all / dt / 1e6,
            "unit": "Mcells*sweeps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{nx}x{ny}x{nz} synthetic velocity, {args.star}-FS star, "
                                   f"{nstart} starts (start-{args.starts}), converged multi-start solve",
                       "grid": [nx, ny, nz], "star_offsets": int(len(offs)), "starts": int(nstart),
                       "starts_per_gpu": len(mine), "parallelism": f"starts sharded over {world} GPU(s)",
                       "kernel_variant": st["kernel_variant"],
                       "passes_per_start_mean": sweeps_all / args.steps / nstart,
                       "full_sweep_equivalents_per_start_mean": relaxed_all / cells / args.steps / nstart},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic() if default_workload else None,
                         "kernel": "sweep", "launches": int(launches),
                         "avg_launch_ms": st["sweep_kernel_ms"] / launches,
                         "algorithmic_bytes_per_launch": alg_bytes / launches,
                         "note": "12 B per cell*pass; with 817 relaxations per cell*pass this star is "
                                 "VALU-bound, see roofline_valu"},
            "roofline_valu": {"achieved": lane, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-ops/s",
                              "frac": lane / VALU_PEAK_TLANEOPS,
                              "relaxations_per_pass": st["relaxations_per_sweep"],
                              "laneops_per_relaxation": LANEOPS_PER_RELAX},
        }
        if world == 1 and not args.no_cpu and v_host is not None:
            out["cpu_baseline"] = cpu_baseline(P, v_host, offs, starts[0])
            ncores = min(16, os.cpu_count() or 1)
            if ncores > 1:
                multi = cpu_baseline_cores((nx, ny, nz), args.star, args.starts, ncores)
                if multi is not None:
                    out["cpu_baseline_all_cores"] = multi
        print(json.dumps(out), flush=True)
    sol.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
