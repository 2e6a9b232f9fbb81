#!/usr/bin/env python3
"""Benchmark of the multi-start travel-time sweep (the hot path of
serial_new/sweep-tt-multistart.c) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Default workload (BASELINE.json metric config): synthetic 241x241x51 velocity volume (the
real model is absent from the reference tree), the reference's default 818-offset star
(docs/818-FS.txt, serial_new/Makefile:15) and the 24 start points of
docs/start-24-241-241-51.txt.  One "step" = one complete multi-start solve: initialise
every travel-time box on the device (all INFINITY, start 0), relax to convergence, and -
for N > 1 - gather all boxes on rank 0 over RCCL.  The starts are sharded over the N ranks
(strong scaling: the 24 starts are the fixed total work); inputs are resident in HBM when
the timed region starts.  Other workloads (parity-test configurations of BASELINE.json, the
HBM-regime run `--star six --grid 1024,1024,512 --starts 111 --nstarts 14`) are selected
with --grid / --star / --starts / --nstarts; every label in the line is derived from them.

value: Mcells*sweeps/s = (cells relaxed against the whole forward star, summed over passes
and starts) / wall seconds / 1e6.  The GPU schedules relax only what can change, so this is
NOT comparable across schedules; `time_to_solution` (seconds to the converged boxes, GPU
measured, CPU extrapolated from measured seconds per reference sweep x the recorded
reference sweep counts) is.

roofline: the roof that binds the dominant kernel.  Per cell*pass the path moves 12
compulsory bytes (read v, read T, write T) and does 4 flops (add, mul, add, min) per star
offset: above ~10 flops per byte (every shipped star) the vector ALUs bind ("valu": fp32
vector flops against 256 CU x 4 SIMD x 32 lanes x 2.4 GHz), below it HBM does ("hbm").
`roofline_hbm` always carries the HBM view the BASELINE metric names.  `traffic` = bytes
that crossed the L2's memory side per launch, measured IN THIS RUN by two rocprofv3 --pmc
child passes (FETCH_SIZE, WRITE_SIZE; gfx950 read correction x2) of the same workload.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TFLOPS = 78.6             # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz, one flop per lane-op (no FMA:
                                    # the reference rounds the multiply and the add separately)
BYTES_PER_CELL_SWEEP = 12           # read v, read tt, write tt
FLOPS_PER_RELAX = 4                 # add, mul, add, min
BALANCE_FLOPS_PER_BYTE = VALU_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup's CPU
    quota where one is set (a container can see 256 cores and own 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(P, v, offs, start, sweeps=4):
    """The CPU restatement of serial_new (oracle/, kind "port") timed on this host:
    `sweeps` reference-order passes of one start from the initial state."""
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
            "kind": "port", "seconds": round(dt, 2), "seconds_per_sweep": dt / sweeps,
            "sample": f"{sweeps} reference-order passes of start 0 on the same grid/star "
                      f"(single thread, gcc -O3; {usable_cores()} usable of {os.cpu_count()} logical cores)"}


def cpu_baseline_cores(shape, star, starts_name, ncores, sweeps=1):
    """The same CPU restatement with one start per core (the strategy of the reference's
    mpi/backup.c:351-363), `ncores` processes side by side, each `sweeps` reference-order
    passes of its own start.  Child processes: they never touch the GPU."""
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
    return {"value": sum(cells * sweeps / t for t in secs) / 1e6, "unit": "Mcells*sweeps/s",
            "cores": ncores, "kind": "port", "seconds": round(max(secs), 2),
            "seconds_per_sweep_per_core": sum(secs) / len(secs) / sweeps,
            "wall_seconds_incl_startup": round(wall, 2),
            "sample": f"{sweeps} reference-order pass(es) of {ncores} different starts, one process per "
                      f"usable core, side by side ({os.cpu_count()} logical cores on the host, "
                      f"{len(os.sched_getaffinity(0))} in the affinity mask, {usable_cores()} within the CPU quota)"}


def cpu_leg_b2(P):
    """BASELINE.md section 3, leg B2: start-1 of the 241x241x51 workload relaxed to convergence in the reference's
    order (sweep body serial_new/sweep-tt-multistart.c:198-256, driver loop old/sweep-serial/...:189-211) with the CPU
    restatement on ONE host core; the box's SHA-256 against the one recorded from the unmodified reference."""
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    v = P.inputs.velocity_model(241, 241, 51, 20160507)
    fs = O.make_star(P.inputs.read_triples(P.inputs.star_path("818")))
    start = P.inputs.read_triples(P.inputs.starts_path("1"))[0]
    t0 = time.perf_counter()
    tt, sweeps, stores = O.converge(v, fs, start, order=0)
    dt = time.perf_counter() - t0
    sha = hashlib.sha256(tt.tobytes()).hexdigest()
    want = None
    try:
        d = json.load(open(os.path.join(ROOT, "tests", "golden", "big_digests.json")))
        want = d["syn241_818_%d_%d_%d" % tuple(int(x) for x in start)]["sha256"]
    except Exception:
        pass
    return {"leg": "B2", "grid": [241, 241, 51], "star": "818-FS", "start": [int(x) for x in start], "cores": 1, "kind": "port",
            "sweeps_incl_confirming": int(sweeps), "stores": int(stores), "seconds": round(dt, 2),
            "seconds_per_sweep": dt / max(int(sweeps), 1), "value": v.size * int(sweeps) / dt / 1e6, "unit": "Mcells*sweeps/s",
            "sha256": sha, "matches_reference_digest": (sha == want) if want else None,
            "sample": "start-1 to convergence, reference order, single thread, gcc -O3 (measured in this run)"}


def gather_report(gather, ranks_info, shards, cells):
    """config.gather of the line: where the boxes went and why (multistart.plan_gather) and, for N > 1, what the
    timed steps saw - bytes that crossed to the root per step and the rate over the slowest rank's gather time (the
    root waits for all senders).  Everything here is MEASURED; nothing is quoted from another run."""
    out = dict(gather)
    if ranks_info:
        moved = sum(len(sh) for r, sh in enumerate(shards) if r != 0) * cells * 4
        slowest = max(r["gather_ms"] for r in ranks_info)
        out["bytes_to_root_per_step"] = int(moved)
        out["slowest_rank_gather_ms"] = slowest
        out["achieved_GBps"] = moved / (slowest / 1e3) / 1e9 if slowest > 0 else None
        out["note"] = ("gather_ms per rank = wall time from the end of its solve to the completion of its part of the "
                       "gather, so a rank that finishes its shard early also waits there for the slowest solver")
    return out


def reference_sweep_counts(star, starts):
    """Sweeps the unmodified reference needed per start (recorded with the reference itself,
    tests/golden/big_digests.json), for the starts that were recorded."""
    try:
        d = json.load(open(os.path.join(ROOT, "tests", "golden", "big_digests.json")))
    except Exception:
        return []
    out = []
    for i, j, k in starts:
        rec = d.get(f"syn241_{star}_{i}_{j}_{k}")
        if rec:
            out.append(int(rec["sweeps"]))
    return out


def workload_args(args):
    return ["--grid", args.grid, "--star", args.star, "--starts", args.starts,
            "--nstarts", str(args.nstarts), "--kernel", str(args.kernel)] + \
        (["--gate-speed", str(args.gate_speed)] if args.gate_speed is not None else []) + \
        (["--pair-min-starts", str(args.pair_min_starts)] if args.pair_min_starts is not None else []) + \
        (["--async-mode", str(args.async_mode)] if args.async_mode is not None else []) + \
        (["--async-gate", str(args.async_gate)] if args.async_gate is not None else []) + \
        (["--defer-margin", str(args.defer_margin)] if args.defer_margin is not None else []) + \
        (["--inunit", str(args.inunit)] if args.inunit is not None else []) + \
        (["--prepass", str(args.prepass)] if args.prepass else []) + \
        (["--lib", args.lib] if args.lib else [])


def measure_traffic(args, kernel_patterns):
    """HBM-side bytes per sweep-kernel launch of THIS workload: two child runs of this script
    under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE need separate passes), started before
    this process touches the GPU.  Returns None when the profiler is not usable."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    sums, launches = {}, 0
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ttsweep_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--child", "--steps", "1", "--warmup", "1",
                   "--no-cpu"] + workload_args(args)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=150, env=env, cwd="/tmp")
            if r.returncode != 0:
                return None
            total, n = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    name = row["Kernel_Name"]
                    counted = any(p in name for p in kernel_patterns[0])
                    if counted or any(p in name for p in kernel_patterns[1]):
                        total += float(row["Counter_Value"])
                        n += counted
            if n == 0:
                return None
            sums[counter], launches = total, n
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch_b = sums["FETCH_SIZE"] * 1024.0 * 2.0      # gfx950: FETCH_SIZE reads half of wide coalesced loads
    write_b = sums["WRITE_SIZE"] * 1024.0
    return {"bytes_per_launch": (fetch_b + write_b) / launches, "fetch_bytes_per_launch": fetch_b / launches,
            "write_bytes_per_launch": write_b / launches, "launches_profiled": launches,
            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run "
                      "(kernels: " + ", ".join(kernel_patterns[0] + kernel_patterns[1]) + "; read counter x2, gfx950)"}


def host_program_end_to_end(P, shape, star, starts, v_host):
    """The actual drop-in product: the plain-C host program (the reference's main(), vbox /
    star / start files in, libttsweep.so behind sweepXYZ) on the same workload, output.tt
    suppressed.  Wall seconds of the whole process and of its sweep loop (host buffers:
    transfers included)."""
    exe = os.path.join(ROOT, "uoparallel-seismic-project_amd", "host", "sweep-tt-multistart")
    if not os.path.exists(exe) or v_host is None:
        return None
    d = tempfile.mkdtemp(prefix="ttsweep_host_", dir="/tmp")
    try:
        vfile = os.path.join(d, "model.vbox")
        P.inputs.write_vbox(vfile, v_host, (1, 1, 1))
        sfile = os.path.join(d, "starts.txt")
        with open(sfile, "w") as f:
            f.write(f"{len(starts)}\n" + "".join(f"{i} {j} {k}\n" for i, j, k in starts))
        env = dict(os.environ, TTSWEEP_NO_OUTPUT="1")
        runs = []
        for _ in range(2):      # (the first run of a fresh box pages the system HIP runtime in from the image)
            time.sleep(1.5)     # (the driver tears the previous GPU process down asynchronously: a process that
                                #  starts right behind another one's exit waits for that inside its own HIP
                                #  initialisation - 0.1-0.25 s that belong to the other process)
            t0 = time.perf_counter()
            r = subprocess.run([exe, vfile, P.inputs.star_path(star), sfile], capture_output=True, text=True,
                               timeout=600, env=env, cwd=d)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return None
            loop = [ln for ln in r.stdout.splitlines() if ln.startswith("ttsweep: sweep loop")]
            dev = [ln for ln in r.stdout.splitlines() if "ms on device" in ln]
            runs.append({"process_wall_seconds": round(wall, 3),
                         "sweep_loop_wall_seconds": float(loop[-1].split()[3]) if loop else None,
                         "device_ms": float(dev[0].split()[-4]) if dev else None})
        out = dict(runs[1])
        out["first_run_on_this_box"] = runs[0]
        out["what"] = ("host/sweep-tt-multistart <vbox> <star> <starts> with TTSWEEP_NO_OUTPUT=1: process start, VBOX load, "
                       "context creation (its HIP runtime initialises on a thread of its own from process start), pinned "
                       "host<->device transfers of every box, solve, confirming second driver pass (answered from box "
                       "digests).  Run twice, each 1.5 s after the previous GPU process has exited (its teardown would "
                       "otherwise be waited for inside this one's HIP initialisation): the figures are the second "
                       "run's; the first run of a fresh box also pages the system libamdhip64 / HSA runtime in from "
                       "the container image (first_run_on_this_box)")
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


HBM_REGIME = {"grid": "1024,1024,512", "star": "six", "starts": "111", "nstarts": 14}


def hbm_regime_wanted(args):
    """The default line (the BASELINE metric workload, N = 1) also carries the north star's
    "HBM-roofline run": BASELINE.json config 5's grid with the 6-neighbour star, one GPU's share
    (14 of the 111 starts), measured in the same process after the headline."""
    return (not args.child and not args.no_hbm_regime and args.gpus == 1 and args.grid == "241,241,51"
            and args.star == "818" and args.starts == "24" and args.nstarts == 0 and args.kernel == 0)


def hbm_regime_args(args):
    a = argparse.Namespace(**vars(args))
    a.grid, a.star, a.starts, a.nstarts = (HBM_REGIME[k] for k in ("grid", "star", "starts", "nstarts"))
    a.gate_speed = a.pair_min_starts = a.async_mode = a.async_gate = a.defer_margin = a.inunit = None
    a.prepass = 0
    return a


def hbm_regime(P, torch, dev, dev_index, traffic, with_cpu, steps=2):
    """six-FS on 1024x1024x512, 14 starts, TILE kernel (ordered 8-ordering Gauss-Seidel tile
    sweeps): the regime where HBM, not the vector ALUs, binds (24 flops per 12 B).  achieved =
    12 B x cells of the tiles relaxed / time of the sweep launches (HIP events on the library's
    stream around them).  One launch per solve (column pipelines, ttsweep_column.hip) or, under
    the hyperplane driver, one launch per tile hyperplane of an ordering sweep."""
    nx, ny, nz = map(int, HBM_REGIME["grid"].split(","))
    cells = nx * ny * nz
    offs = P.inputs.read_triples(P.inputs.star_path(HBM_REGIME["star"]))
    fs = P.inputs.make_fs(offs)
    starts = P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path(HBM_REGIME["starts"])), nx, ny, nz)
    starts = starts[:HBM_REGIME["nstarts"]]
    t_gen = time.perf_counter()
    v_dev = P.inputs.velocity_model_device(nx, ny, nz, 20160507, dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    sol = P.TravelTimeSolver((nx, ny, nz), fs, device=dev_index)
    sol.set_velocity(v_dev)
    tt = torch.empty((len(starts), nx, ny, nz), dtype=torch.float32, device=dev)
    sol.solve_device(starts, tt, init=True)                 # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sol.solve_device(starts, tt, init=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sol.set_option(P.OPT_TIMING, 1)
    sol.solve_device(starts, tt, init=True)
    st = sol.stats()
    sol.set_option(P.OPT_TIMING, 0)
    kern_s = st["sweep_kernel_ms"] / 1e3
    launches = max(st["launches"], 1)
    alg = BYTES_PER_CELL_SWEEP * st["cells_relaxed"]
    gbs = alg / kern_s / 1e9 if kern_s > 0 else 0.0
    out = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "traffic": traffic["bytes_per_launch"] if traffic else None,
           "traffic_source": traffic["source"] if traffic else "not measured in this run",
           "kernel": ("column_solve_kernel (ONE launch per solve: the ordering sweeps, their order and the test for rest run "
                      "on the device)" if launches == 1 else
                      "tile_six_kernel (one tile hyperplane of an ordering sweep: plans and relaxes its due tiles)"),
           "launches": int(launches), "avg_launch_ms": st["sweep_kernel_ms"] / launches,
           "frac_over_solve": alg / dt / 1e9 / HBM_PEAK_GBS if dt > 0 else 0.0,
           "fallbacks": int(st.get("fallbacks", 0)),
           "algorithmic_bytes_per_launch": alg / launches,
           "workload": f"{nx}x{ny}x{nz} synthetic velocity, six-FS (6-neighbour star), {len(starts)} starts "
                       f"(one GPU's share of start-111), converged multi-start solve",
           "kernel_variant": st["kernel_variant"], "ms_per_solve": dt * 1e3, "steps": steps,
           "ordering_sweeps_per_start_mean": st["sweeps_total"] / len(starts),
           "grid_equivalents_of_tiles_per_start_mean": st["cells_relaxed"] / cells / len(starts),
           "model_generation_s": round(t_gen, 2),
           "note": "frac counts the bytes of the tiles this schedule actually relaxed: round 5 orders each start's sweeps from "
                   "its nearest corner (TTSWEEP_OPT_TILE_ORDER 111) and converges with a third fewer tile relaxations than "
                   "rounds 3-4 (order 0) at the same speed per tile - less time, LOWER frac; compare ms_per_solve, and "
                   "same_run_with_the_order_of_rounds_3_4 below (same box, same process)"}
    # the same solve with the sequence of orderings rounds 3 - 4 used (every start swept from the corner (0, 0, 0), reflected
    # Gray code): what the fraction and the time would be without the round-5 schedule, measured side by side
    sol.set_option(P.OPT_TILE_ORDER, 0)
    sol.solve_device(starts, tt, init=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sol.solve_device(starts, tt, init=True)
    torch.cuda.synchronize()
    dt0 = time.perf_counter() - t0
    sol.set_option(P.OPT_TIMING, 1)
    sol.solve_device(starts, tt, init=True)
    st0 = sol.stats()
    sol.set_option(P.OPT_TIMING, 0)
    alg0 = BYTES_PER_CELL_SWEEP * st0["cells_relaxed"]
    out["same_run_with_the_order_of_rounds_3_4"] = {
        "ms_per_solve": dt0 * 1e3, "avg_launch_ms": st0["sweep_kernel_ms"] / max(st0["launches"], 1),
        "frac": alg0 / (st0["sweep_kernel_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS if st0["sweep_kernel_ms"] > 0 else 0.0,
        "grid_equivalents_of_tiles_per_start_mean": st0["cells_relaxed"] / cells / len(starts),
        "ordering_sweeps_per_start_mean": st0["sweeps_total"] / len(starts), "fallbacks": int(st0.get("fallbacks", 0))}
    if with_cpu:
        # the CPU restatement beside it: ONE reference-order pass of one start on the same grid
        # (a run to convergence is infeasible: the reference order needs of the order of a
        # thousand passes here); the model is the one the GPU generated
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        O.build()
        v_host = v_dev.cpu().numpy()
        ofs = O.make_star(offs)
        t_cpu = O.tt_init(v_host.shape, starts[0])
        t0 = time.perf_counter()
        O.sweep(v_host, t_cpu, ofs, starts[0])
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": cells / dtc / 1e6, "unit": "Mcells*sweeps/s", "cores": 1, "kind": "port",
                               "seconds": round(dtc, 2),
                               "sample": "ONE reference-order pass of start 0 on the same 1024x1024x512 grid and "
                                         "six-FS star (single thread, gcc -O3)"}
    sol.close()
    return out


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
    ap.add_argument("--gate-speed", type=float, default=None, help="schedule knob of the STRIP kernel (cells/pass)")
    ap.add_argument("--pair-min-starts", type=int, default=None,
                    help="schedule knob of the STRIP kernel: units of two planes from this many starts on")
    ap.add_argument("--async-mode", type=int, default=None, choices=[-1, 0, 1],
                    help="schedule knob of the STRIP kernel: 1 one launch per solve, 0 a launch pair per pass, -1 library default")
    ap.add_argument("--async-gate", type=float, default=None, help="schedule knob: cells per round by which the gate of a one-launch solve opens")
    ap.add_argument("--inunit", type=int, default=None, help="schedule knob: passes of a unit that improved against its own planes (one-launch STRIP solve)")
    ap.add_argument("--handoff", type=int, default=None, help="schedule knob: TTSWEEP_OPT_ASYNC_HANDOFF (workers publish successor units themselves)")
    ap.add_argument("--waves", type=int, default=None, choices=[-1, 4, 8], help="schedule knob: TTSWEEP_OPT_ASYNC_WAVES (waves that relax a unit)")
    ap.add_argument("--special", type=int, default=None, help="schedule knob: TTSWEEP_OPT_ASYNC_SPECIAL")
    ap.add_argument("--cpu-b2", action="store_true",
                    help="BASELINE.md leg B2 in this run: start-1 relaxed to convergence in the reference's order on ONE host "
                         "core with the CPU restatement (about 190 s; not part of the driver's default run)")
    ap.add_argument("--defer-margin", type=float, default=None,
                    help="schedule knob: improvements are told at once only to units not nearer to the start by more than this (cells)")
    ap.add_argument("--prepass", type=int, default=0,
                    help="schedule knob: relax the first N star entries to convergence first (TTSWEEP_OPT_PREPASS_ENTRIES)")
    ap.add_argument("--lib", default=None, help="another build of libttsweep.so (A/B builds, tools/exp)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child passes")
    ap.add_argument("--no-host", action="store_true", help="skip the host-program end-to-end leg")
    ap.add_argument("--no-hbm-regime", action="store_true",
                    help="skip the HBM-regime run (six-FS, 1024x1024x512, 14 starts) the default line carries")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)     # profiled child of measure_traffic
    ap.add_argument("--gather", default="auto", choices=["auto", "device", "host"],
                    help="N > 1: where the boxes are gathered (auto: by size, multistart.plan_gather)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                         "the multi-rank path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()
    if args.child:
        args.no_cpu = args.no_traffic = args.no_host = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import ttsweep_pkg
    P = ttsweep_pkg.load()
    if args.lib:
        P._lib.use_library(args.lib)
    nx, ny, nz = map(int, args.grid.split(","))
    cells = nx * ny * nz
    offs = P.inputs.read_triples(P.inputs.star_path(args.star))
    fs = P.inputs.make_fs(offs)
    npull = len(P.build_pull_star(fs))
    starts = P.inputs.read_triples(P.inputs.starts_path(args.starts))
    if args.nstarts:
        starts = starts[:args.nstarts]
    small = (nx, ny, nz) == (241, 241, 51)
    if not small:
        starts = P.inputs.scaled_starts(starts, nx, ny, nz)
    nstart = len(starts)
    valu_bound = FLOPS_PER_RELAX * npull / BYTES_PER_CELL_SWEEP > BALANCE_FLOPS_PER_BYTE
    tile_star = npull <= 26 and args.kernel in (0, 3)      # (the library's own choice is read back below)
    # kernels of one sweep launch: the first group is counted (one per launch), all are summed
    patterns = ([["column_solve_kernel", "tile_six_kernel", "tile_sweep_kernel"], ["tile_plan_kernel"]] if tile_star
                else [["sweep_units_kernel"], ["plan_pass_kernel"]])

    # ---- legs that run other processes on the GPU: before this one initialises it
    traffic = host_e2e = None
    v_host = P.inputs.velocity_model(nx, ny, nz, 20160507) if small else None
    hbm_traffic = None
    if world == 1 and not args.no_traffic:
        traffic = measure_traffic(args, patterns)
        if hbm_regime_wanted(args):
            hbm_traffic = measure_traffic(hbm_regime_args(args), [["column_solve_kernel", "tile_six_kernel", "tile_sweep_kernel"], []])
    if world == 1 and not args.no_host and small:
        host_e2e = host_program_end_to_end(P, (nx, ny, nz), args.star, starts, v_host)

    import torch
    dist = None
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    # Process groups.  The DEFAULT group is gloo - barriers, plans, the max-over-ranks clock, the diagnostics - and
    # comes up first.  RCCL (backend "nccl" IS RCCL on ROCm) is a group of its own for the gather alone, and the ranks
    # AGREE on whether it works before anybody depends on it: every rank tries the group and one tiny all-reduce on
    # its device, the outcomes are min-reduced over gloo, and unless every rank succeeded all of them go on under gloo
    # with the host gather (and the line says why).  A rank that fails alone therefore cannot leave the others inside
    # an RCCL rendezvous (round 4 decided per rank).
    rccl = None
    rccl_note = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        dist.init_process_group("gloo")
        if args.backend == "nccl":
            err = None
            try:
                rccl = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180))
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe, group=rccl)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    err = f"RCCL probe all-reduce returned {probe.item()} for {world} ranks"
            except Exception as e:      # noqa: BLE001 - whatever RCCL / the rendezvous raises
                err = f"rank {rank}: RCCL group / probe failed: {type(e).__name__}: {e}"[:300]
            errs = [None] * world
            dist.all_gather_object(errs, err)
            bad = [x for x in errs if x]
            if bad:
                rccl_note = bad[0] if len(bad) == world else f"{len(bad)} of {world} ranks: {bad[0]}"[:300]
                rccl = None
                args.backend = "gloo"

    if small:
        v_dev = torch.from_numpy(v_host).to(dev)
    else:
        v_dev = P.inputs.velocity_model_device(nx, ny, nz, 20160507, dev)
    # starts differ in cost (22 to 76 reference sweeps): balance the shards by estimated cost
    shards = P.multistart.all_shards(nstart, world, starts=starts, shape=(nx, ny, nz))
    mine = shards[rank]
    my_starts = starts[mine]

    sol = P.TravelTimeSolver((nx, ny, nz), fs, device=dev_index)
    if args.kernel:
        sol.set_option(P.OPT_KERNEL, args.kernel)
    if args.gate_speed is not None:
        sol.set_option(P.OPT_GATE_SPEED_MILLI, int(round(args.gate_speed * 1000)))
    if args.pair_min_starts is not None:
        sol.set_option(P.OPT_PAIR_MIN_STARTS, args.pair_min_starts)
    if args.async_mode is not None:
        sol.set_option(P.OPT_ASYNC, args.async_mode)
    if args.async_gate is not None:
        sol.set_option(P.OPT_ASYNC_GATE_MILLI, int(round(args.async_gate * 1000)))
    if args.defer_margin is not None:
        sol.set_option(P.OPT_DEFER_MARGIN_MILLI, int(round(args.defer_margin * 1000)))
    if args.inunit is not None:
        sol.set_option(P.OPT_ASYNC_INUNIT, args.inunit)
    if args.handoff is not None:
        sol.set_option(P.OPT_ASYNC_HANDOFF, args.handoff)
    if args.waves is not None:
        sol.set_option(P.OPT_ASYNC_WAVES, args.waves)
    if args.special is not None:
        sol.set_option(P.OPT_ASYNC_SPECIAL, args.special)
    sol.set_velocity(v_dev)
    if args.prepass:
        sol.set_option(P.OPT_PREPASS_ENTRIES, args.prepass)
    tt = torch.empty((len(mine), nx, ny, nz), dtype=torch.float32, device=dev)

    # where the final gather goes: rank 0's device memory while the result set fits a stated
    # share of what is free there, host memory (one shared array every rank's GPU copies into)
    # beyond that - config 5's 111 x 2.1 GB do not fit one GPU - and always under the gloo
    # rehearsal backend, which cannot move device tensors
    gather = {"path": "none", "why": "one rank"}
    if dist is not None:
        plan = [None]
        if rank == 0:
            free_dev = torch.cuda.mem_get_info(dev)[0] if args.backend == "nccl" else None
            plan[0] = P.multistart.plan_gather(nstart, cells * 4, free_dev)
            if args.gather != "auto":
                plan[0] = {"path": args.gather, "bytes": nstart * cells * 4, "why": "--gather"}
        dist.broadcast_object_list(plan, src=0)
        gather = plan[0]
        if rccl_note:
            gather = dict(gather, path="host", why=rccl_note)

    # per-rank diagnostics of the timed steps (N > 1): where a step's time goes on every rank
    phase = {"solve_s": 0.0, "gather_s": 0.0, "solve_device_ms": 0.0, "fallbacks": 0, "steps": 0}

    def step(timed=False):
        t_a = time.perf_counter()
        sol.solve_device(my_starts, tt, init=True)      # (returns when the boxes are converged: the library synchronises)
        t_b = time.perf_counter()
        out_ = P.multistart.gather_boxes(tt, nstart, dist, dst=0, shards=shards, path=gather["path"], group=rccl)
        if timed:
            if dist is not None:
                torch.cuda.synchronize()                # (this rank's part of the gather has completed)
            st_ = sol.stats()
            phase["solve_s"] += t_b - t_a
            phase["gather_s"] += time.perf_counter() - t_b
            phase["solve_device_ms"] += st_["solve_ms"]
            phase["fallbacks"] += int(st_.get("fallbacks", 0))
            phase["steps"] += 1
        return out_

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if dist is not None and gather["path"] == "device":
        # the device gather's first run (grouped RCCL send / receive pairs): if it fails on ANY rank, every rank
        # switches to the host path (agreed over the gloo group) and the line carries the reason
        err = None
        try:
            step()
            torch.cuda.synchronize()
        except Exception as e:          # noqa: BLE001
            err = f"device gather failed on rank {rank}: {type(e).__name__}: {e}"[:300]
        errs = [None] * world
        dist.all_gather_object(errs, err)
        bad = [x for x in errs if x]
        if bad:
            gather = dict(gather, path="host", why=bad[0])
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    sweeps_local = 0            # passes launched (a pass relaxes only what is due)
    relaxed_local = 0           # cells actually relaxed against the whole star
    for _ in range(args.steps):
        step(timed=True)
        st_ = sol.stats()
        sweeps_local += st_["sweeps_total"]
        relaxed_local += st_["cells_relaxed"]
    fence()
    dt = time.perf_counter() - t0
    dt_local = dt

    # max over ranks of the elapsed time, sum over ranks of the passes executed
    agg = torch.tensor([dt, float(sweeps_local), float(relaxed_local)], dtype=torch.float64)
    ranks_info = None
    if dist is not None:
        tmax = agg[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ssum = agg[1:].clone()
        dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
        dt, sweeps_all, relaxed_all = float(tmax.item()), float(ssum[0].item()), float(ssum[1].item())
        # every rank's share of a step, for the line: the first multi-GPU run has to say WHY it scales as it does
        k = max(phase["steps"], 1)
        mine_info = {"rank": rank, "device": dev_index, "starts": len(mine), "start_ids": [int(x) for x in mine],
                     "shard_cost_estimate": round(sum(P.multistart.start_cost(starts[i], (nx, ny, nz)) for i in mine), 1),
                     "solve_ms": phase["solve_s"] / k * 1e3, "solve_device_ms": phase["solve_device_ms"] / k,
                     "gather_ms": phase["gather_s"] / k * 1e3, "step_ms": dt_local / max(args.steps, 1) * 1e3,
                     "full_sweep_equivalents_per_start": relaxed_local / cells / max(args.steps, 1) / max(len(mine), 1),
                     "fallbacks": phase["fallbacks"]}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine_info)
    else:
        sweeps_all, relaxed_all = float(sweeps_local), float(relaxed_local)

    # one extra, instrumented solve: HIP events around every sweep launch on the
    # library's own stream (not part of the timed region)
    sol.set_option(P.OPT_TIMING, 1)
    sol.solve_device(my_starts, tt, init=True)
    st = sol.stats()
    sol.set_option(P.OPT_TIMING, 0)

    failed = False
    if rank == 0:
        kern_s = st["sweep_kernel_ms"] / 1e3
        launches = max(st["launches"], 1)
        alg_bytes = BYTES_PER_CELL_SWEEP * st["cells_relaxed"]
        relax = st["relaxations_per_sweep"] * (st["cells_relaxed"] / cells)
        flops = FLOPS_PER_RELAX * relax
        gbs = alg_bytes / kern_s / 1e9 if kern_s > 0 else 0.0
        tfl = flops / kern_s / 1e12 if kern_s > 0 else 0.0
        kname = {1: "sweep_cell_kernel", 2: "plan_pass_kernel + sweep_units_kernel (one pass)" if st["launches"] > 1 else
                 "sweep_units_kernel<16, np, true> (ONE launch per solve: ring planners + workers, convergence detected on the device)",
                 3: ("column_solve_kernel (one launch per solve: column pipelines)" if st["launches"] == 1 else
                     "tile_six_kernel / tile_sweep_kernel (one tile hyperplane of an ordering sweep)")}[st["kernel_variant"]]
        common = {"kernel": kname, "launches": int(launches), "avg_launch_ms": st["sweep_kernel_ms"] / launches,
                  "algorithmic_bytes_per_launch": alg_bytes / launches,
                  "algorithmic_flops_per_launch": flops / launches,
                  "traffic": traffic["bytes_per_launch"] if traffic else None,
                  "traffic_source": traffic["source"] if traffic else
                  "not measured in this run (--no-traffic, N > 1, or rocprofv3 unavailable)"}
        hbm = dict({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS}, **common)
        valu = dict({"bound": "valu", "achieved": tfl, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tfl / VALU_PEAK_TFLOPS}, **common)
        per_cell = f"{FLOPS_PER_RELAX * npull} flops per {BYTES_PER_CELL_SWEEP} B per cell*pass " \
                   f"({npull} offsets; machine balance {BALANCE_FLOPS_PER_BYTE:.1f} flops/B)"
        hbm["note"] = valu["note"] = per_cell + (": the vector ALUs bind" if valu_bound else ": HBM binds")
        gpu_s = dt / args.steps
        out = {
            "metric": f"Mcells*sweeps/s ({nx}x{ny}x{nz}, {len(offs)}-offset star, {nstart} starts, to convergence)",
            "value": relaxed_all / dt / 1e6,
            "unit": "Mcells*sweeps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": gpu_s * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{nx}x{ny}x{nz} synthetic velocity, {args.star}-FS star, "
                                   f"{nstart} starts (start-{args.starts}), converged multi-start solve",
                       "grid": [nx, ny, nz], "star_offsets": int(len(offs)), "starts": int(nstart),
                       "starts_per_gpu": len(mine), "parallelism": f"starts sharded over {world} GPU(s)",
                       "gather": gather_report(gather, ranks_info, shards, cells),
                       "ranks": ranks_info,
                       "kernel_variant": st["kernel_variant"],
                       "driver": ("one launch per solve (ring planners + workers, convergence detected on the device)"
                                  if st["kernel_variant"] in (2, 3) and st["launches"] == 1 else "a launch (pair) per pass / hyperplane, convergence tested on the host"),
                       "fallbacks": int(st.get("fallbacks", 0)) + (sum(r["fallbacks"] for r in ranks_info) if ranks_info else phase["fallbacks"]),
                       "library": os.path.relpath(P._lib.LIB_PATH, ROOT),
                       "passes_per_start_mean": sweeps_all / args.steps / nstart,
                       "full_sweep_equivalents_per_start_mean": relaxed_all / cells / args.steps / nstart},
            "value_note": "cells x passes this solver actually relaxed per second (sweep equivalents, in-unit passes counted): a "
                          "schedule that converges with LESS relaxation scores LOWER here at the same speed per relaxation - "
                          "compare ms_per_step / time_to_solution across rounds and schedules (round 3: 34.3 ms at 4.79 "
                          "equivalents per start; this line's equivalents: config.full_sweep_equivalents_per_start_mean)",
            "roofline": valu if valu_bound else hbm,
            "roofline_hbm": hbm,
            "time_to_solution": {"gpu_all_starts_s": gpu_s, "gpu_per_start_s": gpu_s / nstart,
                                 "what": "seconds from initial boxes to converged boxes, all starts batched"},
        }
        if traffic:
            out["roofline"]["traffic_detail"] = traffic
        if host_e2e:
            out["end_to_end_host_program"] = host_e2e
        if not args.no_cpu and v_host is not None:
            # (N > 1: serial_new on the box's own host cores in the same run, one core, on rank 0)
            cb = cpu_baseline(P, v_host, offs, starts[0])
            out["cpu_baseline"] = cb
            ncores = usable_cores() if world == 1 else 1
            if ncores > 1:
                multi = cpu_baseline_cores((nx, ny, nz), args.star, args.starts, ncores)
                if multi is not None:
                    out["cpu_baseline_all_cores"] = multi
            ref = reference_sweep_counts(args.star, starts)
            if ref:
                mean_sweeps = sum(ref) / len(ref)
                per_start = cb["seconds_per_sweep"] * mean_sweeps
                tts_ = out["time_to_solution"]
                tts_["cpu_per_start_s_extrapolated"] = per_start
                # the same solve in units of the reference's own sweeps: cells x (sweeps the unmodified reference needs for
                # these starts) per second of this solver
                if len(ref) == nstart:
                  out["reference_sweeps_replaced_per_s"] = {
                    "value": cells * sum(ref) / gpu_s / 1e6, "unit": "Mcells*reference sweeps/s",
                    "what": f"{len(ref)} starts x {mean_sweeps:.1f} reference sweeps (mean, tests/golden/big_digests.json) x {cells} cells / gpu_all_starts_s"}
                tts_["cpu_all_starts_one_core_s_extrapolated"] = per_start * nstart
                if "cpu_baseline_all_cores" in out:
                    c = out["cpu_baseline_all_cores"]
                    waves = -(-nstart // c["cores"])
                    tts_["cpu_all_starts_all_cores_s_extrapolated"] = \
                        c["seconds_per_sweep_per_core"] * mean_sweeps * waves
                tts_["cpu_basis"] = (f"measured seconds per reference-order sweep on this host x the sweeps the "
                                     f"unmodified reference needed (mean {mean_sweeps:.1f} over {len(ref)} of the "
                                     f"{nstart} starts, tests/golden/big_digests.json); one start per core")
        if args.cpu_b2 and small:
            out["cpu_baseline_b2"] = cpu_leg_b2(P)
        if hbm_regime_wanted(args) and world == 1:
            sol.close()
            del tt, v_dev
            torch.cuda.empty_cache()
            out["roofline_hbm_regime"] = hbm_regime(P, torch, dev, dev_index, hbm_traffic, not args.no_cpu)
        if out["config"]["fallbacks"]:
            # a one-launch solve that gives up is finished by the second driver with the same result - but not in the
            # time this line is about: the run counts as failed
            out["error"] = f"{out['config']['fallbacks']} solve(s) of the measured steps fell back to the pass / hyperplane driver"
            failed = True
        print(json.dumps(out), flush=True)
    sol.close()
    if dist is not None:
        flag = [failed]
        dist.broadcast_object_list(flag, src=0)
        failed = bool(flag[0])
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
