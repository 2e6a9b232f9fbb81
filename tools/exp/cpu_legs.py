#!/usr/bin/env python3
"""BASELINE.md section 3, legs B2 and B4, measured once on the GPU box's host cores with the
CPU restatement of serial_new (oracle/, test infrastructure; this script is a measurement
helper, not part of the product):

  B2  start-1 of the 241x241x51 workload relaxed to convergence in the reference's order
      (the loop of old/sweep-serial/sweep-tt-multistart.c:189-211 around the sweep body
      serial_new/sweep-tt-multistart.c:198-256), 818-FS: wall seconds, sweeps, seconds per sweep;
  B4  ONE reference-order sweep of one start on the synthetic 512x512x256 grid, 818-FS and six-FS
      (a CPU run to convergence is infeasible there).

Both legs run side by side (one core each).  Prints one JSON object (profiles/r03_cpu_legs.json).
"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def host_model(P, nx, ny, nz, seed=20160507):
    """The large-grid velocity model (counter-hash noise) evaluated on the CPU: the same
    function bench.py evaluates on the device."""
    import torch
    return P.inputs.velocity_model_device(nx, ny, nz, seed, torch.device("cpu")).numpy()


def leg_b2(q):
    import oracle as O
    import ttsweep_pkg
    P = ttsweep_pkg.load()
    v = P.inputs.velocity_model(241, 241, 51, 20160507)
    fs = O.make_star(P.inputs.read_triples(P.inputs.star_path("818")))
    start = P.inputs.read_triples(P.inputs.starts_path("1"))[0]
    t0 = time.perf_counter()
    tt, sweeps, stores = O.converge(v, fs, start, order=0)
    dt = time.perf_counter() - t0
    import hashlib
    q.put(("B2", {"grid": [241, 241, 51], "star": "818-FS", "start": [int(x) for x in start],
                  "sweeps_incl_confirming": int(sweeps), "stores": int(stores), "seconds": round(dt, 2),
                  "seconds_per_sweep": dt / max(sweeps, 1), "cores": 1,
                  "sha256": hashlib.sha256(tt.tobytes()).hexdigest(),
                  "what": "reference-order passes until a pass stores nothing (serial_new sweep body, "
                          "old/sweep-serial driver loop), single thread, gcc -O3"}))


def leg_b4(q, star):
    import oracle as O
    import ttsweep_pkg
    P = ttsweep_pkg.load()
    nx, ny, nz = 512, 512, 256
    v = host_model(P, nx, ny, nz)
    offs = P.inputs.read_triples(P.inputs.star_path(star))
    fs = O.make_star(offs)
    start = P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path("111"))[:1], nx, ny, nz)[0]
    tt = O.tt_init(v.shape, start)
    t0 = time.perf_counter()
    stores = O.sweep(v, tt, fs, start)
    dt = time.perf_counter() - t0
    q.put((f"B4_{star}", {"grid": [nx, ny, nz], "star": f"{star}-FS ({len(offs)} entries)",
                          "start": [int(x) for x in start], "sweeps": 1, "stores": int(stores),
                          "seconds": round(dt, 2), "mcells_sweeps_per_s": nx * ny * nz / dt / 1e6, "cores": 1,
                          "what": "ONE reference-order pass of one start from the initial state, single thread"}))


def main():
    import oracle as O
    O.build()
    six = os.path.join(ROOT, "data", "stars", "six-FS.txt")
    legs = [(leg_b2, ()), (leg_b4, ("818",))]
    if os.path.exists(six):
        legs.append((leg_b4, ("six",)))
    q = mp.Queue()
    procs = [mp.Process(target=f, args=(q,) + a) for f, a in legs]
    t0 = time.perf_counter()
    for p in procs:
        p.start()
    out = {}
    for _ in procs:
        k, v = q.get()
        out[k] = v
        print(f"[{time.perf_counter() - t0:7.1f} s] {k} done", file=sys.stderr, flush=True)
    for p in procs:
        p.join()
    out["host"] = {"logical_cores": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
