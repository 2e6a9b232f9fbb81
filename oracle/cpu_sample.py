#!/usr/bin/env python3
"""Test infrastructure (bench.py's cpu_baseline legs only): time `sweeps` reference-order
passes of ONE start of the benchmark workload with the CPU restatement, in a process of its
own.  Prints the seconds the passes took.  Several of these run side by side give the
"one start per core" figure (the strategy of the reference's mpi/backup.c:351-363).

  python oracle/cpu_sample.py NX NY NZ STAR START_FILE START_INDEX SWEEPS
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    nx, ny, nz = (int(a) for a in sys.argv[1:4])
    star, starts_name, index, sweeps = sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
    import oracle as O
    import ttsweep_pkg
    P = ttsweep_pkg.load()
    v = P.inputs.velocity_model(nx, ny, nz, 20160507)
    fs = O.make_star(P.inputs.read_triples(P.inputs.star_path(star)))
    starts = P.inputs.read_triples(P.inputs.starts_path(starts_name))
    start = starts[index % len(starts)]
    tt = O.tt_init(v.shape, start)
    t0 = time.perf_counter()
    for _ in range(sweeps):
        O.sweep(v, tt, fs, start)
    print(time.perf_counter() - t0, flush=True)


if __name__ == "__main__":
    main()
