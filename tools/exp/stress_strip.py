"""Randomised cross-check of the STRIP kernel (one- and two-plane units, default rule) against the CELL
kernel (both on the GPU, bit for bit): random grid shapes, the shipped big stars, 1-5 starts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, ttsweep_pkg
P = ttsweep_pkg.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 20
def solve(kernel, pair_min, v, fs, starts, tts):
    with P.TravelTimeSolver(v.shape, fs) as sol:
        sol.set_option(P.OPT_KERNEL, kernel)
        if pair_min is not None:
            sol.set_option(P.OPT_PAIR_MIN_STARTS, pair_min)
        sol.set_velocity(v)
        return sol.solve(starts, tts), sol.stats()
bad = 0
for case in range(ncase):
    shape = tuple(int(x) for x in rng.integers(1, [120, 120, 90]))
    v = rng.uniform(0.05, 1.0, size=shape).astype(np.float32)
    star = ("818", "5", "3")[case % 3]
    offs = P.inputs.read_triples(P.inputs.star_path(star))
    fs = P.inputs.make_fs(offs)
    ns = int(rng.integers(1, 6))
    starts = np.stack([rng.integers(0, n, size=ns) for n in shape], axis=1).astype(np.int32)
    def fresh():
        out = []
        for st in starts:
            t = np.full(shape, np.inf, np.float32); t[tuple(st)] = 0; out.append(t)
        return out
    ref = fresh()
    solve(1, None, v, fs, starts, ref)
    oks = []
    for pm in (None, 0, 1 << 20):
        a = fresh()
        rc, st = solve(2, pm, v, fs, starts, a)
        oks.append(st["kernel_variant"] == 2 and all(np.array_equal(x, y) for x, y in zip(a, ref)))
    print(f"case {case}: shape {shape} star {star} starts {ns}: default/two-plane/one-plane == cell: {oks}", flush=True)
    bad += sum(not o for o in oks)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
