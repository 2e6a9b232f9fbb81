"""Randomised cross-check of the TILE kernel against the CELL kernel (both on the GPU, bit for bit):
random grid shapes, random small stars within TILE's reach, 1-4 starts, fresh and resumed solves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, ttsweep_pkg
P = ttsweep_pkg.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
six = P.inputs.read_triples(P.inputs.star_path("six"))
shell = np.array([[a, b, c] for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)] + [[1, 1, 1]], np.int32)
def solve(kernel, v, fs, starts, tts):
    with P.TravelTimeSolver(v.shape, fs) as sol:
        sol.set_option(P.OPT_KERNEL, kernel)
        if kernel == 3 and rng.integers(0, 2): sol.set_option(P.OPT_TILE_ORDER, int(rng.integers(0, 10)) + 10 * int(rng.integers(0, 3)) + 100 * int(rng.integers(0, 5)))     # (round 5; else the default)
        sol.set_velocity(v)
        rc = sol.solve(starts, tts)
        return rc, sol.stats()
bad = 0
for case in range(ncase):
    shape = tuple(int(x) for x in rng.integers(1, [150, 150, 200]))
    v = rng.uniform(0.05, 1.0, size=shape).astype(np.float32)
    kind = case % 3
    if kind == 0: offs = six
    elif kind == 1: offs = shell
    else:
        n = int(rng.integers(2, 14))
        offs = np.stack([rng.integers(-2, 3, size=n), rng.integers(-2, 3, size=n), rng.integers(-4, 5, size=n)], axis=1)
        offs = offs[np.any(offs != 0, axis=1)].astype(np.int32)
        if len(offs) < 2: offs = six
    fs = P.inputs.make_fs(offs)
    ns = int(rng.integers(1, 5))
    starts = np.stack([rng.integers(0, n, size=ns) for n in shape], axis=1).astype(np.int32)
    def fresh():
        out = []
        for st in starts:
            t = np.full(shape, np.inf, np.float32); t[tuple(st)] = 0; out.append(t)
        return out
    a, b = fresh(), fresh()
    rc_t, st_t = solve(3, v, fs, starts, a)
    rc_c, st_c = solve(1, v, fs, starts, b)
    ok = st_t["kernel_variant"] == 3 and all(np.array_equal(x, y) for x, y in zip(a, b))
    # resume: damage the TILE result, solve again with TILE
    dmg = [t.copy() for t in a]
    for t, st in zip(dmg, starts):
        lo = [int(rng.integers(0, n)) for n in shape]; hi = [min(n, l + int(rng.integers(1, 60))) for n, l in zip(shape, lo)]
        t[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = np.inf
        t[tuple(st)] = 0
    rc_r, _ = solve(3, v, fs, starts, dmg)
    ok2 = all(np.array_equal(x, y) for x, y in zip(dmg, a))
    print(f"case {case}: shape {shape} star {['six','shell26','random'][kind]} ({len(offs)} offs) starts {ns}: "
          f"tile==cell {ok}, resumed==tile {ok2}, sweeps tile {st_t['sweeps_total']}", flush=True)
    bad += (not ok) + (not ok2)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
