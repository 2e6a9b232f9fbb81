"""One-launch (async) STRIP solve against the pass driver on the headline workload: times, work, and the
24 converged boxes against the reference's SHA-256 digests (tests/golden/big_digests.json).
usage: async_probe.py [nstarts] [pair_min] [low] [high] [special] [reps]"""
import os, sys, json, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ttsweep_pkg
P = ttsweep_pkg.load()
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 24
pair = int(sys.argv[2]) if len(sys.argv) > 2 else -1
low = int(sys.argv[3]) if len(sys.argv) > 3 else 0
high = int(sys.argv[4]) if len(sys.argv) > 4 else 0
special = int(sys.argv[5]) if len(sys.argv) > 5 else 0
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dig = json.load(open(os.path.join(root, "tests", "golden", "big_digests.json")))
want = {}
for key, w in dig.items():
    _, sname, i, j, k = key.split("_")
    if sname == "818": want[(int(i), int(j), int(k))] = w["sha256"]
v = P.inputs.velocity_model(241, 241, 51, 20160507)
fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("818")))
starts = np.asarray(P.inputs.read_triples(P.inputs.starts_path("24")), dtype=np.int32)[:nst]
dev = torch.device("cuda:0")
tt = torch.empty((len(starts),) + v.shape, dtype=torch.float32, device=dev)
OPT_ASYNC, OPT_LOW, OPT_HIGH, OPT_SPECIAL = P.OPT_ASYNC, P.OPT_ASYNC_LOW, P.OPT_ASYNC_HIGH, P.OPT_ASYNC_SPECIAL
with P.TravelTimeSolver(v.shape, fs) as sol:
    sol.set_option(P.OPT_TIMING, 1)
    if pair >= 0: sol.set_option(P.OPT_PAIR_MIN_STARTS, pair)
    sol.set_velocity(v)
    for mode in (0, 1, 0, 1):
        sol.set_option(OPT_ASYNC, mode)
        if low: sol.set_option(OPT_LOW, low)
        if high: sol.set_option(OPT_HIGH, high)
        if special: sol.set_option(OPT_SPECIAL, special)
        for rep in range(reps):
            t0 = time.perf_counter()
            rc = sol.solve_device(starts, tt, init=True)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            st = sol.stats()
            print(f"async={mode} rep {rep}: rc {rc} wall {wall:.2f} ms solve {st['solve_ms']:.2f} ms kernels {st['sweep_kernel_ms']:.2f} ms "
                  f"sweep-eq {st['cells_relaxed'] / st['cells'] / len(starts):.3f} sweeps_max {st['sweeps_max']} launches {st['launches']}", flush=True)
        host = tt.cpu().numpy()
        bad = 0; checked = 0
        for s, box in zip(starts, host):
            w = want.get(tuple(int(x) for x in s))
            if w is None: continue
            checked += 1
            if hashlib.sha256(box.tobytes()).hexdigest() != w: bad += 1
        print(f"async={mode}: {checked} boxes checked against the reference digests, {bad} differ", flush=True)
