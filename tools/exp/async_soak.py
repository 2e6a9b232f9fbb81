"""Soak run of the one-launch STRIP solve: the 24-start headline workload N times under randomly drawn schedule
knobs (fill marks, gate speeds, dead-edge interval, deferral margin, unit size), every result compared on the
device with the first one, the first one with the reference's SHA-256 digests.  A stale read that cost an update
would show as a box that differs.  usage: async_soak.py [N] [seed]"""
import os, sys, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ttsweep_pkg
P = ttsweep_pkg.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dig = json.load(open(os.path.join(root, "tests", "golden", "big_digests.json")))
want = {}
for key, w in dig.items():
    _, sname, i, j, k = key.split("_")
    if sname == "818": want[(int(i), int(j), int(k))] = w["sha256"]
v = P.inputs.velocity_model(241, 241, 51, 20160507)
fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("818")))
starts = np.asarray(P.inputs.read_triples(P.inputs.starts_path("24")), dtype=np.int32)
dev = torch.device("cuda:0")
tt = torch.empty((len(starts),) + v.shape, dtype=torch.float32, device=dev)
first = None
bad = 0
with P.TravelTimeSolver(v.shape, fs) as sol:
    sol.set_velocity(v)
    sol.set_option(P.OPT_ASYNC, 1)
    for it in range(N):
        if it:
            sol.set_option(P.OPT_ASYNC_LOW, int(rng.integers(1, 200)))
            sol.set_option(P.OPT_ASYNC_HIGH, int(rng.integers(2, 600)))
            sol.set_option(P.OPT_ASYNC_GATE_MILLI, int(rng.integers(300, 4000)))
            sol.set_option(P.OPT_ASYNC_GATE_FAST_MILLI, int(rng.integers(300, 6000)))
            sol.set_option(P.OPT_ASYNC_SPECIAL, int(rng.integers(1, 1000)))
            sol.set_option(P.OPT_DEFER_MARGIN_MILLI, int(rng.integers(-500, 4000)))
            sol.set_option(P.OPT_ASYNC_INUNIT, int(rng.integers(-1, 5)))
            sol.set_option(P.OPT_PAIR_MIN_STARTS, int(rng.choice([0, 1 << 20])))
            # round 5: direct hand-off, the eight-wave instance of small shards (one-plane units), a short wall-clock limit
            # on the launch (a solve that gives up would show as a fallback, below)
            sol.set_option(P.OPT_ASYNC_HANDOFF, int(rng.integers(-1, 4)))
            sol.set_option(P.OPT_ASYNC_WAVES, int(rng.choice([-1, 4, 8])))
            sol.set_option(P.OPT_ASYNC_TIMEOUT_MILLI, 2000)
        nst = int(rng.integers(1, len(starts) + 1)) if it else len(starts)
        rc = sol.solve_device(starts[:nst], tt[:nst], init=True)
        torch.cuda.synchronize()
        assert rc == 1
        if sol.stats()["fallbacks"]:
            print(f"solve {it}: the launch gave up ({nst} starts)", flush=True)
            bad += 1
        if first is None:
            first = tt.clone()
            host = first.cpu().numpy()
            wrong = sum(1 for s, box in zip(starts, host) if want.get(tuple(int(x) for x in s)) != hashlib.sha256(box.tobytes()).hexdigest())
            print(f"first solve: {wrong} of {len(starts)} boxes differ from the reference digests", flush=True)
            bad += wrong
        elif not torch.equal(tt[:nst], first[:nst]):
            ndiff = int((tt[:nst] != first[:nst]).sum().item())
            print(f"solve {it}: {ndiff} cells differ ({nst} starts)", flush=True)
            bad += 1
        if it % 50 == 49: print(f"{it + 1} solves, {bad} bad", flush=True)
print("BAD:", bad)
sys.exit(1 if bad else 0)
