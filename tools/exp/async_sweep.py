"""Knob sweep of the one-launch STRIP solve on the headline grid.
usage: async_sweep.py nstarts cfg [cfg ...]   cfg = async:pair:low:high:special:policy:gate_milli:margin_milli[:fast_gate_milli[:in-unit passes]]  (pair -1 = default rule, gate -1 = default, margin -1000000000 = off)"""
import os, sys, json, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ttsweep_pkg
P = ttsweep_pkg.load()
if os.environ.get("TTSWEEP_LIB"): P._lib.use_library(os.environ["TTSWEEP_LIB"])
nst = int(sys.argv[1])
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
for cfg in sys.argv[2:]:
    f = [int(x) for x in cfg.split(":")]
    mode, pair, low, high, special, policy, gate, margin = f[:8]
    fast = f[8] if len(f) > 8 else -1
    # (policy 2: `gate` is the window in milli-cells)
    with P.TravelTimeSolver(v.shape, fs) as sol:
        sol.set_option(P.OPT_TIMING, 1)
        if pair >= 0: sol.set_option(P.OPT_PAIR_MIN_STARTS, pair)
        sol.set_velocity(v)
        sol.set_option(P.OPT_ASYNC, mode)
        if low: sol.set_option(P.OPT_ASYNC_LOW, low)
        if high: sol.set_option(P.OPT_ASYNC_HIGH, high)
        if special: sol.set_option(P.OPT_ASYNC_SPECIAL, special)
        sol.set_option(P.OPT_ASYNC_POLICY, policy)
        if gate >= 0: sol.set_option(P.OPT_ASYNC_WINDOW_MILLI if policy == 2 else (P.OPT_ASYNC_GATE_MILLI if mode == 1 else P.OPT_GATE_SPEED_MILLI), gate)
        sol.set_option(P.OPT_DEFER_MARGIN_MILLI, margin)
        if fast >= 0: sol.set_option(P.OPT_ASYNC_GATE_FAST_MILLI, fast)
        if len(f) > 9: sol.set_option(P.OPT_ASYNC_INUNIT, f[9])
        best = None
        for rep in range(3):
            rc = sol.solve_device(starts, tt, init=True)
            torch.cuda.synchronize()
            st = sol.stats()
            if best is None or st["solve_ms"] < best["solve_ms"]: best = st
        host = tt.cpu().numpy()
        bad = sum(1 for s, box in zip(starts, host)
                  if want.get(tuple(int(x) for x in s)) not in (None, hashlib.sha256(box.tobytes()).hexdigest()))
        print(f"{cfg:>24}: solve {best['solve_ms']:7.2f} ms  sweep-eq {best['cells_relaxed'] / best['cells'] / len(starts):6.3f}  "
              f"ms/sweep-eq/start {best['solve_ms'] / (best['cells_relaxed'] / best['cells']):6.3f}  bad boxes {bad}", flush=True)
