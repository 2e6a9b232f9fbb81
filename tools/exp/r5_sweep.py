"""Knob sweep of the one-launch STRIP solve on the headline grid, named knobs (round 5).
usage: r5_sweep.py nstarts[,nstarts...] cfg [cfg ...]      cfg = comma-separated key=value, '-' for the defaults
keys: handoff, waves, inunit, gate (milli cells per round), fast (milli, gate while the ring is empty), margin (milli),
      pair (TTSWEEP_OPT_PAIR_MIN_STARTS), low, high, special, policy, async, queues
Every solve's boxes are compared with the reference's SHA-256 digests (tests/golden/big_digests.json)."""
import os, sys, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ttsweep_pkg
P = ttsweep_pkg.load()
if os.environ.get("TTSWEEP_LIB"): P._lib.use_library(os.environ["TTSWEEP_LIB"])
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dig = json.load(open(os.path.join(root, "tests", "golden", "big_digests.json")))
want = {}
for key, w in dig.items():
    _, sname, i, j, k = key.split("_")
    if sname == "818": want[(int(i), int(j), int(k))] = w["sha256"]
fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("818")))
if os.environ.get("GRID"):      # another grid (GRID=512,512,256): the starts of start-111 scaled onto it, boxes compared across the configurations
    shape = tuple(int(x) for x in os.environ["GRID"].split(","))
    v = P.inputs.velocity_model_device(*shape, 20160507, torch.device("cuda:0"))
    all_starts = np.asarray(P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path("111")), *shape), dtype=np.int32)
    want = {}
else:
    v = P.inputs.velocity_model(241, 241, 51, 20160507)
    all_starts = np.asarray(P.inputs.read_triples(P.inputs.starts_path("24")), dtype=np.int32)
first_sum = {}
KEYS = {"handoff": "OPT_ASYNC_HANDOFF", "waves": "OPT_ASYNC_WAVES", "inunit": "OPT_ASYNC_INUNIT", "gate": "OPT_ASYNC_GATE_MILLI",
        "fast": "OPT_ASYNC_GATE_FAST_MILLI", "margin": "OPT_DEFER_MARGIN_MILLI", "pair": "OPT_PAIR_MIN_STARTS",
        "low": "OPT_ASYNC_LOW", "high": "OPT_ASYNC_HIGH", "special": "OPT_ASYNC_SPECIAL", "policy": "OPT_ASYNC_POLICY",
        "async": "OPT_ASYNC", "queues": "OPT_QUEUES", "timeout": "OPT_ASYNC_TIMEOUT_MILLI"}
reps = int(os.environ.get("REPS", "5"))
dev = torch.device("cuda:0")
for nst in [int(x) for x in sys.argv[1].split(",")]:
    starts = all_starts[:nst]
    tt = torch.empty((len(starts),) + v.shape, dtype=torch.float32, device=dev)
    print(f"== {nst} starts", flush=True)
    for cfg in sys.argv[2:]:
        with P.TravelTimeSolver(v.shape, fs) as sol:
            sol.set_option(P.OPT_TIMING, 1)
            pre = [kv for kv in cfg.split(",") if kv not in ("", "-") and kv.split("=")[0] in ("pair", "queues")]
            post = [kv for kv in cfg.split(",") if kv not in ("", "-") and kv.split("=")[0] not in ("pair", "queues")]
            for kv in pre:
                k, val = kv.split("=")
                sol.set_option(getattr(P, KEYS[k]), int(val))
            sol.set_velocity(v)
            for kv in post:
                k, val = kv.split("=")
                sol.set_option(getattr(P, KEYS[k]), int(val))
            times, best = [], None
            for rep in range(reps):
                sol.solve_device(starts, tt, init=True)
                torch.cuda.synchronize()
                st = sol.stats()
                times.append(st["solve_ms"])
                if best is None or st["solve_ms"] < best["solve_ms"]: best = st
            if want:
                host = tt.cpu().numpy()
                bad = sum(1 for s, box in zip(starts, host)
                          if want.get(tuple(int(x) for x in s)) not in (None, hashlib.sha256(box.tobytes()).hexdigest()))
            else:
                sums = [int(tt[i].view(torch.int32).to(torch.int64).sum().item()) for i in range(len(starts))]
                bad = sum(1 for a, b in zip(sums, first_sum.setdefault(nst, sums)) if a != b)
            times.sort()
            print(f"{cfg:>40}: solve min {times[0]:7.3f} med {times[len(times) // 2]:7.3f} ms  kernel {best['sweep_kernel_ms']:7.3f}  "
                  f"sweep-eq {best['cells_relaxed'] / best['cells'] / len(starts):6.3f}  fallbacks {best['fallbacks']}  bad boxes {bad}",
                  flush=True)
