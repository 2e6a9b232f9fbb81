#!/usr/bin/env python3
"""Parity smoke test of an A/B build of libttsweep.so (python tools/exp/check_lib.py gpurun_exp/NAME.so):
the golden 818 / 5 / 3-FS fixtures with units of one and of two planes, and the SHA-256 digests of
the first starts of the benchmark workload.  Exit code 0 = bit-exact."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ttsweep_pkg

P = ttsweep_pkg.load()
if len(sys.argv) > 1:
    P._lib.use_library(sys.argv[1])
z = np.load(os.path.join(ROOT, "tests", "golden", "g24.npz"))
v = z["v"]
bad = 0
for star in ("818", "5", "3"):
    fs = P.inputs.make_fs(z[f"star_{star}"])
    keys = [f"{star}_{k}" for k in ("mid", "corner", "deadin", "deadout")]
    starts = np.array([z[f"start_{k}"] for k in keys], dtype=np.int32)
    for pair_min in (1 << 20, 0):
        with P.TravelTimeSolver(v.shape, fs) as sol:
            sol.set_option(P.OPT_KERNEL, 2)
            sol.set_option(P.OPT_PAIR_MIN_STARTS, pair_min)
            sol.set_velocity(v)
            tts = []
            for st in starts:
                tt = np.full(v.shape, np.inf, dtype=np.float32)
                tt[tuple(st)] = 0
                tts.append(tt)
            sol.solve(starts, tts)
        for k, tt in zip(keys, tts):
            if not np.array_equal(tt.view(np.uint32), z[f"tt_{k}"].view(np.uint32)):
                print("MISMATCH", k, "pair_min", pair_min)
                bad += 1
dig = json.load(open(os.path.join(ROOT, "tests", "golden", "big_digests.json")))
shape = (241, 241, 51)
vv = P.inputs.velocity_model(*shape, 20160507)
starts = P.inputs.read_triples(P.inputs.starts_path("24"))
fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("818")))
import torch
with P.TravelTimeSolver(shape, fs) as sol:
    sol.set_velocity(vv)
    tt = torch.empty((len(starts),) + shape, dtype=torch.float32, device="cuda:0")
    sol.solve_device(starts, tt, init=True)
    for n, (i, j, k) in enumerate(starts):
        rec = dig.get(f"syn241_818_{i}_{j}_{k}")
        if rec and hashlib.sha256(tt[n].cpu().numpy().tobytes()).hexdigest() != rec["sha256"]:
            print("DIGEST MISMATCH start", n)
            bad += 1
print("check_lib:", sys.argv[1] if len(sys.argv) > 1 else "default", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
