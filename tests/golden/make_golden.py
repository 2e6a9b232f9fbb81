#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF.

Runs only where the reference checkout exists (this container): it drives the
unmodified reference translation unit (oracle/_ref/libttref.so, built by
oracle/Makefile from /root/reference/serial_new/sweep-tt-multistart.c) and
records inputs and the reference's outputs.  The fixtures are data only
(velocity values, star offsets, start points, travel times, change counts).

  python tests/golden/make_golden.py            # small fixtures (seconds)
  python tests/golden/make_golden.py --big      # + 241x241x51 digests (minutes)
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402


def synth(nx, ny, nz, seed):
    """SURVEY.md Appendix C model."""
    i = np.arange(nx)[:, None, None]
    j = np.arange(ny)[None, :, None]
    k = np.arange(nz)[None, None, :]
    base = (0.18 + 0.10 * k / (nz - 1) + 0.02 * np.sin(i / 9) * np.cos(j / 11)).astype(np.float32)
    u = np.random.default_rng(seed).uniform(-0.005, 0.005, size=(nx, ny, nz))
    return base + u.astype(np.float32)


def shipped(name):
    return O.read_triples(os.path.join(ROOT, "data", "stars", f"{name}-FS.txt"))


# deliberately NOT point-symmetric; the last entry is excluded by the reference's
# exclusive bound (serial_new/sweep-tt-multistart.c:160)
NONSYM = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, -1, 0], [0, -1, -1], [2, 1, -1]], np.int32)
# 6-neighbour shell + a sacrificial last entry
SIX = np.array([[-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1], [1, 1, 1]], np.int32)


def stars():
    return {"3": shipped("3"), "5": shipped("5"), "818": shipped("818"), "nonsym": NONSYM, "six": SIX}


def start_set(shape, offs):
    n = np.array(shape)
    last = np.array(offs[-1])
    mid = n // 2
    # "deadin": start - off[last] lies inside the grid (the dead edge of SURVEY 0-3 exists);
    # "deadout": it lies outside (no dead edge)
    deadin = np.clip(mid + [0, 1, 0], np.maximum(last, 0), n - 1 + np.minimum(last, 0))
    deadout = np.array([n[0] // 3, n[1] // 3, n[2] - 1])
    ax = int(np.flatnonzero(last)[0])
    deadout[ax] = 0 if last[ax] > 0 else n[ax] - 1
    assert np.all(deadin - last >= 0) and np.all(deadin - last < n) and np.all(deadin < n)
    assert np.any(deadout - last < 0) or np.any(deadout - last >= n)
    return {"mid": mid, "corner": np.array([0, 0, 0]), "deadin": deadin, "deadout": deadout}


def small_fixtures():
    grids = {"g24": ((24, 20, 12), 1), "g9": ((9, 7, 5), 2)}
    for gname, (shape, seed) in grids.items():
        v = synth(*shape, seed)
        out = {"v": v}
        meta = {}
        for sname, offs in stars().items():
            out[f"star_{sname}"] = offs
            for stname, st in start_set(shape, offs).items():
                st = np.asarray(st, np.int32)
                (tt,), sweeps = O.ref_converge(v, offs, [st])
                key = f"{sname}_{stname}"
                out[f"tt_{key}"] = tt
                out[f"start_{key}"] = st
                meta[key] = {"sweeps": int(sweeps)}
        # order-dependent single-pass states + change counts pin the sweep body itself
        for sname in ("3", "818", "nonsym"):
            offs = stars()[sname]
            st = np.asarray(start_set(shape, offs)["mid"], np.int32)
            R = O.ref()
            assert R.ttref_setup(*shape, v.reshape(-1), len(offs), offs.reshape(-1).copy(), 10.0, 1,
                                 st.copy())
            counts = []
            for n in range(3):
                counts.append(R.ttref_sweep_default(0))
                out[f"pass{n + 1}_{sname}"] = np.ctypeslib.as_array(R.ttref_tt(0), shape=shape).copy()
            R.ttref_teardown()
            meta[f"pass_{sname}"] = {"counts": counts, "start": st.tolist()}
        # a sub-range of the star (general starstart/starstop)
        offs = stars()["3"]
        st = np.asarray(start_set(shape, offs)["mid"], np.int32)
        (tt,), sweeps = O.ref_converge(v, offs, [st], starstart=5, starstop=60)
        out["tt_3_range_5_60"] = tt
        meta["3_range_5_60"] = {"sweeps": int(sweeps), "start": st.tolist()}
        # the scaled lengths the reference main() computes (:122,:127)
        R = O.ref()
        offs = stars()["818"]
        assert R.ttref_setup(*shape, v.reshape(-1), len(offs), offs.reshape(-1).copy(), 10.0, 1,
                             np.zeros(3, np.int32))
        out["fs_d_818"] = np.array([R.ttref_fs_d(l) for l in range(len(offs))], np.float32)
        R.ttref_teardown()
        out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"{gname}.npz"), **out)
        print(gname, "cases:", len(meta))


def vbox_fixture():
    """A VBOX file written by the reference writer: pins the byte format + checksum."""
    v = synth(6, 5, 4, 3)
    v[1, 2, 3] = -1.5           # bytes >= 0x80 in every position
    path = os.path.join(HERE, "ref_written_6x5x4.vbox")
    assert O.ref().ttref_store_vbox(path.encode(), 1, 1, 1, 6, 5, 4, v.reshape(-1).copy())
    np.save(os.path.join(HERE, "ref_written_6x5x4_values.npy"), v)
    print("vbox fixture", os.path.getsize(path), "bytes")


def big_digests():
    """Converged 241x241x51 boxes of the reference: SHA-256 + spot values only."""
    v = synth(241, 241, 51, 20160507)
    res = {}
    path = os.path.join(HERE, "big_digests.json")
    if os.path.exists(path):
        res = json.load(open(path))
    for sname, st in (("3", (120, 120, 50)), ("818", (120, 120, 50))):
        key = f"syn241_{sname}_{st[0]}_{st[1]}_{st[2]}"
        if key in res:
            continue
        t0 = time.time()
        (tt,), sweeps = O.ref_converge(v, shipped(sname), [np.array(st, np.int32)])
        spots = [(0, 0, 0), (113, 119, 49), (118, 118, 49), (240, 240, 0), (0, 240, 50), (60, 200, 25)]
        res[key] = {
            "sha256": hashlib.sha256(tt.tobytes()).hexdigest(),
            "sweeps": int(sweeps),
            "seconds": round(time.time() - t0, 1),
            "spots": {",".join(map(str, p)): int(tt[p].view(np.uint32)) for p in spots},
            "max": float(tt.max()), "mean": float(tt.astype(np.float64).mean()),
        }
        print(key, res[key]["sha256"], sweeps, "sweeps", res[key]["seconds"], "s", flush=True)
        json.dump(res, open(path, "w"), indent=1)


def big_start24(idx, which="24", star="818"):
    """One start of a BASELINE start file (start-24 by default, --big-file 4 for start-4) on
    the 241x241x51 model, 818-FS, through the reference; written to its own file so several
    can run in parallel (merge_big)."""
    v = synth(241, 241, 51, 20160507)
    starts = O.read_triples(os.path.join(ROOT, "data", "starts", f"start-{which}-241-241-51.txt"))
    st = starts[idx]
    t0 = time.time()
    (tt,), sweeps = O.ref_converge(v, shipped(star), [st.astype(np.int32)])
    res = {f"syn241_{star}_{st[0]}_{st[1]}_{st[2]}": {
        "sha256": hashlib.sha256(tt.tobytes()).hexdigest(), "sweeps": int(sweeps),
        "seconds": round(time.time() - t0, 1), "spots": {}, f"start{which}_index": int(idx),
        "max": float(tt.max()), "mean": float(tt.astype(np.float64).mean())}}
    json.dump(res, open(os.path.join(HERE, f"big_part_{star}_{which}_{idx}.json"), "w"), indent=1)
    print(res, flush=True)


def merge_big():
    path = os.path.join(HERE, "big_digests.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    for f in sorted(os.listdir(HERE)):
        if f.startswith("big_part_") and f.endswith(".json"):
            res.update(json.load(open(os.path.join(HERE, f))))
            os.remove(os.path.join(HERE, f))
    json.dump(res, open(path, "w"), indent=1)
    print(sorted(res))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--big-start", type=int, default=None)
    ap.add_argument("--big-file", default="24")
    ap.add_argument("--big-star", default="818")
    ap.add_argument("--merge-big", action="store_true")
    args = ap.parse_args()
    if args.big_start is not None:
        big_start24(args.big_start, args.big_file, args.big_star)
        sys.exit(0)
    if args.merge_big:
        merge_big()
        sys.exit(0)
    if O.ref() is None:
        sys.exit("reference not available: golden vectors can only be generated beside /root/reference")
    small_fixtures()
    vbox_fixture()
    if args.big:
        big_digests()
