"""Repeats the (40, 24, 160) three-start case of test_tile_six_star_drivers_vs_oracle under the column driver and
compares with the hyperplane driver; with a -DTTSWEEP_COL_TRACE build ($TTSWEEP_LIB) prints the protocol events
around the first tile that differs."""
import os, sys, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ttsweep_pkg
P = ttsweep_pkg.load()
if os.environ.get("TTSWEEP_LIB"):
    P._lib.use_library(os.environ["TTSWEEP_LIB"])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
in_place = int(sys.argv[2]) if len(sys.argv) > 2 else 1
shape, seed = (40, 24, 160), 56
rng = np.random.default_rng(seed)
v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
six = P.inputs.read_triples(P.inputs.star_path("six"))
fs = P.inputs.make_fs(six)
nstart = int(rng.integers(1, 4))
starts = np.stack([rng.integers(0, n, size=nstart) for n in shape], axis=1).astype(np.int32)
dev = torch.device("cuda:0")
trace = os.environ.get("TTSWEEP_COL_TRACE_FILE", "/tmp/col_trace.bin")
with P.TravelTimeSolver(shape, fs) as ref:
    ref.set_option(P.OPT_KERNEL, 3); ref.set_option(P.OPT_ASYNC, 0); ref.set_velocity(v)
    want = torch.empty((nstart,) + shape, dtype=torch.float32, device=dev)
    assert ref.solve_device(starts, want, init=True) == 1
want = want.cpu().numpy()
NI, NJ, NK = (shape[0] + 7) // 8, (shape[1] + 7) // 8, (shape[2] + 31) // 32
fails = 0
with P.TravelTimeSolver(shape, fs) as sol:
    sol.set_option(P.OPT_KERNEL, 3); sol.set_option(P.OPT_TILE_IN_PLACE, in_place); sol.set_velocity(v)
    for rep in range(reps):
        tt = torch.empty((nstart,) + shape, dtype=torch.float32, device=dev)
        rc = sol.solve_device(starts, tt, init=True)
        st = sol.stats()
        got = tt.cpu().numpy()
        bad = np.argwhere(got != want)
        if len(bad) == 0:
            continue
        fails += 1
        s, x, y, z = (int(t) for t in bad[0])
        print(f"rep {rep}: {len(bad)} cells differ, first start {s} cell ({x},{y},{z}) got {got[s,x,y,z]} want {want[s,x,y,z]}; "
              f"sweeps {st['sweeps_total']} fallbacks {st['fallbacks']}", flush=True)
        tiles = sorted({(int(b[0]), int(b[1]) // 8, int(b[2]) // 8, int(b[3]) // 32) for b in bad})
        print("  tiles (s, I, J, K):", tiles)
        if os.path.exists(trace) and fails <= 2:
            rec = np.fromfile(trace, dtype=np.uint32).reshape(-1, 8)
            np.save(f"gpurun_out/col_trace_{fails}.npy", rec)
            _, I, J, K = tiles[0]
            near = {(I, J), (I - 1, J), (I + 1, J), (I, J - 1), (I, J + 1)}
            rows = [r for r in rec if int(r[1]) == s and (int(r[3]) // NJ, int(r[3]) % NJ) in near]
            rows.sort(key=lambda r: (int(r[2]), int(r[7])))
            print(f"  events of start {s} around column ({I},{J}), tile K={K} (NK {NK}); order o = gray(e-1)")
            for r in rows:
                t, e, col = int(r[0]), int(r[2]), int(r[3])
                o = ((e - 1) ^ ((e - 1) >> 1)) & 7
                name = {1: "begin", 2: "run  ", 3: "seal ", 4: "REST "}[t]
                if t == 1: txt = f"mask0 {int(r[4]):05b} known_up {int(r[5])} upmask {int(r[6]):05b}"
                elif t == 2: txt = f"k0 {int(r[4]) & 255} nt {(int(r[4]) >> 8) & 255} kend {int(r[4]) >> 16} tilebits {int(r[5]):05b} duebits {int(r[6]):05b}"
                elif t == 3: txt = f"mymask {int(r[4]):05b} o {int(r[5])}"
                else: txt = f"seen {int(r[4])} improved {int(r[5])}"
                print(f"    e {e:3d} o {o} sz {'-' if o & 4 else '+'} col ({col // NJ},{col % NJ}) {name} {txt}  clk {int(r[7])}")
        if fails >= 3:
            break
print("FAILURES:", fails, "of", reps)
