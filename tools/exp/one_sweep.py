"""Experiment: time ONE full ordering sweep of the TILE kernel over every tile (a converged box
solved again: every tile is due once, nothing improves, nothing is stored)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ttsweep_pkg
P = ttsweep_pkg.load()
shape = tuple(int(x) for x in sys.argv[1].split(","))
nstart = int(sys.argv[2])
dev = torch.device("cuda:0")
v = P.inputs.velocity_model_device(*shape, 20160507, dev)
fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("six")))
starts = P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path("111")), *shape)[:nstart]
with P.TravelTimeSolver(shape, fs) as sol:
    sol.set_velocity(v)
    tt = torch.empty((nstart,) + shape, dtype=torch.float32, device=dev)
    if os.environ.get("SKIP_CONVERGE") != "1":
        sol.solve_device(starts, tt, init=True)
    else:
        tt.fill_(1.0)
    sol.set_option(P.OPT_TIMING, 1)
    for rep in range(2):
        sol.solve_device(starts, tt, init=False)
        st = sol.stats()
        cells = shape[0] * shape[1] * shape[2]
        gb = 12.0 * st["cells_relaxed"] / 1e9
        print(f"{shape} x {nstart}: sweeps {st['sweeps_total']} launches {st['launches']} kernel ms {st['sweep_kernel_ms']:.2f} "
              f"grid-equivalents {st['cells_relaxed'] / cells / nstart:.2f} algorithmic {gb / (st['sweep_kernel_ms'] / 1e3):.0f} GB/s")
