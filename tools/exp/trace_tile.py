"""Per-sweep activity of a TILE solve (-DTTSWEEP_DEBUG_ENV build, TTSWEEP_TRACE=1)."""
import os, sys
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
    os.environ["TTSWEEP_TRACE"] = "1"
    sol.solve_device(starts, tt, init=True)
