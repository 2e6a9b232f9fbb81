"""Per-pass activity of a solve (needs a -DTTSWEEP_DEBUG_ENV build and TTSWEEP_TRACE=1): the
first N benchmark starts as one shard, traced pass by pass (the trace serialises the passes),
then the same solve timed untraced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ttsweep_pkg
P = ttsweep_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shape = (241, 241, 51)
dev = torch.device("cuda:0")
v = torch.from_numpy(P.inputs.velocity_model(*shape, 20160507)).to(dev)
fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("818")))
starts = P.inputs.read_triples(P.inputs.starts_path("24"))
shard = P.multistart.all_shards(len(starts), len(starts) // n, starts, shape)[0]
print("shard", shard, flush=True)
sol = P.TravelTimeSolver(shape, fs)
sol.set_velocity(v)
tt = torch.empty((len(shard),) + shape, dtype=torch.float32, device=dev)
os.environ.pop("TTSWEEP_TRACE", None)
for _ in range(3): sol.solve_device(starts[shard], tt, init=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): sol.solve_device(starts[shard], tt, init=True)
torch.cuda.synchronize()
print(f"untraced: {1e2 * (time.perf_counter() - t0):.2f} ms per solve", flush=True)
os.environ["TTSWEEP_TRACE"] = "1"
sol.solve_device(starts[shard], tt, init=True)
