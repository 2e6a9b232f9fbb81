"""Experiment: the 24 starts as G independent groups (one context, stream and host thread
each) on ONE GPU, so that the tail of one group's pass overlaps the other's work."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ttsweep_pkg
P = ttsweep_pkg.load()
shape = (241, 241, 51)
dev = torch.device("cuda:0")
v = torch.from_numpy(P.inputs.velocity_model(*shape, 20160507)).to(dev)
fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("818")))
starts = P.inputs.read_triples(P.inputs.starts_path("24"))
for G in (1, 2, 3, 4):
    shards = P.multistart.all_shards(len(starts), G, starts, shape)
    sols, tts = [], []
    for g in range(G):
        sol = P.TravelTimeSolver(shape, fs)
        sol.set_velocity(v)
        sols.append(sol)
        tts.append(torch.empty((len(shards[g]),) + shape, dtype=torch.float32, device=dev))
    def run(g):
        sols[g].solve_device(starts[shards[g]], tts[g], init=True)
    def step():
        th = [threading.Thread(target=run, args=(g,)) for g in range(G)]
        for t in th: t.start()
        for t in th: t.join()
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n): step()
    torch.cuda.synchronize()
    print(f"G={G}: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per 24-start solve", flush=True)
    for s in sols: s.close()
