"""Experiment: does the six-FS column solve depend on where the caller's boxes lie?  One process, the same solve with the
travel-time tensor carved out of one big allocation at different byte offsets, and with a gap between the boxes of the starts.
python tools/exp/col_align.py [nx,ny,nz] [nstart]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ttsweep_pkg
P = ttsweep_pkg.load()
shape = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024,1024,512").split(","))
nstart = int(sys.argv[2]) if len(sys.argv) > 2 else 14
dev = torch.device("cuda:0")
v = P.inputs.velocity_model_device(*shape, 20160507, dev)
fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("six")))
starts = P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path("111")), *shape)[:nstart]
cells = shape[0] * shape[1] * shape[2]
big = torch.empty(nstart * cells + (64 << 20), dtype=torch.float32, device=dev)
print("base address mod 2^32: %#x" % (big.data_ptr() & 0xffffffff))
with P.TravelTimeSolver(shape, fs) as sol:
    sol.set_option(P.OPT_TIMING, 1)
    sol.set_velocity(v)
    for off_bytes in (0, 64, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 256, 16 << 20, (16 << 20) + 65536 + 4096, 0):
        o = off_bytes // 4
        tt = big[o:o + nstart * cells].view((nstart,) + shape)
        times = []
        for rep in range(3):
            sol.solve_device(starts, tt, init=True)
            torch.cuda.synchronize()
            times.append(sol.stats()["solve_ms"])
        print(f"offset {off_bytes:>10} B: solve {min(times):7.2f} .. {max(times):7.2f} ms, kernel {sol.stats()['sweep_kernel_ms']:.2f}", flush=True)
