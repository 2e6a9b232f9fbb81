import sys, numpy as np
sys.path.insert(0,'/root/repo')
import ttsweep_pkg; P=ttsweep_pkg.load()
v=P.inputs.velocity_model(241,241,51,20160507)
fs=P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path('818')))
starts=P.inputs.read_triples(P.inputs.starts_path('24'))[:1]
tt=[np.full(v.shape,np.inf,np.float32)]; tt[0][tuple(starts[0])]=0
with P.TravelTimeSolver(v.shape,fs) as sol:
    sol.set_velocity(v); sol.solve(starts,tt); print(sol.stats())
