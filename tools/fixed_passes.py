# timing experiment: run the first N passes of the 24-start solve and report kernel time
import sys, os, numpy as np, torch
sys.path.insert(0,'/root/repo')
import ttsweep_pkg; P=ttsweep_pkg.load()
v=P.inputs.velocity_model(241,241,51,20160507)
fs=P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path('818')))
starts=P.inputs.read_triples(P.inputs.starts_path('24'))
dev=torch.device('cuda:0')
with P.TravelTimeSolver(v.shape,fs) as sol:
    sol.set_velocity(torch.from_numpy(v).to(dev))
    sol.set_option(P.OPT_TIMING,1)
    sol.set_option(P.OPT_MAX_SWEEPS,int(sys.argv[1]))
    tt=torch.empty((len(starts),)+v.shape,dtype=torch.float32,device=dev)
    try: sol.solve_device(starts,tt,init=True)
    except Exception as e: pass
    st=sol.stats(); print(st)
