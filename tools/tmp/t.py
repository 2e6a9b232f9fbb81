import sys, numpy as np, torch, time, os
sys.path.insert(0,'/root/repo')
import ttsweep_pkg; P=ttsweep_pkg.load()
v=P.inputs.velocity_model(241,241,51,20160507)
fs=P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path('818')))
starts=P.inputs.read_triples(P.inputs.starts_path('24'))[:int(os.environ.get('NST','24'))]
dev=torch.device('cuda:0')
with P.TravelTimeSolver(v.shape,fs) as sol:
    sol.set_velocity(torch.from_numpy(v).to(dev))
    tt=torch.empty((len(starts),)+v.shape,dtype=torch.float32,device=dev)
    for i in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter()
        sol.solve_device(starts,tt,init=True)
        torch.cuda.synchronize(); t1=time.perf_counter()
        print('wall ms', (t1-t0)*1e3, 'solve_ms', sol.stats()['solve_ms'])
