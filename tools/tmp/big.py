import sys, numpy as np, torch, time, os
sys.path.insert(0,'/root/repo')
import ttsweep_pkg; P=ttsweep_pkg.load()
nx,ny,nz = map(int, os.environ.get('GRID','1024,1024,512').split(','))
dev=torch.device('cuda:0')
fs=P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path('818')))
starts=P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path('24'))[:2], nx,ny,nz)
v=P.inputs.velocity_model_device(nx,ny,nz,20160507,dev)
with P.TravelTimeSolver((nx,ny,nz),fs) as sol:
    sol.set_velocity(v)
    tt=torch.empty((len(starts),nx,ny,nz),dtype=torch.float32,device=dev)
    sol.solve_device(starts,tt,init=True)
    st=sol.stats(); print('solve_ms', st['solve_ms'], 'passes', st['sweeps_max'], 'eq', st['cells_relaxed']/st['cells']/len(starts))
    for s in range(len(starts)):
        print('start', starts[s], 'validate', sol.validate_device(starts[s], tt[s]), 'min', float(tt[s].min()), 'max', float(tt[s].max()))
