#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in "512,512,256 8" "1024,1024,512 4" "1024,1024,512 14"; do
  set -- $cfg
  python tools/exp/one_sweep.py $1 $2 2>&1 | tail -2
  TTSWEEP_EXPERIMENT_LIB=gpurun_exp/loadsonly.so SKIP_CONVERGE=1 python tools/exp/one_sweep.py $1 $2 2>&1 | tail -2
done
