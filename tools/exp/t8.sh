#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for lib in gpurun_exp/twg4.so gpurun_exp/twg3.so; do
TTSWEEP_EXPERIMENT_LIB=$lib python tools/exp/one_sweep.py 1024,1024,512 14 2>&1 | tail -1
TTSWEEP_EXPERIMENT_LIB=$lib python tools/exp/one_sweep.py 512,512,256 8 2>&1 | tail -1
done
