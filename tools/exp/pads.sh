#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for lib in uoparallel-seismic-project_amd/csrc/libttsweep.so gpurun_exp/ZPAD_24.so gpurun_exp/ZPAD_56.so gpurun_exp/ZPAD_120.so gpurun_exp/YPAD_1.so; do
  echo "== $lib"
  TTSWEEP_EXPERIMENT_LIB=$lib python tools/exp/one_sweep.py 1024,1024,512 14 2>&1 | tail -1
  TTSWEEP_EXPERIMENT_LIB=$lib python tools/exp/one_sweep.py 512,512,256 8 2>&1 | tail -1
done
