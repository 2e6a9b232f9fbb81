#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for lib in uoparallel-seismic-project_amd/csrc/libttsweep.so gpurun_exp/x_*.so; do
  echo "== $lib"
  SKIP_CONVERGE=1 TTSWEEP_EXPERIMENT_LIB=$lib python tools/exp/one_sweep.py 1024,1024,512 14 2>&1 | tail -1
done
