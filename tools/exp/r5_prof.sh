#!/bin/bash
# phase stamps (-DTTSWEEP_PROFILE build, gpurun_exp/stripprof.so) of the 4- and 8-wave unit kernels, 1 / 3 / 8 starts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_prof.txt; : > $O
for n in 1 3 8; do
  for cfg in - waves=8 "$@"; do
    echo "== $n starts, $cfg" >> $O
    REPS=2 TTSWEEP_LIB=gpurun_exp/stripprof.so timeout -k 10 200 python tools/exp/r5_sweep.py $n $cfg 2>&1 | grep -E "^prof|solve" | tail -2 >> $O
  done
done
cat $O
