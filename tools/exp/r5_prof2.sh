#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_prof2.txt; : > $O
for lib in stripprof stripprof_g1; do
for n in 1 3; do
  for cfg in - waves=8; do
    echo "== $lib: $n starts, $cfg" >> $O
    REPS=2 TTSWEEP_LIB=gpurun_exp/$lib.so timeout -k 10 200 python tools/exp/r5_sweep.py $n $cfg 2>&1 | grep -E "^prof" | tail -1 >> $O
  done
done
done
cat $O
