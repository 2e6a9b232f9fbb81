#!/bin/bash
# teams + plane groups, one workgroup per CU: A/B builds gNwM.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_fifth.txt; : > $O
for b in g3w1 g2w1; do
  echo "=== build $b" >> $O
  TTSWEEP_LIB=gpurun_exp/$b.so timeout -k 10 600 python tools/exp/r5_sweep.py 1,3,8 waves=8 waves=8,inunit=2 waves=8,handoff=1 2>&1 | grep -v amdgpu.ids >> $O
  for n in 1 3; do
    echo "== prof $n starts, waves=8" >> $O
    REPS=2 TTSWEEP_LIB=gpurun_exp/${b}prof.so timeout -k 10 200 python tools/exp/r5_sweep.py $n waves=8 2>&1 | grep -E "^prof" | tail -1 >> $O
  done
done
cat $O
