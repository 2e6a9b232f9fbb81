#!/bin/bash
# the latency instance with plane groups: parity (schedule matrix), sweep, phase stamps; A/B build G = 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_third.txt; : > $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule" 2>&1 | tail -5 >> $O || { cat $O; exit 1; }
timeout -k 10 600 python tools/exp/r5_sweep.py 1,3,8 - waves=8 waves=8,inunit=2 waves=8,inunit=0 waves=8,fast=3000 2>&1 | grep -v amdgpu.ids >> $O
echo "== G = 2 build" >> $O
TTSWEEP_LIB=gpurun_exp/g2.so timeout -k 10 600 python tools/exp/r5_sweep.py 1,3 waves=8 2>&1 | grep -v amdgpu.ids >> $O
echo "== default headline (24 starts)" >> $O
timeout -k 10 600 python tools/exp/r5_sweep.py 24 - 2>&1 | grep -v amdgpu.ids >> $O
for n in 1 3; do
  for cfg in - waves=8; do
    echo "== prof $n starts, $cfg" >> $O
    REPS=2 TTSWEEP_LIB=gpurun_exp/stripprof.so timeout -k 10 200 python tools/exp/r5_sweep.py $n $cfg 2>&1 | grep -E "^prof" | tail -1 >> $O
  done
done
cat $O
