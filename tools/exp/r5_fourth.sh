#!/bin/bash
# the latency instance as two teams of four waves (half strips): parity (schedule matrix), sweep, phase stamps; A/B one workgroup per CU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_fourth.txt; : > $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule" 2>&1 | tail -5 >> $O || { cat $O; exit 1; }
timeout -k 10 600 python tools/exp/r5_sweep.py 1,3,8,24 - waves=8 waves=8,inunit=2 waves=8,inunit=0 waves=8,handoff=3 2>&1 | grep -v amdgpu.ids >> $O
echo "== one workgroup per CU (wgs1.so)" >> $O
TTSWEEP_LIB=gpurun_exp/wgs1.so timeout -k 10 600 python tools/exp/r5_sweep.py 1,3 waves=8 2>&1 | grep -v amdgpu.ids >> $O
for n in 1 3 24; do
  for cfg in - waves=8; do
    echo "== prof $n starts, $cfg" >> $O
    REPS=2 TTSWEEP_LIB=gpurun_exp/stripprof.so timeout -k 10 200 python tools/exp/r5_sweep.py $n $cfg 2>&1 | grep -E "^prof" | tail -1 >> $O
  done
done
cat $O
