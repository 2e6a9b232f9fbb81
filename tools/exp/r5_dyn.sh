#!/bin/bash
# the latency instance with a group's items drawn from an LDS counter: parity, timings, phase stamps; A/B dyn0.so (fixed shares)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_dyn.txt; : > $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule or golden_cases or full_size" 2>&1 | tail -5 >> $O || { cat $O; exit 1; }
for lib in "" gpurun_exp/dyn0.so; do
  echo "=== lib '$lib'" >> $O
  TTSWEEP_LIB=$lib timeout -k 10 600 python tools/exp/r5_sweep.py 1,2,3 - waves=8,inunit=2 2>&1 | grep -v amdgpu.ids >> $O
done
TTSWEEP_LIB= timeout -k 10 600 python tools/exp/r5_sweep.py 4,6,8 waves=8 waves=4 2>&1 | grep -v amdgpu.ids >> $O
for n in 1 3; do
    echo "== prof $n starts" >> $O
    REPS=2 TTSWEEP_LIB=gpurun_exp/stripprof.so timeout -k 10 200 python tools/exp/r5_sweep.py $n - 2>&1 | grep -E "^prof" | tail -1 >> $O
done
cat $O
