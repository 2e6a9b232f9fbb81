#!/bin/bash
# the latency instance: parity (schedule matrix, golden cases) + small-shard sweep
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_second.txt; : > $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule" 2>&1 | tail -15 >> $O || { cat $O; exit 1; }
timeout -k 10 600 python tools/exp/r5_sweep.py 1,3 - waves=8 waves=8,inunit=1 waves=8,inunit=2 waves=8,handoff=3 waves=8,handoff=1 waves=8,fast=3000 waves=8,gate=750,fast=3000 waves=8,margin=1000 2>&1 | grep -v amdgpu.ids >> $O
timeout -k 10 300 python tools/exp/r5_sweep.py 8 - waves=8 waves=8,handoff=3 handoff=3 2>&1 | grep -v amdgpu.ids >> $O
cat $O
