#!/bin/bash
# soaks after the second half of round 5: column kernel with random sequences of orderings, STRIP with the round-5 knobs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_soak2.txt; : > $O
timeout -k 10 500 python tools/exp/stress_tile.py 31 150 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
timeout -k 10 500 python tools/exp/stress_tile.py 32 150 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
timeout -k 10 300 python tools/exp/col_repro.py 300 1 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
timeout -k 10 300 python tools/exp/col_repro.py 300 0 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
timeout -k 10 600 python tools/exp/async_soak.py 400 7 2>&1 | grep -v amdgpu.ids | tail -4 >> $O
cat $O
