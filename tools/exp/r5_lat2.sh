#!/bin/bash
# the eight-wave instance with TWO-plane units (pair=0 = pairs from the first start on, waves=8) against the defaults
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_lat2.txt; : > $out
REPS=7 timeout -k 10 600 python tools/exp/r5_sweep.py 1,2,3,4,6,8 - pair=0,waves=8 pair=0,waves=8,inunit=1 pair=0,waves=8,inunit=2 pair=0,waves=4 2>&1 | grep -v amdgpu.ids >> $out
cat $out
