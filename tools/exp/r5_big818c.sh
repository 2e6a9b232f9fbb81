#!/bin/bash
# the new default fill marks on the big grids, and the mark ratio
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_big818c.txt; : > $out
GRID=512,512,256 REPS=3 timeout -k 10 600 python tools/exp/r5_sweep.py 8 - low=512,high=4096 low=2048,high=4096 low=3072,high=4096 low=2048,high=8000 low=1024,high=4096,fast=500 low=1024,high=4096,margin=250 low=1024,high=4096,margin=500 low=1024,high=4096,gate=750 2>&1 | grep -v amdgpu.ids >> $out
GRID=1024,1024,512 REPS=1 timeout -k 10 600 python tools/exp/r5_sweep.py 14 - low=31,high=126 2>&1 | grep -v amdgpu.ids >> $out
REPS=5 timeout -k 10 300 python tools/exp/r5_sweep.py 24,3 - 2>&1 | grep -v amdgpu.ids >> $out
cat $out
