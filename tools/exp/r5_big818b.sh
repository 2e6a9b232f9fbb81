#!/bin/bash
# 818-FS on grids that leave the caches: ring fill marks and the gate's speed while the ring runs dry
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_big818b.txt; : > $out
GRID=512,512,256 REPS=3 timeout -k 10 600 python tools/exp/r5_sweep.py 8 - low=64,high=256 low=128,high=512 low=256,high=1024 low=512,high=2048 low=1024,high=4096 low=128,high=512,fast=500 low=256,high=1024,fast=500 low=256,high=1024,fast=1000 low=256,high=1024,inunit=4 low=256,high=512 low=384,high=512 - 2>&1 | grep -v amdgpu.ids >> $out
GRID=1024,1024,512 REPS=1 timeout -k 10 600 python tools/exp/r5_sweep.py 14 - low=128,high=512 low=256,high=1024 low=256,high=1024,fast=500 2>&1 | grep -v amdgpu.ids >> $out
REPS=5 timeout -k 10 300 python tools/exp/r5_sweep.py 24 - low=64,high=256 low=128,high=512 low=256,high=1024 2>&1 | grep -v amdgpu.ids >> $out
cat $out
