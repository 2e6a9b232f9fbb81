#!/bin/bash
# 818-FS on a grid that leaves the caches (512x512x256 x 8): the schedule knobs, tuned on 241x241x51 so far
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_big818.txt; : > $out
GRID=512,512,256 REPS=2 timeout -k 10 900 python tools/exp/r5_sweep.py 8 - gate=250 gate=1000 gate=2000 fast=500 fast=1000 fast=4000 gate=250,fast=500 gate=250,fast=1000 margin=0 margin=1000 margin=2000 inunit=0 inunit=1 inunit=4 low=16,high=64 low=128,high=512 pair=0 pair=1048576 special=8 special=256 2>&1 | grep -v amdgpu.ids >> $out
cat $out
