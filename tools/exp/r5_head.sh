#!/bin/bash
# the headline (24 starts) once more against the ring's fill marks (smaller ones) and neighbours of the defaults
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_head.txt; : > $out
REPS=9 timeout -k 10 600 python tools/exp/r5_sweep.py 24 - low=8,high=32 low=16,high=64 low=24,high=96 low=31,high=80 low=48,high=126 low=16,high=126 fast=1000 fast=1500 fast=3000 gate=400 gate=600 margin=300 margin=450 inunit=3 inunit=1 - 2>&1 | grep -v amdgpu.ids >> $out
cat $out
