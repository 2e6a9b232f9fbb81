#!/bin/bash
# the small shards (1 and 3 starts, eight-wave instance) against the ring's fill marks and the gate, after the big-grid finding
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_small.txt; : > $out
REPS=7 timeout -k 10 600 python tools/exp/r5_sweep.py 1,3 - low=4,high=16 low=8,high=32 low=16,high=64 low=64,high=256 low=256,high=1024 gate=350 gate=750 fast=1500 fast=3000 margin=250 margin=500 special=8 special=128 inunit=1 queues=4 queues=2 2>&1 | grep -v amdgpu.ids >> $out
cat $out
