#!/bin/bash
# 818-FS 512x512x256 x 8 with the fuller rings: ring policy, gate speeds, margin once more
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_big818d.txt; : > $out
GRID=512,512,256 REPS=3 timeout -k 10 900 python tools/exp/r5_sweep.py 8 - policy=0 gate=250 gate=350 gate=750 gate=1000 fast=500 fast=1000 fast=3000 gate=350,fast=1000 gate=750,fast=1000 margin=250 margin=500 inunit=3 inunit=4 special=128 - 2>&1 | grep -v amdgpu.ids >> $out
cat $out
