#!/bin/bash
# the three-start case that caught the double rest declaration, many times over, both column drivers
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_repro.log; : > $L
timeout -k 10 300 python tools/exp/col_repro.py 300 1 > gpurun_out/r4_repro_plain.log 2>&1; echo "in-place rc $?" >> $L
tail -3 gpurun_out/r4_repro_plain.log >> $L
timeout -k 10 300 python tools/exp/col_repro.py 300 0 > gpurun_out/r4_repro_padded.log 2>&1; echo "padded rc $?" >> $L
tail -3 gpurun_out/r4_repro_padded.log >> $L
grep -v amdgpu.ids $L | tail -50
