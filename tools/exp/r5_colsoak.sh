#!/bin/bash
# soak of the column kernel after the round-5 protocol changes: random grids / stars / starts against the CELL kernel, and the
# three-start case that once caught a double rest declaration, over and over, in both layouts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_colsoak.txt; : > $O
timeout -k 10 500 python tools/exp/stress_tile.py 11 150 2>&1 | grep -v amdgpu.ids | tail -4 >> $O
timeout -k 10 500 python tools/exp/stress_tile.py 12 150 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
timeout -k 10 300 python tools/exp/col_repro.py 300 1 2>&1 | grep -v amdgpu.ids | tail -3 >> $O
timeout -k 10 300 python tools/exp/col_repro.py 300 0 2>&1 | grep -v amdgpu.ids | tail -3 >> $O
cat $O
