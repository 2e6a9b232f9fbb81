#!/bin/bash
# full GPU tier with the round-5 code, then the small-shard lines and the dead-edge interval
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_sixth.txt; : > $O
timeout -k 10 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 >> $O || { cat $O; exit 1; }
timeout -k 10 600 python tools/exp/r5_sweep.py 1,2,3,4,5,6 - waves=4 waves=8 2>&1 | grep -v amdgpu.ids >> $O
timeout -k 10 600 python tools/exp/r5_sweep.py 24,3 - special=1 special=8 special=32 special=512 2>&1 | grep -v amdgpu.ids >> $O
cat $O
