#!/bin/bash
# first GPU pass of round 5: the STRIP parity tests with the new ring protocol + hand-off, then the small-shard sweep
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_first.txt; : > $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule or golden_cases or seeded or full_size or gives_up or starts_per_ring or solve_multi or per_start" 2>&1 | tail -15 >> $O || { cat $O; exit 1; }
timeout -k 10 600 python tools/exp/r5_sweep.py 1,3,24 - handoff=0 handoff=1 handoff=3 handoff=3,fast=4000 handoff=3,gate=1000,fast=4000 handoff=3,inunit=2 2>&1 | grep -v amdgpu.ids >> $O
cat $O
