#!/bin/bash
# the sequence of orderings of the column sweeps (TTSWEEP_OPT_TILE_ORDER): time, sweeps, work; digests across all
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_order.txt
: > $out
ORDERS=0,1,2,3,4,5,6 timeout -k 10 500 python tools/exp/col_probe.py 1024,1024,512 14 3 1 2>&1 | grep -E "^mode 1 order|digests" >> $out
ORDERS=0,2,5 timeout -k 10 300 python tools/exp/col_probe.py 512,512,512 14 3 1,0 2>&1 | grep -E "^mode . order|digests" >> $out
cat $out
