#!/bin/bash
# (experiment of round 4; the code it switched on was measured, recorded under profiles/ and REMOVED: see HISTORY.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_vfilter.txt; : > $O
for n in 24 8 3 1; do
for f in 0 1; do
echo "== $n starts, value filter $f" >> $O
TTSWEEP_VALUE_FILTER=$f python tools/exp/async_sweep.py $n 1:-1:0:0:0:1:-1:500 1:-1:0:0:0:1:-1:-1000000000 1:-1:0:0:0:1:-1:2000 2>&1 | grep solve >> $O
done; done
cat $O
