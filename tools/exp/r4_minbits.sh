#!/bin/bash
# (experiment of round 4; the code it switched on was measured, recorded under profiles/ and REMOVED: see HISTORY.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_minbits.txt; : > $O
for n in 24 8 3; do
for m in 1 2 3 4 6; do
echo "== $n starts, minbits $m" >> $O
TTSWEEP_LIB=gpurun_exp/minbits.so TTSWEEP_MINBITS=$m python tools/exp/async_sweep.py $n 1:-1:0:0:0:1:-1:500 2>&1 | grep solve >> $O
done; done
cat $O
