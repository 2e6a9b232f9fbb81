#!/bin/bash
# (experiment of round 4; the code it switched on was measured, recorded under profiles/ and REMOVED: see HISTORY.md)
# split units (TTSWEEP_SPLIT, experiment): a unit's changed planes dealt over 1 .. 4 ring entries
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_split.txt; : > $O
for n in 1 3; do
for sp in 1 2 3; do
echo "== $n starts, split $sp" >> $O
TTSWEEP_SPLIT=$sp timeout -k 10 120 python tools/exp/async_sweep.py $n 1:-1:0:0:0:1:-1:375 1:-1:0:0:0:1:-1:375:-1:0 1:-1:0:0:0:1:1000:375:2000:0 2>&1 | grep "solve\|rror" >> $O
done; done
cat $O
