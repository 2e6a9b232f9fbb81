#!/bin/bash
# units of one or two planes by the number of starts, with the in-unit passes (pair_min_starts: 0 = always two, 1048576 = always one)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_pair.txt; : > $O
for n in 4 6 8 12 16; do
echo "== $n starts (cfg = async:pair_min_starts:low:high:special:policy:gate_milli:margin_milli)" >> $O
python tools/exp/async_sweep.py $n 1:0:0:0:0:1:-1:375 1:1048576:0:0:0:1:-1:375 2>&1 | grep solve >> $O
done
cat $O
