#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_knobs5.txt; : > $O
for n in 24 8 3 1; do
echo "== $n starts (cfg = async:pair:low:high:special:policy:gate_milli:margin_milli)" >> $O
python tools/exp/async_sweep.py $n 1:-1:0:0:0:1:-1:500 1:-1:0:0:0:1:500:250 1:-1:0:0:0:1:500:375 1:-1:0:0:0:1:600:250 1:-1:0:0:0:1:-1:250 1:-1:0:0:0:1:500:0 1:-1:0:0:0:1:400:250 2>&1 | grep solve >> $O
done
cat $O
bash tools/exp/g512_knobs.sh 0.5:0.25 0.4:0.25 0.5:0.0 0.5:0.125 0.6:0.25
