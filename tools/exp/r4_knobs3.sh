#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_knobs3.txt; : > $O
for n in 24 8 3 1; do
echo "== $n starts (cfg = async:pair:low:high:special:policy:gate_milli:margin_milli[:fast_gate_milli])" >> $O
python tools/exp/async_sweep.py $n 1:-1:0:0:0:1:-1:500 1:-1:0:0:0:1:-1:375 1:-1:0:0:0:1:-1:750 1:-1:0:0:0:1:-1:1000 1:-1:0:0:0:1:600:500 1:-1:0:0:0:1:900:500 >> $O 2>&1
done
grep -v amdgpu.ids $O
