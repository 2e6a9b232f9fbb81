#!/bin/bash
# small shards: units of two planes and more in-unit passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_knobs4.txt; : > $O
for n in 3 1; do
echo "== $n starts (cfg = async:pair:low:high:special:policy:gate_milli:margin_milli:fast_gate_milli:in-unit passes)" >> $O
python tools/exp/async_sweep.py $n 1:-1:0:0:0:1:-1:500 1:0:0:0:0:1:-1:500:-1:2 1:0:0:0:0:1:-1:500:-1:4 1:0:0:0:0:1:-1:500:-1:0 1:-1:0:0:0:1:-1:500:-1:4 1:-1:0:0:0:1:-1:500:-1:1 1:-1:0:0:0:1:-1:500:-1:0 >> $O 2>&1
done
grep -v amdgpu.ids $O
