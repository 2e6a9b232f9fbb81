#!/bin/bash
# schedule knobs of the one-launch STRIP solve again, with the in-unit passes (24 starts)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_knobs2.txt; : > $O
echo "== 24 starts (cfg = async:pair:low:high:special:policy:gate_milli:margin_milli[:fast_gate_milli])" >> $O
python tools/exp/async_sweep.py 24 1:-1:0:0:0:1:-1:1000 1:-1:0:0:0:1:-1:250 1:-1:0:0:0:1:-1:500 1:-1:0:0:0:1:-1:2000 1:-1:0:0:0:1:-1:4000 \
   1:-1:0:0:0:1:500:1000 1:-1:0:0:0:1:750:1000 1:-1:0:0:0:1:1000:1000 1:-1:0:0:0:1:1500:1000 1:-1:0:0:0:1:2000:1000 1:-1:0:0:0:1:3000:1000 \
   1:0:0:0:0:1:-1:1000 1:1000000:0:0:0:1:-1:1000 >> $O 2>&1
echo "== 3 starts" >> $O
python tools/exp/async_sweep.py 3 1:-1:0:0:0:1:-1:1000 1:-1:0:0:0:1:-1:500 1:-1:0:0:0:1:500:1000 1:-1:0:0:0:1:1500:1000 >> $O 2>&1
grep -v amdgpu.ids $O
