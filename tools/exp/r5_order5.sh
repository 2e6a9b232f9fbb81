#!/bin/bash
# after the default sequence of orderings changed (29): tile / column tests, the random soak with random sequences, profile, bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_order5.txt; : > $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "tile or column" 2>&1 | tail -3 >> $O || exit 1
timeout -k 10 500 python tools/exp/stress_tile.py 21 150 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
timeout -k 10 300 python tools/exp/col_repro.py 200 1 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
TTSWEEP_LIB=gpurun_exp/colprof.so timeout -k 10 300 python tools/exp/col_probe.py 1024,1024,512 14 1 1 2>&1 | grep -E "column prof|mode 1 order" | head -4 | cut -c1-1000 >> $O
timeout -k 10 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/r5_order5_bench.json
cat $O
