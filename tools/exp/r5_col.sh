#!/bin/bash
# column kernel: acquire only before staging, address table only for columns that run: TILE parity, then 1024x1024x512 x 14 A/B (colbase.so = before)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_col.txt; : > $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile or column" 2>&1 | tail -5 >> $O || { cat $O; exit 1; }
for lib in gpurun_exp/colbase.so "" gpurun_exp/colbase.so ""; do
  echo "=== lib '$lib'" >> $O
  TTSWEEP_LIB=$lib timeout -k 10 300 python tools/exp/col_probe.py 1024,1024,512 14 3 1 2>&1 | grep "mode 1 (" >> $O
done
echo "=== profile build" >> $O
TTSWEEP_LIB=gpurun_exp/colprof.so timeout -k 10 300 python tools/exp/col_probe.py 1024,1024,512 14 2 1 2>&1 | grep -E "column prof|mode 1 \(" | tail -3 >> $O
cat $O
