#!/bin/bash
# batched claims of the column kernel: tests, then time against the batch size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_batch.txt; : > $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "tile or column" 2>&1 | tail -5 >> $O || { cat $O; exit 1; }
for b in 1 2 4 8; do
BATCH=$b timeout -k 10 200 python tools/exp/col_probe.py 1024,1024,512 14 3 1 2>&1 | grep -E "^mode 1 order" | tail -1 | sed "s/^/batch $b: /" >> $O
done
for b in 1 8; do
BATCH=$b timeout -k 10 200 python tools/exp/col_probe.py 512,512,512 14 3 1 2>&1 | grep -E "^mode 1 order" | tail -1 | sed "s/^/batch $b: /" >> $O
done
cut -c1-250 $O
