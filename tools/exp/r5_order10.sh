#!/bin/bash
# the model-aware default (-1) against 111 and 115: tests first, then geometries and start depths
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_order10.txt
: > $out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "tile or column" 2>&1 | tail -2 >> $out || { cat $out; exit 1; }
for g in "1024,1024,512 14" "1024,1024,256 14" "512,512,512 14" "241,241,51 24" "768,768,768 6" "512,512,256 8"; do
ORDERS=-1,111,115 timeout -k 10 500 python tools/exp/col_probe.py $g 2 1 2>&1 | grep -E "^mode 1 order|digests" | awk 'NR%2==0 || /digests/' >> $out
done
for k in 0.5 0.0; do
for g in "1024,1024,512 14" "512,512,512 14"; do
echo "== starts at k = $k (nz - 1), $g" >> $out
START_K=$k ORDERS=-1,111,115 timeout -k 10 500 python tools/exp/col_probe.py $g 2 1 2>&1 | grep -E "^mode 1 order|digests" | awk 'NR%2==0 || /digests/' >> $out
done; done
cut -c1-150 $out
