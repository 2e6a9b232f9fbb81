#!/bin/bash
# tables 1 and 5 from the nearest corner when the starts do NOT lie on the face k = nz - 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_order9.txt
: > $out
for k in 0.5 0.25 0.0; do
for g in "1024,1024,512 14" "1024,1024,256 14" "512,512,512 14"; do
echo "== starts at k = $k (nz - 1), $g" >> $out
START_K=$k ORDERS=111,115,112 timeout -k 10 500 python tools/exp/col_probe.py $g 2 1 2>&1 | grep -E "^mode 1 order|digests" | awk 'NR%2==0 || /digests/' >> $out
done; done
cut -c1-175 $out
