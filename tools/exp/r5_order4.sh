#!/bin/bash
# per-start sequences: roles by how central the start is along each axis (63 +, 84 +), x-fastest tables
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_order4.txt
: > $out
# 8: x-fast cyclic, nearest corner; 29 / 50: + central lateral axis as x / as y; 71 / 92: all three by centrality; 7: reflected x-fast nearest corner; 70, 72..: z-fast tables with roles by centrality
for g in "1024,1024,512 14" "512,512,512 14" "768,512,256 20" "640,1024,384 10"; do
ORDERS=0,8,29,50,71,92,7,28,30,72,93,74,95 timeout -k 10 400 python tools/exp/col_probe.py $g 2 1 2>&1 | grep -E "^mode 1 order|digests" | awk 'NR%2==0 || /digests/' >> $out
done
cut -c1-175 $out
