#!/bin/bash
# the rest of the matrix: tables with z flipping every sweep (5, 6) and z-fastest (2, 4) from the nearest corner, with the axis roles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_order7.txt
: > $out
for g in "1024,1024,512 14" "512,512,512 14" "768,512,256 20"; do
ORDERS=111,15,16,115,116,215,216,112,114,212,214,113,213 timeout -k 10 500 python tools/exp/col_probe.py $g 2 1 2>&1 | grep -E "^mode 1 order|digests" | awk 'NR%2==0 || /digests/' >> $out
done
cut -c1-175 $out
