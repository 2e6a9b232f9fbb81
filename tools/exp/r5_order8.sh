#!/bin/bash
# table 5 (z flips with every sweep: each lateral quadrant down, then up) against table 1 from the nearest corner, more geometries
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_order8.txt
: > $out
for g in "640,1024,384 10" "1024,1024,256 14" "512,512,256 8" "241,241,51 24" "1024,512,512 14" "768,768,768 6" "1024,1024,512 7" "1024,1024,512 28"; do
ORDERS=111,115,15,215,105,111 timeout -k 10 500 python tools/exp/col_probe.py $g 2 1 2>&1 | grep -E "^mode 1 order|digests" | awk 'NR%2==0 || /digests/' >> $out
done
cut -c1-175 $out
