#!/bin/bash
# sequences whose second eight sweeps flip z fastest (tables 7 - 9), against the default (111), on four geometries
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_order6.txt
: > $out
for g in "1024,1024,512 14" "512,512,512 14" "768,512,256 20" "640,1024,384 10"; do
ORDERS=111,117,118,119,111 timeout -k 10 400 python tools/exp/col_probe.py $g 2 1 2>&1 | grep -E "^mode 1 order|digests" | awk 'NR%2==0 || /digests/' >> $out
done
cut -c1-175 $out
