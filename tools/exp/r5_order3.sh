#!/bin/bash
# per-start sequences of orderings: nearest corner first (7 +), the more / less central lateral axis in the table's x role (21 +, 42 +)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_order3.txt
: > $out
ORDERS=0,9,11,30,32,51,53,10,8 timeout -k 10 400 python tools/exp/col_probe.py 1024,1024,512 14 2 1 2>&1 | grep -E "^mode 1 order|digests" | awk 'NR%2==0 || /digests/' >> $out
ORDERS=0,9,11,30,32,51,53 timeout -k 10 300 python tools/exp/col_probe.py 512,512,512 14 2 1 2>&1 | grep -E "^mode 1 order|digests" | awk 'NR%2==0 || /digests/' >> $out
ORDERS=0,9,11,30,32,51,53 timeout -k 10 300 python tools/exp/col_probe.py 768,512,256 20 2 1 2>&1 | grep -E "^mode 1 order|digests" | awk 'NR%2==0 || /digests/' >> $out
cut -c1-200 $out
