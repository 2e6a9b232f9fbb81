#!/bin/bash
# more sequences of orderings (raw, -DTTSWEEP_DEBUG_ENV build: TTSWEEP_COL_ORDSEQ) and the corner each start's first sweep begins at
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_order2.txt
: > $out
hex() { python3 -c "import sys; q=[int(x) for x in sys.argv[1].split(',')]; q=(q*16)[:16]; print('%x' % sum(o << (4*e) for e, o in enumerate(q)))" $1; }
for seq in 0,4,6,2,3,7,5,1 0,4,0,4,6,2,6,2,3,7,3,7,5,1,5,1 0,4,6,2,3,7,5,1,1,5,7,3,2,6,4,0 0,2,6,4,5,7,3,1 0,4,6,2,3,7,5,1,0,4,5,1,3,7,6,2 0,4,6,2,0,4,6,2,3,7,5,1,3,7,5,1 0,4,6,7,3,2,0,1,5,4,6,7,3,2,0,1 ; do
  echo "== $seq" >> $out
  TTSWEEP_LIB=gpurun_exp/coldbg.so TTSWEEP_COL_ORDSEQ=$(hex $seq) timeout -k 10 120 python tools/exp/col_probe.py 1024,1024,512 14 2 1 2>&1 | grep -E "^mode 1 order" | tail -1 >> $out
done
echo "== first corner: order 4, 11, 18; 2, 9, 16" >> $out
ORDERS=4,11,18,2,9,16 timeout -k 10 300 python tools/exp/col_probe.py 1024,1024,512 14 2 1 2>&1 | grep -E "^mode 1 order|digests" >> $out
cat $out
