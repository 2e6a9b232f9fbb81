#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_big.log; : > $L
timeout -k 10 300 python tools/exp/col_probe.py 512,512,256 8 2 >> $L 2>&1; echo "probe512x8 rc $?" >> $L
timeout -k 10 500 python tools/exp/col_probe.py 1024,1024,512 14 2 >> $L 2>&1; echo "probe1024x14 rc $?" >> $L
cat $L
