#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_prof.log; : > $L
TTSWEEP_LIB=gpurun_exp/colprof.so timeout -k 10 300 python tools/exp/col_probe.py 512,512,256 8 2 1 >> $L 2>&1; echo "rc $?" >> $L
TTSWEEP_LIB=gpurun_exp/colprof.so timeout -k 10 500 python tools/exp/col_probe.py 1024,1024,512 14 2 1 >> $L 2>&1; echo "rc $?" >> $L
cat $L
