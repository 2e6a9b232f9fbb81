#!/bin/bash
# A/B of builds of the column kernel: bash tools/exp/r4_ab.sh NAME [NAME ...]   (gpurun_exp/NAME.so; "default")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for name in "$@"; do
  lib=""; [ "$name" != default ] && lib=gpurun_exp/$name.so
  echo "== $name"
  TTSWEEP_LIB=$lib timeout -k 10 300 python tools/exp/col_probe.py 1024,1024,512 14 3 1 2>&1 | grep "mode 1 (" | tail -2
done
