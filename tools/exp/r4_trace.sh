#!/bin/bash
# kernel trace + stats of one six-FS 1024x1024x512 x 14 solve (column driver)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/trace_col; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/exp/col_probe.py 1024,1024,512 14 2 1 > $out.log 2>&1
echo rc $?
f=$(find $out -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-220
