#!/bin/bash
# phase stamps of the unit kernel (-DTTSWEEP_PROFILE build) for 24, 3 and 1 starts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime --lib gpurun_exp/stripprof.so --steps 2 --warmup 1"
for n in 24 3 1; do
  echo "== $n starts"
  timeout -k 10 200 $B --nstarts $n > gpurun_out/stripprof_$n.json 2> gpurun_out/stripprof_$n.err; echo "rc $?"
  grep "^prof" gpurun_out/stripprof_$n.err | tail -2
  python3 -c "import json;d=json.loads(open('gpurun_out/stripprof_$n.json').read().strip().splitlines()[-1]);print('ms',d['ms_per_step'])"
done
