#!/bin/bash
# in-unit passes (TTSWEEP_OPT_ASYNC_INUNIT: a unit that improved is relaxed again against its own planes): solve times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r4_inunit.txt; : > $O
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime --steps 4 --warmup 1"
run() {   # name, args...
  name=$1; shift
  for k in 0 1 2 3; do
    timeout -k 10 300 $B "$@" --inunit $k > gpurun_out/inunit_tmp.json 2> gpurun_out/inunit_tmp.err || echo "FAILED $name $k" >> $O
    python3 -c "import json;d=json.loads(open('gpurun_out/inunit_tmp.json').read().strip().splitlines()[-1]);print('$name in-unit passes',$k,'ms %.2f'%d['ms_per_step'],'sweep-eq/start %.2f'%d['config']['full_sweep_equivalents_per_start_mean'],'frac %.3f'%d['roofline']['frac'],'fallbacks',d['config']['fallbacks'])" >> $O
  done
}
run "241x241x51 x 24" 
run "241x241x51 x 8" --nstarts 8
run "241x241x51 x 4 (start-4)" --starts 4
run "241x241x51 x 3" --nstarts 3
run "241x241x51 x 2" --nstarts 2
run "241x241x51 x 1" --nstarts 1
run "512x512x256 x 8" --grid 512,512,256 --starts 111 --nstarts 8 --steps 2
cat $O
