#!/bin/bash
# unit order: shells of N cells ordered by position (TTSWEEP_OPT_UNIT_ORDER_SHELL_MILLI) on the grids that leave the caches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_shell.txt; : > $O
B="python bench.py --no-cpu --no-host --no-traffic --no-hbm-regime"
for sh in 0 4 8 16 32; do
  timeout -k 10 400 $B --grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1 --order-shell $sh > gpurun_out/shell_$sh.json 2> gpurun_out/shell_$sh.err || { echo "shell $sh FAILED" >> $O; tail -3 gpurun_out/shell_$sh.err >> $O; }
  python3 - gpurun_out/shell_$sh.json "512x512x256 x 8, shell $sh" >> $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2].ljust(30), "ms %.2f"%d["ms_per_step"], "frac %.3f"%r["frac"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"], "fallbacks", d["config"]["fallbacks"], flush=True)
PY
done
for sh in 0 8; do
  timeout -k 10 400 $B --steps 3 --warmup 1 --order-shell $sh > gpurun_out/shell24_$sh.json 2> gpurun_out/shell24_$sh.err
  python3 - gpurun_out/shell24_$sh.json "241x241x51 x 24, shell $sh" >> $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2].ljust(30), "ms %.2f"%d["ms_per_step"], "frac %.3f"%r["frac"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"], "fallbacks", d["config"]["fallbacks"], flush=True)
PY
done
cat $O
