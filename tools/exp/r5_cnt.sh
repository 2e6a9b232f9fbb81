#!/bin/bash
# arrival counters instead of a barrier per staged plane (four-wave one-launch instances): parity, headline, 512 grid; A/B cnt0.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_cnt.txt; : > $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule or golden_cases or full_size or seeded" 2>&1 | tail -5 >> $O || { cat $O; exit 1; }
for lib in "" gpurun_exp/cnt0.so; do
  echo "=== lib '$lib'" >> $O
  TTSWEEP_LIB=$lib timeout -k 10 600 python tools/exp/r5_sweep.py 24,8,4 - 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== prof 24 starts" >> $O
REPS=2 TTSWEEP_LIB=gpurun_exp/stripprof.so timeout -k 10 200 python tools/exp/r5_sweep.py 24 - 2>&1 | grep -E "^prof" | tail -1 >> $O
B="python bench.py --no-cpu --no-host --no-traffic --no-hbm-regime"
for lib in "" "--lib gpurun_exp/cnt0.so"; do
  timeout -k 10 400 $B --grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1 $lib > gpurun_out/cnt512.json 2> gpurun_out/cnt512.err
  python3 - gpurun_out/cnt512.json "512x512x256 x 8 $lib" >> $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2].ljust(40), "ms %.2f"%d["ms_per_step"], "frac %.3f"%r["frac"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"], "fallbacks", d["config"]["fallbacks"], flush=True)
PY
done
cat $O
