#!/bin/bash
# 818-FS on 512x512x256 x 8 starts (one launch per solve): gate per round x deferral margin
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/g512k; mkdir -p $O
B="python bench.py --no-cpu --no-host --no-traffic --no-hbm-regime --grid 512,512,256 --starts 111 --nstarts 8 --steps 1 --warmup 1"
for cfg in "$@"; do
  g=${cfg%%:*}; m=${cfg#*:}
  timeout -k 10 300 $B --async-gate $g --defer-margin $m > $O/${g}_$m.json 2> $O/${g}_$m.err || { echo "$cfg FAILED"; tail -3 $O/${g}_$m.err; exit 1; }
  python3 - "$O/${g}_$m.json" "gate $g margin $m" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2].ljust(24), "ms %.2f"%d["ms_per_step"], "frac %.3f"%r["frac"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"], flush=True)
PY
done
