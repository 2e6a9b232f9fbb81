#!/bin/bash
# The other BASELINE configurations (818-FS) and the small shards, one launch per solve against a launch pair per pass.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/oc2; mkdir -p $O
B="python bench.py --no-cpu --no-host --no-traffic --no-hbm-regime"
for mode in 1 0; do
  for cfg in "n24:--steps 5 --warmup 1" "n3:--steps 5 --warmup 2 --nstarts 3" "n1:--steps 5 --warmup 2 --nstarts 1" "start4:--starts 4 --steps 5 --warmup 2" \
             "g512:--grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1" "g1024:--grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 1"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    timeout -k 10 400 $B $args --async-mode $mode "$@" > $O/${tag}_m$mode.json 2> $O/${tag}_m$mode.err || { echo "$tag mode $mode FAILED"; tail -3 $O/${tag}_m$mode.err; exit 1; }
    python3 - "$O/${tag}_m$mode.json" "${tag}_m$mode" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2].ljust(12), "ms %.2f"%d["ms_per_step"], "frac %.3f"%r["frac"], "launch_ms %.4f"%r["avg_launch_ms"], "n", r["launches"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"], flush=True)
PY
  done
done
