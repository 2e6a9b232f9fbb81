#!/bin/bash
# A/B of builds of libttsweep.so: bash tools/exp/ab.sh NAME [NAME ...]  (gpurun_exp/NAME.so; "default" = csrc/libttsweep.so)
# parity smoke test of every build (tools/exp/check_lib.py), then the headline, the 3-start shard and the 512 grid.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/ab; mkdir -p $O
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime"
for name in "$@"; do
  lib=""; [ "$name" != default ] && lib="--lib gpurun_exp/$name.so"
  if [ "$name" != default ]; then timeout -k 10 200 python tools/exp/check_lib.py gpurun_exp/$name.so 2>&1 | tail -n 1; fi
  for cfg in "n24:--steps 5 --warmup 1" "n3:--steps 5 --warmup 1 --nstarts 3" "g512:--steps 2 --warmup 1 --grid 512,512,256 --starts 111 --nstarts 8"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    $B $args $lib > $O/${name}_$tag.json 2> $O/${name}_$tag.err
    python3 - "$O/${name}_$tag.json" "${name}_$tag" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2].ljust(22), "ms %.2f"%d["ms_per_step"], "frac %.3f"%r["frac"], "launch_ms %.4f"%r["avg_launch_ms"], "n", r["launches"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
  done
done
