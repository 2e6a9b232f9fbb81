#!/bin/bash
# A/B of builds on the TILE workloads: bash tools/exp/abtile.sh NAME [NAME ...]  ("default" = csrc/libttsweep.so)
cd "${GRAFT_REPO_ROOT:-.}"
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime --star six"
for name in "$@"; do
  lib=""; [ "$name" != default ] && lib="--lib gpurun_exp/$name.so"
  for cfg in "g1024:--grid 1024,1024,512 --starts 111 --nstarts 14 --steps 2 --warmup 1" "g512:--grid 512,512,256 --starts 111 --nstarts 8 --steps 3 --warmup 1" "g241:--steps 5 --warmup 1"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    $B $args $lib 2>/dev/null | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('${name}_$tag'.ljust(22), 'ms %.2f'%d['ms_per_step'], r['bound'], 'frac %.3f'%r['frac'], 'launch_ms %.4f'%r['avg_launch_ms'], 'n', r['launches'])
except Exception as e: print('${name}_$tag FAILED', e)"
  done
done
