#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2h; mkdir -p $O
export TTSWEEP_EXPERIMENT_LIB=gpurun_exp/tileprof.so
for cfg in "241,241,51 24 4" "241,241,51 24 24" "512,512,256 111 1" "512,512,256 111 8" "1024,1024,512 111 1" "1024,1024,512 111 4"; do
  set -- $cfg
  python bench.py --no-cpu --no-traffic --no-host --star six --grid $1 --starts $2 --nstarts $3 --steps 1 --warmup 0 > $O/p.json 2>$O/p.err
  echo "grid $1 nstarts $3: $(grep 'tile prof' $O/p.err | tail -1)"
  python - <<PY
import json; d=json.load(open("$O/p.json")); h=d["roofline_hbm"]; print("   ms", round(d["ms_per_step"],2), "hbm_frac", round(h["frac"],4), "launches", h["launches"], "avg_ms", round(h["avg_launch_ms"],4))
PY
done
