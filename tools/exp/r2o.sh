#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2o; mkdir -p $O
show() { python - <<PY
import json; d=json.load(open("$1")); r=d["roofline"]
print("$1", "ms", round(d["ms_per_step"],2), "frac", round(r["frac"],4), "avg_ms", round(r["avg_launch_ms"],4), "passes", round(d["config"]["passes_per_start_mean"],1), "eq", round(d["config"]["full_sweep_equivalents_per_start_mean"],2))
PY
}
for gs in 2.5 3 3.5 4 4.5; do
python bench.py --no-cpu --no-traffic --no-host --steps 5 --warmup 2 --gate-speed $gs > $O/b24_$gs.json 2>$O/err && show $O/b24_$gs.json
done
for gs in 3.5 5; do
python bench.py --no-cpu --no-traffic --no-host --steps 10 --warmup 2 --nstarts 3 --gate-speed $gs > $O/b3_$gs.json 2>$O/err && show $O/b3_$gs.json
done
