#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2v; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
show() { python - <<PY
import json; d=json.load(open("$1")); r=d["roofline"]
print("$1", "ms", round(d["ms_per_step"],2), "frac", round(r["frac"],4), "avg_ms", round(r["avg_launch_ms"],4), "passes", round(d["config"]["passes_per_start_mean"],1), "eq", round(d["config"]["full_sweep_equivalents_per_start_mean"],2))
PY
}
for n in 1 3; do
python bench.py --no-cpu --no-traffic --no-host --steps 10 --warmup 2 --nstarts $n > $O/b$n.json 2>$O/err && show $O/b$n.json
done
python bench.py --no-cpu --no-traffic --no-host --steps 5 --warmup 2 > $O/b24.json 2>$O/err && show $O/b24.json
