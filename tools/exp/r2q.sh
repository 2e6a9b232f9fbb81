#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2q; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not 1024 and not 512 and not full_size and not above_2" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
show() { python - <<PY
import json; d=json.load(open("$1")); r=d["roofline"]
print("$1", "ms", round(d["ms_per_step"],2), "frac", round(r["frac"],4), "avg_ms", round(r["avg_launch_ms"],4), "passes", round(d["config"]["passes_per_start_mean"],1), "eq", round(d["config"]["full_sweep_equivalents_per_start_mean"],2))
PY
}
for n in 3 8; do for pm in 1000 0; do
python bench.py --no-cpu --no-traffic --no-host --steps 8 --warmup 2 --nstarts $n --pair-min-starts $pm > $O/b${n}_$pm.json 2>$O/err && show $O/b${n}_$pm.json
done; done
python bench.py --no-cpu --no-traffic --no-host --steps 5 --warmup 2 > $O/b24.json 2>$O/err && show $O/b24.json
for gs in 3 4; do python bench.py --no-cpu --no-traffic --no-host --steps 5 --warmup 2 --gate-speed $gs > $O/b24_g$gs.json 2>$O/err && show $O/b24_g$gs.json; done
