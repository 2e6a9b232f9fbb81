#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2t; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile or golden_cases" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
show() { python - <<PY
import json; d=json.load(open("$1")); r=d["roofline"]; h=d["roofline_hbm"]
print("$1", "ms", round(d["ms_per_step"],2), "hbm_frac", round(h["frac"],4), "launches", r["launches"], "avg_ms", round(r["avg_launch_ms"],4), "passes", round(d["config"]["passes_per_start_mean"],1), "eq", round(d["config"]["full_sweep_equivalents_per_start_mean"],2))
PY
}
python bench.py --no-cpu --no-traffic --no-host --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1 > $O/six512.json 2>$O/six512.err && show $O/six512.json
python bench.py --no-cpu --no-traffic --no-host --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 1 > $O/six1024.json 2>$O/six1024.err && show $O/six1024.json
python bench.py --no-cpu --no-traffic --no-host --star six --steps 3 --warmup 1 > $O/six241.json 2>$O/six241.err && show $O/six241.json
python tools/exp/one_sweep.py 1024,1024,512 14 2>&1 | tail -1
TTSWEEP_EXPERIMENT_LIB=gpurun_exp/tileprof.so python bench.py --no-cpu --no-traffic --no-host --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 0 > $O/p.json 2>$O/p.err; grep 'tile prof' $O/p.err | tail -1
