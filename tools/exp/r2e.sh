#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2e; mkdir -p $O
show() { python - <<PY
import json; d=json.load(open("$1")); r=d["roofline"]; h=d["roofline_hbm"]
print("$1", "ms", round(d["ms_per_step"],2), "bound", r["bound"], "frac", round(r["frac"],4), "hbm_frac", round(h["frac"],4), "launches", r["launches"], "avg_ms", round(r["avg_launch_ms"],4), "passes", round(d["config"]["passes_per_start_mean"],1), "eq", round(d["config"]["full_sweep_equivalents_per_start_mean"],2), "traffic", r["traffic"], "alg", r["algorithmic_bytes_per_launch"])
PY
}
python bench.py --no-cpu --no-traffic --no-host --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1 > $O/six512.json 2>$O/six512.err; show $O/six512.json
python bench.py --no-cpu --no-traffic --no-host --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 1 > $O/six1024.json 2>$O/six1024.err; show $O/six1024.json
python bench.py --no-cpu --no-traffic --no-host --star six --steps 3 --warmup 1 > $O/six241.json 2>$O/six241.err; show $O/six241.json
python bench.py --no-cpu --no-host --star six --grid 1024,1024,512 --starts 111 --nstarts 4 --steps 1 --warmup 1 > $O/six1024t.json 2>$O/six1024t.err; show $O/six1024t.json
( time python bench.py ) > $O/default.json 2>$O/default.err; tail -3 $O/default.err; show $O/default.json; cat $O/default.json | python -c "import json,sys; d=json.load(sys.stdin); print({k:d[k] for k in d if k not in ('roofline','roofline_hbm','config')})"
