#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2l; mkdir -p $O
show() { python - <<PY
import json; d=json.load(open("$1")); r=d["roofline"]
print("$1", "ms", round(d["ms_per_step"],2), "frac", round(r["frac"],4), "avg_ms", round(r["avg_launch_ms"],4), "eq", round(d["config"]["full_sweep_equivalents_per_start_mean"],2))
PY
}
for rep in 1 2; do
for lib in uoparallel-seismic-project_amd/csrc/libttsweep.so gpurun_exp/dma2.so; do
export TTSWEEP_EXPERIMENT_LIB=$lib
n=$(basename $lib .so)
python bench.py --no-cpu --no-traffic --no-host --steps 5 --warmup 2 > $O/b_$n.json 2>$O/err && show $O/b_$n.json
done; done
export TTSWEEP_EXPERIMENT_LIB=gpurun_exp/prof.so
python bench.py --no-cpu --no-traffic --no-host --steps 1 --warmup 0 > $O/prof.json 2>$O/prof.err; grep "^prof" $O/prof.err | tail -2
