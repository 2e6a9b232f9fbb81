#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2j; mkdir -p $O
show() { python - <<PY
import json; d=json.load(open("$1")); r=d["roofline"]; h=d["roofline_hbm"]
print("$1", "ms", round(d["ms_per_step"],2), "hbm_frac", round(h["frac"],4), "launches", r["launches"], "avg_ms", round(r["avg_launch_ms"],4), "passes", round(d["config"]["passes_per_start_mean"],1), "eq", round(d["config"]["full_sweep_equivalents_per_start_mean"],2))
PY
}
for lib in uoparallel-seismic-project_amd/csrc/libttsweep.so gpurun_exp/z64.so gpurun_exp/z16.so; do
export TTSWEEP_EXPERIMENT_LIB=$lib
n=$(basename $lib .so)
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_kernel_small" 2>&1 | tail -1
python bench.py --no-cpu --no-traffic --no-host --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1 > $O/six512_$n.json 2>$O/err && show $O/six512_$n.json
python bench.py --no-cpu --no-traffic --no-host --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 1 > $O/six1024_$n.json 2>$O/err && show $O/six1024_$n.json
done
export TTSWEEP_EXPERIMENT_LIB=gpurun_exp/z64prof.so
python bench.py --no-cpu --no-traffic --no-host --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 0 > $O/p.json 2>$O/p.err
echo "z64 1024x14: $(grep 'tile prof' $O/p.err | tail -1)"
