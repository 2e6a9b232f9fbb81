#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2k; mkdir -p $O
show() { python - <<PY
import json; d=json.load(open("$1")); r=d["roofline"]; h=d["roofline_hbm"]
print("$1", "ms", round(d["ms_per_step"],2), "hbm_frac", round(h["frac"],4), "avg_ms", round(r["avg_launch_ms"],4))
PY
}
for lib in uoparallel-seismic-project_amd/csrc/libttsweep.so gpurun_exp/twg3.so gpurun_exp/twg4.so gpurun_exp/twg10.so; do
export TTSWEEP_EXPERIMENT_LIB=$lib
n=$(basename $lib .so)
python bench.py --no-cpu --no-traffic --no-host --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1 > $O/six512_$n.json 2>$O/err && show $O/six512_$n.json
python bench.py --no-cpu --no-traffic --no-host --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 1 > $O/six1024_$n.json 2>$O/err && show $O/six1024_$n.json
done
