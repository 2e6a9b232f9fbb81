#!/bin/bash
# round-2 batch C: workgroups per CU / register cap variants of the unit kernel (A/B in one session)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c; mkdir -p $O
run() { # name lib extra-args
  local name=$1 lib=$2; shift 2
  TTSWEEP_EXPERIMENT_LIB=$lib python bench.py --no-cpu --steps 5 --warmup 2 "$@" > $O/$name.json 2>> $O/err.log
  python - <<PY
import json; d=json.load(open("$O/$name.json")); print("$name", round(d["ms_per_step"],2), round(d["roofline_valu"]["frac"],3), d["config"]["passes_per_start_mean"], round(d["config"]["full_sweep_equivalents_per_start_mean"],2))
PY
}
for rep in 1 2; do
run base_$rep uoparallel-seismic-project_amd/csrc/libttsweep.so
run wg3_$rep gpurun_exp/wg3.so
run wg3g2_$rep gpurun_exp/wg3g2.so
run wg2g3_$rep gpurun_exp/wg2g3.so
done
run base_n3 uoparallel-seismic-project_amd/csrc/libttsweep.so --nstarts 3
run wg3_n3 gpurun_exp/wg3.so --nstarts 3
