#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2r; mkdir -p $O
export TMPDIR=/tmp
pmc() {  # name counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-traffic --no-host > $O/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
pmc sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM
pmc sq3 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU2 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM
python3 - <<'PY'
import csv, glob, collections, os
O="gpurun_out/r2r"
for d in sorted(glob.glob(O+"/pmc_*/")):
    tot=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "sweep_units" in row["Kernel_Name"]:
                tot[row["Counter_Name"]]+=float(row["Counter_Value"]); n[row["Counter_Name"]]+=1
    print(os.path.basename(d.rstrip("/")), {k:(v, n[k]) for k,v in tot.items()})
PY
