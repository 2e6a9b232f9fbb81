#!/bin/bash
# round-2 batch B: GPU suite after the hardening changes + PMC passes on the unit kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
export TMPDIR=/tmp
pmc() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu > $O/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pmc icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
pmc sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM
pmc l2 TCC_HIT_sum TCC_MISS_sum
pmc clk GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
python3 - <<'PY'
import csv, glob, collections, os
O="gpurun_out/r2b"
for d in sorted(glob.glob(O+"/pmc_*/")):
    tot=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "sweep_units" in row["Kernel_Name"]:
                tot[row["Counter_Name"]]+=float(row["Counter_Value"]); n[row["Counter_Name"]]+=1
    print(os.path.basename(d.rstrip("/")), {k:(v, n[k]) for k,v in tot.items()})
PY
