#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2u; mkdir -p $O
export TMPDIR=/tmp
pmc() {  # name benchargs -- counters
  local name=$1 bargs=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 bench.py --no-cpu --no-traffic --no-host $bargs > $O/$name.log 2>&1
  echo "pmc $name rc=$?"
}
T="--star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 0"
pmc tile_tlb "$T" TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum
pmc tile_tcp "$T" TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pmc tile_ea "$T" TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
pmc tile_l2 "$T" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum
python3 - <<'PY'
import csv, glob, collections, os
O="gpurun_out/r2u"
for d in sorted(glob.glob(O+"/*/")):
    tot=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "tile_sweep" in row["Kernel_Name"]:
                tot[row["Counter_Name"]]+=float(row["Counter_Value"]); n[row["Counter_Name"]]+=1
    print(os.path.basename(d.rstrip("/")), {k:(v, n[k]) for k,v in tot.items()})
PY
