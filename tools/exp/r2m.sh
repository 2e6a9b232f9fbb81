#!/bin/bash
# memory-path counters for both kernels
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
pmc() {  # tag name benchargs -- counters
  local tag=$1 name=$2 bargs=$3; shift 3
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/${tag}_$name -- python3 bench.py --no-cpu --no-traffic --no-host $bargs > $O/${tag}_$name.log 2>&1
  echo "pmc $tag $name rc=$?"
}
S="--steps 1 --warmup 1"
T="--star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 0"
for tag in strip tile; do
  if [ $tag = strip ]; then B="$S"; else B="$T"; fi
  pmc $tag tlb "$B" TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum
  pmc $tag tcp "$B" TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
  pmc $tag ta "$B" TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum
  pmc $tag ea "$B" TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
  pmc $tag ea2 "$B" TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum
  pmc $tag sqlvl "$B" SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
done
python3 - <<'PY'
import csv, glob, collections, os
O="gpurun_out/r2m"
for d in sorted(glob.glob(O+"/*/")):
    tot=collections.defaultdict(float); n=collections.Counter(); dur=0
    pat = "sweep_units" if "strip" in d else "tile_sweep"
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                tot[row["Counter_Name"]]+=float(row["Counter_Value"]); n[row["Counter_Name"]]+=1
    print(os.path.basename(d.rstrip("/")), {k:(v, n[k]) for k,v in tot.items()})
PY
