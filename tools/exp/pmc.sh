#!/bin/bash
# One rocprofv3 --pmc pass (at most 8 SQ or 4 TCP/TCC counters) around bench.py, summed per kernel.
#   gpurun -- 'bash tools/exp/pmc.sh NAME "BENCH ARGS" KERNEL_SUBSTRING COUNTER [COUNTER ...]'
# e.g. bash tools/exp/pmc.sh sq1 "--steps 2 --warmup 1" sweep_units SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU
# Counters go in their own run (never together with --stats / --sys-trace); TA_* counters crash
# rocprofv3 on this image.  Output: gpurun_out/pmc_NAME/ (CSV) and one summary line.
cd "${GRAFT_REPO_ROOT:-.}"
name=$1; bargs=$2; pat=$3; shift 3
export TMPDIR=/tmp
out=gpurun_out/pmc_$name; mkdir -p "$out"
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out" -- \
    python3 bench.py --no-cpu --no-traffic --no-host $bargs > "$out.log" 2>&1 || { echo "pmc $name failed"; exit 1; }
python3 - "$out" "$pat" <<'PY'
import collections, csv, glob, sys
tot, n = collections.defaultdict(float), collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
print(sys.argv[1], {k: (v, n[k]) for k, v in tot.items()})
PY
