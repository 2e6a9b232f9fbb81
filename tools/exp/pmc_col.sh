#!/bin/bash
# One rocprofv3 --pmc pass (at most 8 SQ counters) around tools/exp/col_probe.py, summed for column_solve_kernel.
#   bash tools/exp/pmc_col.sh NAME LIB "PROBE ARGS" COUNTER [COUNTER ...]      (LIB: "" = the default build)
cd "${GRAFT_REPO_ROOT:-.}"
name=$1; lib=$2; pargs=$3; shift 3
export TMPDIR=/tmp
out=gpurun_out/pmc_$name; mkdir -p "$out"
[ -n "$lib" ] && export TTSWEEP_LIB=$lib
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out" -- \
    python3 tools/exp/col_probe.py $pargs > "$out.log" 2>&1 || { echo "pmc $name failed"; tail -5 "$out.log"; exit 1; }
python3 - "$out" column_solve <<'PY'
import collections, csv, glob, sys
tot, n = collections.defaultdict(float), collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
print(sys.argv[1], {k: (v, n[k]) for k, v in tot.items()})
PY
