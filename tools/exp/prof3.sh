#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/p3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p3/prof -- python3 bench.py --nstarts 3 --steps 5 --warmup 2 --no-cpu --no-host --no-traffic > gpurun_out/p3/log 2>&1
echo rc=$?
grep -h '"metric"' gpurun_out/p3/log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], d['roofline']['launches'], d['roofline']['avg_launch_ms'])"
f=$(find gpurun_out/p3/prof -name "*kernel_stats.csv" | head -1); cut -c1-160 $f | head -8
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/p3/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# gaps between consecutive kernels of the sweep loop
names = [r['Kernel_Name'][:40] for r in rows]
import statistics
gaps = {}
for a, b in zip(rows, rows[1:]):
    k = (a['Kernel_Name'][:25], b['Kernel_Name'][:25])
    gaps.setdefault(k, []).append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:6]:
    print(k, len(v), 'median gap us', round(statistics.median(v), 2), 'mean', round(sum(v) / len(v), 2))
PY
