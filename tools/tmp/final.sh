cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_err.txt
cd /tmp
rm -rf /tmp/st; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /tmp/st.log 2>&1
cp $(find /tmp/st -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/kernel_stats.csv
rm -f $GRAFT_REPO_ROOT/gpurun_out/traffic.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /tmp/pm.log 2>&1
  python3 - $(find /tmp/pm -name '*counter_collection.csv' | head -1) $c <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/traffic.txt
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-30:]
    tot[k] += float(r['Counter_Value']); n[k] += 1
for k in tot: print(sys.argv[2], k, n[k], tot[k])
PY
done
rm -f $GRAFT_REPO_ROOT/gpurun_out/sq.txt
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /tmp/pm.log 2>&1
  python3 - $(find /tmp/pm -name '*counter_collection.csv' | head -1) <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/sq.txt
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'sweep_units' in r['Kernel_Name']:
        tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in tot: print(k, n[k], tot[k])
PY
done
