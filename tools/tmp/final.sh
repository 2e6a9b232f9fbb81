cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_err.txt
python bench.py --no-cpu --nstarts 3 > gpurun_out/bench_3.json 2>> gpurun_out/bench_err.txt
python bench.py --no-cpu --steps 1 --warmup 1 --grid 512,512,256 --nstarts 8 > gpurun_out/bench_512.json 2>> gpurun_out/bench_err.txt
python bench.py --no-cpu --steps 1 --warmup 1 --grid 1024,1024,512 --nstarts 2 > gpurun_out/bench_1024.json 2>> gpurun_out/bench_err.txt
python bench.py --no-cpu --steps 1 --warmup 1 --grid 512,512,256 --nstarts 2 --star six > gpurun_out/bench_six.json 2>> gpurun_out/bench_err.txt
cd /tmp
rm -rf /tmp/st; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /tmp/st.log 2>&1
cp $(find /tmp/st -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/kernel_stats.csv
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
