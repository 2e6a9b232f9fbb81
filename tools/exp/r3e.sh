#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r3e_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3e_pytest.log
tail -6 gpurun_out/r3e_pytest.log
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime"
$B --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 2 --warmup 1 > gpurun_out/r3e_six1024.json 2> gpurun_out/r3e_six1024.err; echo "six1024 rc=$?"
$B --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 3 --warmup 1 > gpurun_out/r3e_six512.json 2> gpurun_out/r3e_six512.err; echo "six512 rc=$?"
$B --steps 5 --warmup 1 > gpurun_out/r3e_default.json 2> gpurun_out/r3e_default.err; echo "default rc=$?"
$B --steps 5 --warmup 1 --nstarts 3 > gpurun_out/r3e_n3.json 2> gpurun_out/r3e_n3.err; echo "n3 rc=$?"
$B --steps 5 --warmup 1 --starts 4 > gpurun_out/r3e_s4.json 2> gpurun_out/r3e_s4.err; echo "s4 rc=$?"
$B --steps 2 --warmup 1 --grid 512,512,256 --starts 111 --nstarts 8 > gpurun_out/r3e_818_512.json 2> gpurun_out/r3e_818_512.err; echo "818_512 rc=$?"
rm -rf gpurun_out/r3e_stats; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3e_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-traffic --no-host --no-hbm-regime > gpurun_out/r3e_default_prof.json 2> gpurun_out/r3e_default_prof.err; echo "prof rc=$?"
f=$(find gpurun_out/r3e_stats -name "*kernel_stats.csv" | head -1); cut -c1-100,300-420 $f | head -6
find gpurun_out/r3e_stats -name "*kernel_trace.csv" -size +20M -delete
for f in six1024 six512 default n3 s4 818_512; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r3e_{f}.json").read().strip().splitlines()[-1])
    r=d["roofline"]; print(f, "ms_per_step %.2f"%d["ms_per_step"], r["bound"], "frac %.3f"%r["frac"], "avg_launch_ms %.4f"%r["avg_launch_ms"], "launches", r["launches"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"])
except Exception as e: print(f, "FAILED", e)
PY
done
