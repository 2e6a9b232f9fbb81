#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "tile or golden_cases or 1024 or 512 or fuzz or tiny or zero_vel" > gpurun_out/r3d_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3d_pytest.log
tail -4 gpurun_out/r3d_pytest.log
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime"
$B --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 2 --warmup 1 > gpurun_out/r3d_six1024.json 2> gpurun_out/r3d_six1024.err; echo "six1024 rc=$?"
$B --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 3 --warmup 1 > gpurun_out/r3d_six512.json 2> gpurun_out/r3d_six512.err; echo "six512 rc=$?"
rm -rf gpurun_out/r3d_trace; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3d_trace -- python3 bench.py --no-cpu --no-traffic --no-host --no-hbm-regime --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 0 > gpurun_out/r3d_six_traced.json 2> gpurun_out/r3d_six_traced.err; echo "trace rc=$?"
python tools/exp/trace_six.py gpurun_out/r3d_trace 270 > gpurun_out/r3d_trace_summary.txt 2>&1; cat gpurun_out/r3d_trace_summary.txt
find gpurun_out/r3d_trace -name "*.csv" -size +20M -delete
for f in six1024 six512; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r3d_{f}.json").read().strip().splitlines()[-1])
    r=d["roofline"]; print(f, "ms_per_step %.2f"%d["ms_per_step"], r["bound"], "frac %.3f"%r["frac"], "avg_launch_ms %.4f"%r["avg_launch_ms"], "launches", r["launches"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"])
except Exception as e: print(f, "FAILED", e)
PY
done
