#!/bin/bash
# round 3, first GPU pass: parity suite, then the lines the refactor should have moved
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3a_pytest.log
tail -5 gpurun_out/r3a_pytest.log
python bench.py --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 2 --warmup 1 --no-cpu --no-host --no-traffic > gpurun_out/r3a_six1024.json 2> gpurun_out/r3a_six1024.err; echo "six1024 rc=$?"
python bench.py --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 3 --warmup 1 --no-cpu --no-host --no-traffic > gpurun_out/r3a_six512.json 2> gpurun_out/r3a_six512.err; echo "six512 rc=$?"
python bench.py --steps 3 --warmup 1 --no-cpu --no-traffic > gpurun_out/r3a_default.json 2> gpurun_out/r3a_default.err; echo "default rc=$?"
python bench.py --steps 3 --warmup 1 --no-cpu --no-traffic --no-host --prepass 98 > gpurun_out/r3a_prepass98.json 2> gpurun_out/r3a_prepass98.err; echo "prepass98 rc=$?"
python bench.py --steps 3 --warmup 1 --no-cpu --no-traffic --no-host --prepass 146 > gpurun_out/r3a_prepass146.json 2> gpurun_out/r3a_prepass146.err; echo "prepass146 rc=$?"
python bench.py --steps 5 --warmup 1 --no-cpu --no-traffic --no-host --nstarts 3 > gpurun_out/r3a_n3.json 2> gpurun_out/r3a_n3.err; echo "n3 rc=$?"
python bench.py --steps 5 --warmup 1 --no-cpu --no-traffic --no-host --nstarts 3 --prepass 98 > gpurun_out/r3a_n3_prepass98.json 2> gpurun_out/r3a_n3_prepass98.err; echo "n3 prepass rc=$?"
python tools/exp/host_phases.py > gpurun_out/r3a_host_phases.txt 2>&1; echo "host phases rc=$?"
for f in six1024 six512 default prepass98 prepass146 n3 n3_prepass98; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r3a_{f}.json").read().strip().splitlines()[-1])
    r=d["roofline"]; print(f, "ms_per_step %.2f"%d["ms_per_step"], r["bound"], "frac %.3f"%r["frac"], "avg_launch_ms %.4f"%r["avg_launch_ms"], "launches", r["launches"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"], "e2e", d.get("end_to_end_host_program"))
except Exception as e: print(f, "FAILED", e)
PY
done
cat gpurun_out/r3a_host_phases.txt
