#!/bin/bash
# round 3, second GPU pass: parity suite, tile kernel after the stamp change, A/B of the unit kernel
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r3b_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3b_pytest.log
tail -8 gpurun_out/r3b_pytest.log
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime"
$B --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 2 --warmup 1 > gpurun_out/r3b_six1024.json 2> gpurun_out/r3b_six1024.err; echo "six1024 rc=$?"
$B --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 3 --warmup 1 > gpurun_out/r3b_six512.json 2> gpurun_out/r3b_six512.err; echo "six512 rc=$?"
for lib in nopin slab3; do timeout -k 10 200 python tools/exp/check_lib.py gpurun_exp/$lib.so 2>&1 | tail -3; done
$B --steps 5 --warmup 1 > gpurun_out/r3b_default.json 2> gpurun_out/r3b_default.err; echo "default rc=$?"
$B --steps 5 --warmup 1 --lib gpurun_exp/nopin.so > gpurun_out/r3b_nopin.json 2> gpurun_out/r3b_nopin.err; echo "nopin rc=$?"
$B --steps 5 --warmup 1 --lib gpurun_exp/slab3.so > gpurun_out/r3b_slab3.json 2> gpurun_out/r3b_slab3.err; echo "slab3 rc=$?"
$B --steps 5 --warmup 1 --nstarts 3 > gpurun_out/r3b_n3.json 2> gpurun_out/r3b_n3.err; echo "n3 rc=$?"
$B --steps 5 --warmup 1 --nstarts 3 --lib gpurun_exp/slab3.so > gpurun_out/r3b_n3_slab3.json 2> gpurun_out/r3b_n3_slab3.err; echo "n3 slab3 rc=$?"
$B --steps 2 --warmup 1 --grid 512,512,256 --starts 111 --nstarts 8 > gpurun_out/r3b_818_512.json 2> gpurun_out/r3b_818_512.err; echo "818_512 rc=$?"
$B --steps 2 --warmup 1 --grid 512,512,256 --starts 111 --nstarts 8 --lib gpurun_exp/slab3.so > gpurun_out/r3b_818_512_slab3.json 2> gpurun_out/r3b_818_512_slab3.err; echo "818_512 slab3 rc=$?"
for f in six1024 six512 default nopin slab3 n3 n3_slab3 818_512 818_512_slab3; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r3b_{f}.json").read().strip().splitlines()[-1])
    r=d["roofline"]; print(f, "ms_per_step %.2f"%d["ms_per_step"], r["bound"], "frac %.3f"%r["frac"], "avg_launch_ms %.4f"%r["avg_launch_ms"], "launches", r["launches"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"])
except Exception as e: print(f, "FAILED", e)
PY
done
