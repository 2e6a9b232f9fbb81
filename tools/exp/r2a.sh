#!/bin/bash
# round-2 experiment batch A: baseline, gate-speed sweep for small shards, counter list
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2a; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
python bench.py --no-cpu --steps 5 --warmup 2 > $O/base24.json 2> $O/base24.err
echo "base24 done"; tail -c 600 $O/base24.json
for gs in 3.5 5 7 10; do
  TTSWEEP_GATE_SPEED=$gs python bench.py --no-cpu --nstarts 3 --steps 10 --warmup 2 > $O/n3_gs$gs.json 2>> $O/n3.err
  python - <<PY
import json; d=json.load(open("$O/n3_gs$gs.json")); print("n3 gate $gs", round(d["ms_per_step"],2), d["config"]["passes_per_start_mean"], d["config"]["full_sweep_equivalents_per_start_mean"])
PY
done
for gs in 4.5 6; do
  TTSWEEP_GATE_SPEED=$gs python bench.py --no-cpu --steps 5 --warmup 2 > $O/n24_gs$gs.json 2>> $O/n24.err
  python - <<PY
import json; d=json.load(open("$O/n24_gs$gs.json")); print("n24 gate $gs", round(d["ms_per_step"],2), d["config"]["passes_per_start_mean"], d["config"]["full_sweep_equivalents_per_start_mean"])
PY
done
python -c "
import ctypes
for n in ('libhiprtc.so','/opt/rocm/lib/libhiprtc.so'):
    try:
        ctypes.CDLL(n); print('hiprtc ok', n)
    except OSError as e: print('hiprtc fail', n, e)
"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config_512 or config_1024" 2>&1 | tail -5
