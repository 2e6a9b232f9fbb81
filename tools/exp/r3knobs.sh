#!/bin/bash
# schedule knobs re-measured after the round-3 changes (gate speed, units of one / two planes)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3knobs; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime --steps 5 --warmup 1"
run() { name=$1; shift; $B "$@" > $O/$name.json 2> $O/$name.err; python3 - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[2].ljust(22), "ms %.2f"%d["ms_per_step"], "frac %.3f"%r["frac"], "launch_ms %.4f"%r["avg_launch_ms"], "n", r["launches"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run n24_default
run n24_gate3 --gate-speed 3.0
run n24_gate4 --gate-speed 4.0
run n24_np1 --pair-min-starts 100
run n16_default --nstarts 16
run n16_np2 --nstarts 16 --pair-min-starts 0
run n8_default --nstarts 8
run n8_np2 --nstarts 8 --pair-min-starts 0
run n3_default --nstarts 3
run n3_np2 --nstarts 3 --pair-min-starts 0
run n3_gate3 --nstarts 3 --gate-speed 3.0
run n3_gate4 --nstarts 3 --gate-speed 4.0
run n3_gate5 --nstarts 3 --gate-speed 5.0
run n1_default --nstarts 1
