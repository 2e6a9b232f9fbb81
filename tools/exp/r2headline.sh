#!/bin/bash
# Regenerates the headline artefacts under profiles/ (default bench line, rocprof kernel stats).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r2h
python bench.py > gpurun_out/r2h/bench.log 2>&1; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2h/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-traffic --no-host > gpurun_out/r2h/prof.log 2>&1; echo "prof rc=$?"
ls -t $(find gpurun_out/r2h/prof -name "*kernel_stats.csv") | head -1 | xargs -I{} cp {} gpurun_out/r2h/kernel_stats.csv
bash tools/exp/other_configs.sh
