#!/bin/bash
# Regenerates the six-FS artefacts under profiles/ (bench lines with live traffic, rocprof kernel stats).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r2f
python bench.py --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 2 --warmup 1 --no-cpu --no-host > gpurun_out/r2f/six1024.log 2>&1
echo "six1024 rc=$?"
python bench.py --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 3 --warmup 1 --no-cpu --no-host > gpurun_out/r2f/six512.log 2>&1
echo "six512 rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/prof -- python3 bench.py --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 2 --warmup 1 --no-cpu --no-host --no-traffic > gpurun_out/r2f/prof.log 2>&1
echo "prof rc=$?"
find gpurun_out/r2f/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2f/six1024_kernel_stats.csv
grep -h '"metric"' gpurun_out/r2f/prof.log | tail -1 > gpurun_out/r2f/prof_line.json
