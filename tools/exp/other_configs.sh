#!/bin/bash
# The other BASELINE configurations (818-FS) and the small shards: one bench line each.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/oc
python bench.py --grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1 --no-cpu --no-host --no-traffic > gpurun_out/oc/g512_818.log 2>&1; echo "g512 rc=$?"
python bench.py --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 1 --no-cpu --no-host --no-traffic > gpurun_out/oc/g1024_818.log 2>&1; echo "g1024 rc=$?"
python bench.py --nstarts 3 --steps 5 --warmup 2 --no-cpu --no-host --no-traffic > gpurun_out/oc/n3.log 2>&1; echo "n3 rc=$?"
python bench.py --starts 4 --steps 5 --warmup 2 --no-cpu --no-host --no-traffic > gpurun_out/oc/start4.log 2>&1; echo "start4 rc=$?"
