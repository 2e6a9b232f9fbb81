#!/bin/bash
# round 4, first contact of the column pipelines with the GPU
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_kernel_small or damaged" > gpurun_out/r4_first_pytest.log 2>&1
echo "pytest small rc $?" | tee -a gpurun_out/r4_first.log
tail -5 gpurun_out/r4_first_pytest.log
timeout -k 10 200 python tools/exp/col_probe.py 128,128,64 2 2 >> gpurun_out/r4_first.log 2>&1
echo "probe128 rc $?" | tee -a gpurun_out/r4_first.log
timeout -k 10 300 python tools/exp/col_probe.py 512,512,256 2 2 >> gpurun_out/r4_first.log 2>&1
echo "probe512 rc $?" | tee -a gpurun_out/r4_first.log
cat gpurun_out/r4_first.log
