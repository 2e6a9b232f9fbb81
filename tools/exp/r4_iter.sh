#!/bin/bash
# parity smoke + timing of the column kernel: small TILE tests, 512 grid both drivers, 1024 x 14 column driver
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_iter.log; : > $L
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_kernel_small or damaged or tile_six_star_drivers" > gpurun_out/r4_iter_pytest.log 2>&1
echo "pytest small rc $?" >> $L; tail -3 gpurun_out/r4_iter_pytest.log >> $L
timeout -k 10 300 python tools/exp/col_probe.py 512,512,256 8 2 >> $L 2>&1; echo "probe512 rc $?" >> $L
timeout -k 10 500 python tools/exp/col_probe.py 1024,1024,512 14 2 1 >> $L 2>&1; echo "probe1024 rc $?" >> $L
grep -v amdgpu.ids $L
