#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile or golden_cases" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
