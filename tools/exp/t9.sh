#!/bin/bash
cd "$GRAFT_REPO_ROOT"
TTSWEEP_EXPERIMENT_LIB=gpurun_exp/tileprof.so python tools/exp/one_sweep.py 1024,1024,512 14 2>&1 | grep "tile prof" | head -1
