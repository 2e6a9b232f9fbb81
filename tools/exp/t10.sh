#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for lib in uoparallel-seismic-project_amd/csrc/libttsweep.so gpurun_exp/nowork.so uoparallel-seismic-project_amd/csrc/libttsweep.so gpurun_exp/nowork.so; do
TTSWEEP_EXPERIMENT_LIB=$lib python bench.py --nstarts 3 --steps 8 --warmup 2 --no-traffic --no-host --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric')][0]); print('$lib', d['ms_per_step'])"
done
