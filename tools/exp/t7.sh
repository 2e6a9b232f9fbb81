#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile" 2>&1 | tail -3
python tools/exp/one_sweep.py 1024,1024,512 14 2>&1 | tail -1
python tools/exp/one_sweep.py 512,512,256 8 2>&1 | tail -1
TTSWEEP_EXPERIMENT_LIB=gpurun_exp/tileprof.so python tools/exp/one_sweep.py 1024,1024,512 14 2>&1 | grep "tile prof" | tail -1
python bench.py --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 2 --warmup 1 --no-traffic --no-host --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric')][0]); print('bench six 1024x14: ms', d['ms_per_step'], 'hbm frac', d['roofline']['frac'])"
python bench.py --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 3 --warmup 1 --no-traffic --no-host --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric')][0]); print('bench six 512x8: ms', d['ms_per_step'], 'hbm frac', d['roofline']['frac'])"
