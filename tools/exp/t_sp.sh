#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not tile and not 1024 and not 512" 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-traffic --no-host --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric')][0]); print('headline ms', d['ms_per_step'], 'valu frac', d['roofline']['frac'], d['config']['full_sweep_equivalents_per_start_mean'])"
python bench.py --nstarts 3 --steps 8 --warmup 2 --no-traffic --no-host --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric')][0]); print('3 starts ms', d['ms_per_step'], 'valu frac', d['roofline']['frac'], d['config']['full_sweep_equivalents_per_start_mean'])"
