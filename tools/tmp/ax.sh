run() {
  python bench.py --no-cpu --steps 1 --warmup 1 $EXTRA 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']; print('   ms', round(d['ms_per_step'],1), 'passes', round(c['passes_per_start_mean'],1), 'eq', round(c['full_sweep_equivalents_per_start_mean'],2))
    elif 'rror' in l: print(l.rstrip())
"
}
echo "512 grid gate speeds"
for sp in 5 6 7 8 10; do echo "speed $sp"; TTSWEEP_GATE_SPEED=$sp EXTRA="--grid 512,512,256 --nstarts 8" run; done
echo "256x256x128 8 starts"
for sp in 3.5 5 7; do echo "speed $sp"; TTSWEEP_GATE_SPEED=$sp EXTRA="--grid 256,256,128 --nstarts 8" run; done
