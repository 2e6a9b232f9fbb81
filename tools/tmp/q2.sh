run() {
  python bench.py --no-cpu --steps 1 --warmup 1 $EXTRA 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']; print('   ms', round(d['ms_per_step'],1), 'passes', round(c['passes_per_start_mean'],1), 'eq', round(c['full_sweep_equivalents_per_start_mean'],2))
    elif 'rror' in l: print(l.rstrip())
"
}
echo "big 512x512x256 8 starts dens 0.5 / 2.0 / ungated"
EXTRA="--grid 512,512,256 --nstarts 8" TTSWEEP_COOP_DENSITY=0.5 run
EXTRA="--grid 512,512,256 --nstarts 8" TTSWEEP_COOP_DENSITY=2.0 run
EXTRA="--grid 512,512,256 --nstarts 8" TTSWEEP_GATE_SPEED=0 run
echo "1024x1024x512 2 starts"
EXTRA="--grid 1024,1024,512 --nstarts 2" run
echo "stars 3, 5, six on 241 grid (4 starts)"
for st in 3 5 six; do EXTRA="--star $st --nstarts 4" run; EXTRA="--star $st --nstarts 4" TTSWEEP_GATE_SPEED=0 run; done
