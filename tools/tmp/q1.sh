set -e
run() {
  python bench.py --no-cpu --steps 2 --warmup 1 $EXTRA 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']; print('   ms', round(d['ms_per_step'],1), 'passes', round(c['passes_per_start_mean'],1), 'eq', round(c['full_sweep_equivalents_per_start_mean'],2))
"
}
for dens in 0.5 2.0; do for sp in 0 3 3.5 4; do for r0 in 8; do
  echo "dens $dens speed $sp r0 $r0"
  EXTRA="--nstarts 24" TTSWEEP_COOP_DENSITY=$dens TTSWEEP_GATE_SPEED=$sp TTSWEEP_GATE_R0=$r0 run
done; done; done
echo "big 512x512x256 8 starts"
EXTRA="--grid 512x512x256 --nstarts 8" TTSWEEP_COOP_DENSITY=0.5 run
EXTRA="--grid 512x512x256 --nstarts 8" TTSWEEP_COOP_DENSITY=2.0 run
