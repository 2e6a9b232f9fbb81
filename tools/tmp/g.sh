run() {
  python bench.py --no-cpu --steps 3 --warmup 1 $EXTRA 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']; print('   ms', round(d['ms_per_step'],1), 'passes', round(c['passes_per_start_mean'],1), 'eq', round(c['full_sweep_equivalents_per_start_mean'],2))
"
}
for ns in 3 24; do for sp in 3 3.5 4 4.5 5; do for r0 in 8 16; do
  echo "nstarts $ns speed $sp r0 $r0"
  EXTRA="--nstarts $ns" TTSWEEP_GATE_SPEED=$sp TTSWEEP_GATE_R0=$r0 run
done; done; done
