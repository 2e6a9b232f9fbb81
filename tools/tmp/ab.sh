cd $GRAFT_REPO_ROOT
one() {
  cp tools/tmp/lib$1.so uoparallel-seismic-project_amd/csrc/libttsweep.so
  python bench.py --no-cpu --steps 4 --warmup 1 $2 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1 $2 ms', round(d['ms_per_step'],2))
"
}
for i in 1 2 3; do one A ""; one B ""; done
for i in 1 2; do one A "--nstarts 3"; one B "--nstarts 3"; done
