#!/bin/bash
# interleaved A/B/.. of builds of the library on the six-FS column workload: GEOM=1024,1024,512 NST=14 r5_ab.sh A.so B.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G=${GEOM:-1024,1024,512}; N=${NST:-14}
out=gpurun_out/r5_ab.txt; : > $out
for rep in 1 2 3; do
for lib in "$@"; do
TTSWEEP_LIB=$lib timeout -k 10 200 python tools/exp/col_probe.py $G $N 3 1 2>&1 | grep -E "^mode 1 order" | tail -1 | sed "s|^|$lib: |" | cut -c1-200 >> $out
done; done
cat $out
