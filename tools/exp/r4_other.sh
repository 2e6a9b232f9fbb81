#!/bin/bash
# round 4: the lines of the other configurations (with HBM-side traffic for the grids that leave the caches), the
# one-launch STRIP solve's counts of units that improved nothing (with the deferral), the column kernel's phase profile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_other; mkdir -p $O
B="python bench.py --no-cpu --no-host --no-hbm-regime"
$B --steps 3 --warmup 1 --nstarts 3 > $O/n3.json 2> $O/n3.err; echo "n3 rc $?"
$B --steps 3 --warmup 1 --nstarts 1 > $O/n1.json 2> $O/n1.err; echo "n1 rc $?"
$B --steps 3 --warmup 1 --starts 4 > $O/start4.json 2> $O/start4.err; echo "start4 rc $?"
$B --steps 2 --warmup 1 --grid 512,512,256 --starts 111 --nstarts 8 > $O/g512_818.json 2> $O/g512_818.err; echo "g512 rc $?"
timeout -k 10 500 $B --steps 1 --warmup 1 --grid 1024,1024,512 --starts 111 --nstarts 14 > $O/g1024_818.json 2> $O/g1024_818.err; echo "g1024 rc $?"
$B --steps 2 --warmup 1 --grid 512,512,256 --starts 111 --nstarts 8 --star six > $O/g512_six.json 2> $O/g512_six.err; echo "g512six rc $?"
$B --no-traffic --steps 2 --warmup 1 --lib gpurun_exp/asyncstats.so > $O/asyncstats.json 2> $O/asyncstats.err; echo "asyncstats rc $?"; grep "one-launch solve" $O/asyncstats.err | tail -2
TTSWEEP_LIB=gpurun_exp/colprof.so timeout -k 10 300 python tools/exp/col_probe.py 1024,1024,512 14 2 1 > $O/colprof.txt 2>&1; echo "colprof rc $?"; grep "column prof" $O/colprof.txt | tail -3
