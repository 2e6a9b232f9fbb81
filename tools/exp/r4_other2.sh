#!/bin/bash
# round 4, final kernels: the lines of the other configurations (with HBM-side traffic) -> gpurun_out/r4_other/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_other; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu --no-host --no-hbm-regime"
$B --steps 3 --warmup 1 --nstarts 3 > $O/n3.json 2> $O/n3.err; echo "n3 rc $?"
$B --steps 3 --warmup 1 --nstarts 1 > $O/n1.json 2> $O/n1.err; echo "n1 rc $?"
$B --steps 3 --warmup 1 --starts 4 > $O/start4.json 2> $O/start4.err; echo "start4 rc $?"
$B --steps 2 --warmup 1 --grid 512,512,256 --starts 111 --nstarts 8 > $O/g512_818.json 2> $O/g512_818.err; echo "g512 rc $?"
timeout -k 10 500 $B --steps 1 --warmup 1 --grid 1024,1024,512 --starts 111 --nstarts 14 > $O/g1024_818.json 2> $O/g1024_818.err; echo "g1024 rc $?"
$B --steps 2 --warmup 1 --grid 512,512,256 --starts 111 --nstarts 8 --star six > $O/g512_six.json 2> $O/g512_six.err; echo "g512six rc $?"
bash tools/exp/r4_stripprof.sh > $O/stripprof.txt 2>&1; grep "^prof\|^==\|^ms" $O/stripprof.txt
for f in n3 n1 start4 g512_818 g1024_818 g512_six; do python3 -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$f','ms %.2f'%d['ms_per_step'],'frac %.3f'%r['frac'],'traffic',r.get('traffic'),'alg',r['algorithmic_bytes_per_launch'])"; done
