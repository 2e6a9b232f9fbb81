#!/bin/bash
# Regenerates the round-3 artefacts under profiles/ (run on the GPU box: gpurun -- 'bash tools/exp/r3final.sh').
# Everything goes to gpurun_out/r3final/; copy what is wanted into profiles/ afterwards.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r3final; rm -rf $O; mkdir -p $O
echo "== default bench line (traffic passes, host program, CPU legs, HBM-regime run)"
python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "rc=$?"
echo "== kernel stats of the default workload"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-traffic --no-host --no-hbm-regime > $O/prof_line.json 2> $O/prof.err; echo "rc=$?"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
rm -rf $O/prof
echo "== HBM regime: six-FS 1024x1024x512 x 14, line with live traffic + kernel stats"
python bench.py --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 2 --warmup 1 --no-cpu --no-host > $O/six1024_bench_line.json 2> $O/six1024.err; echo "rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof6 -- python3 bench.py --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 2 --warmup 1 --no-cpu --no-host --no-traffic > $O/six1024_prof_line.json 2> $O/six1024_prof.err; echo "rc=$?"
find $O/prof6 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/six1024_kernel_stats.csv
python tools/exp/trace_six.py $O/prof6 270 > $O/six1024_launch_classes.txt 2>&1
rm -rf $O/prof6
python bench.py --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 3 --warmup 1 --no-cpu --no-host > $O/six512_bench_line.json 2> $O/six512.err; echo "rc=$?"
echo "== the other BASELINE configurations (818-FS)"
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime"
$B --steps 5 --warmup 1 --nstarts 3 > $O/n3_line.json 2> $O/n3.err; echo "n3 rc=$?"
$B --steps 5 --warmup 1 --starts 4 > $O/start4_line.json 2> $O/start4.err; echo "start4 rc=$?"
$B --steps 2 --warmup 1 --grid 512,512,256 --starts 111 --nstarts 8 > $O/818_512_line.json 2> $O/818_512.err; echo "818_512 rc=$?"
$B --steps 1 --warmup 1 --grid 1024,1024,512 --starts 111 --nstarts 14 > $O/818_1024_line.json 2> $O/818_1024.err; echo "818_1024 rc=$?"
$B --steps 3 --warmup 1 --prepass 146 > $O/prepass146_line.json 2> $O/prepass146.err; echo "prepass146 rc=$?"
$B --steps 3 --warmup 1 --prepass 98 > $O/prepass98_line.json 2> $O/prepass98.err; echo "prepass98 rc=$?"
$B --steps 3 --warmup 1 --nstarts 3 --prepass 98 > $O/n3_prepass98_line.json 2> $O/n3_prepass98.err; echo "n3 prepass98 rc=$?"
echo "== two ranks on this one GPU (gloo rehearsal of the sharded path: shards, host gather)"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --backend gloo --grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1 --no-cpu > $O/gloo2_512_line.json 2> $O/gloo2_512.err; echo "gloo2-512 rc=$?"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu > $O/gloo2_241_line.json 2> $O/gloo2_241.err; echo "gloo2-241 rc=$?"
echo "== SQ counters of the unit kernel"
bash tools/exp/pmc.sh r3sq1 "--steps 2 --warmup 1 --no-hbm-regime" sweep_units SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU > $O/pmc_sq1.txt 2>&1
bash tools/exp/pmc.sh r3sq2 "--steps 2 --warmup 1 --no-hbm-regime" sweep_units SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA > $O/pmc_sq2.txt 2>&1
bash tools/exp/pmc.sh r3sq3 "--steps 2 --warmup 1 --no-hbm-regime --star six --grid 1024,1024,512 --starts 111 --nstarts 14" tile_six SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT > $O/pmc_sq3.txt 2>&1
for f in 1 2 3; do tail -n 1 $O/pmc_sq$f.txt; done
rm -rf gpurun_out/pmc_r3sq1 gpurun_out/pmc_r3sq2 gpurun_out/pmc_r3sq3
echo "== summary"
for f in $O/*line.json; do
python3 - "$f" <<'PY'
import json,sys,os
f=sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d["roofline"]; e=d.get("end_to_end_host_program") or {}
    print(os.path.basename(f).ljust(28), "ms %.2f"%d["ms_per_step"], r["bound"], "frac %.3f"%r["frac"], "launch_ms %.4f"%r["avg_launch_ms"], "n", r["launches"], "eq/start %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"], "traffic", r.get("traffic"), "gather", d["config"].get("gather",{}).get("path"), "loop_s", e.get("sweep_loop_wall_seconds"))
    h=d.get("roofline_hbm_regime")
    if h: print("   hbm regime: frac %.3f ms %.1f launch_ms %.4f traffic %s cpu %s"%(h["frac"],h["ms_per_solve"],h["avg_launch_ms"],h.get("traffic"),h.get("cpu_baseline",{}).get("value")))
except Exception as ex: print(os.path.basename(f), "FAILED", ex)
PY
done
