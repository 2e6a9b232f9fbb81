#!/bin/bash
# Regenerates the round-4 artefacts under profiles/ (run on the GPU box: gpurun -- 'bash tools/exp/r4final.sh').
# Everything goes to gpurun_out/r4final/; tools/exp/collect_r4.py copies what is wanted into profiles/ afterwards.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r4final; rm -rf $O; mkdir -p $O
echo "== default bench line (traffic passes, host program, CPU legs, HBM-regime run)"
timeout -k 10 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "rc=$?"
echo "== kernel stats of the default workload"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-traffic --no-host --no-hbm-regime > $O/prof_line.json 2> $O/prof.err; echo "rc=$?"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
rm -rf $O/prof
echo "== HBM regime: six-FS 1024x1024x512 x 14, line with live traffic + kernel stats"
timeout -k 10 600 python bench.py --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 3 --warmup 1 --no-cpu --no-host > $O/six1024_bench_line.json 2> $O/six1024.err; echo "rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof6 -- python3 bench.py --star six --grid 1024,1024,512 --starts 111 --nstarts 14 --steps 3 --warmup 1 --no-cpu --no-host --no-traffic > $O/six1024_prof_line.json 2> $O/six1024_prof.err; echo "rc=$?"
find $O/prof6 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/six1024_kernel_stats.csv
rm -rf $O/prof6
timeout -k 10 300 python bench.py --star six --grid 512,512,256 --starts 111 --nstarts 8 --steps 3 --warmup 1 --no-cpu --no-host > $O/six512_bench_line.json 2> $O/six512.err; echo "rc=$?"
echo "== one ordering sweep with every tile due (a converged box solved again)"
timeout -k 10 300 python tools/exp/one_sweep.py 1024,1024,512 14 > $O/six_full_sweep.txt 2>&1; echo "rc=$?"
echo "== where the column wavefronts' time goes (-DTTSWEEP_COL_PROFILE build)"
TTSWEEP_LIB=gpurun_exp/colprof.so timeout -k 10 400 python tools/exp/col_probe.py 1024,1024,512 14 2 1 > $O/col_profile.txt 2>&1; echo "rc=$?"
echo "== the small shards (818-FS)"
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime"
$B --steps 5 --warmup 1 --nstarts 3 > $O/n3_line.json 2> $O/n3.err; echo "n3 rc=$?"
$B --steps 5 --warmup 1 --nstarts 1 > $O/n1_line.json 2> $O/n1.err; echo "n1 rc=$?"
$B --steps 5 --warmup 1 --starts 4 > $O/start4_line.json 2> $O/start4.err; echo "start4 rc=$?"
echo "== units that improved nothing (-DTTSWEEP_ASYNC_STATS build)"
$B --steps 2 --warmup 1 --lib gpurun_exp/asyncstats.so > $O/asyncstats_line.json 2> $O/asyncstats.err; echo "asyncstats rc=$?"
$B --steps 2 --warmup 1 --nstarts 3 --lib gpurun_exp/asyncstats.so > $O/asyncstats_n3_line.json 2> $O/asyncstats_n3.err; echo "asyncstats n3 rc=$?"
echo "== SQ counters of the unit kernel and of the column kernel"
bash tools/exp/pmc.sh r4sq1 "--steps 2 --warmup 1 --no-hbm-regime" sweep_units SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU > $O/pmc_sq1.txt 2>&1
bash tools/exp/pmc.sh r4sq2 "--steps 2 --warmup 1 --no-hbm-regime" sweep_units SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA > $O/pmc_sq2.txt 2>&1
bash tools/exp/pmc.sh r4sq3 "--steps 2 --warmup 1 --no-hbm-regime --star six --grid 1024,1024,512 --starts 111 --nstarts 14" column_solve SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU > $O/pmc_sq3.txt 2>&1
bash tools/exp/pmc.sh r4sq4 "--steps 2 --warmup 1 --no-hbm-regime --star six --grid 1024,1024,512 --starts 111 --nstarts 14" column_solve SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR > $O/pmc_sq4.txt 2>&1
for f in 1 2 3 4; do tail -n 1 $O/pmc_sq$f.txt; done
rm -rf gpurun_out/pmc_r4sq1 gpurun_out/pmc_r4sq2 gpurun_out/pmc_r4sq3 gpurun_out/pmc_r4sq4
echo "== summary"
for f in $O/*line.json; do
python3 - "$f" <<'PY'
import json,sys,os
f=sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d["roofline"]; e=d.get("end_to_end_host_program") or {}
    print(os.path.basename(f).ljust(28), "ms %.2f"%d["ms_per_step"], r["bound"], "frac %.3f"%r["frac"], "over solve", r.get("frac_over_solve"), "launch_ms %.4f"%r["avg_launch_ms"], "n", r["launches"], "traffic", r.get("traffic"), "fallbacks", d["config"].get("fallbacks"), "loop_s", e.get("sweep_loop_wall_seconds"))
    h=d.get("roofline_hbm_regime")
    if h: print("   hbm regime: frac %.3f over solve %s ms %.1f launch_ms %.4f traffic %s cpu %s"%(h["frac"],h.get("frac_over_solve"),h["ms_per_solve"],h["avg_launch_ms"],h.get("traffic"),h.get("cpu_baseline",{}).get("value")))
except Exception as ex: print(os.path.basename(f), "FAILED", ex)
PY
done
grep -h "async stats\|column prof" $O/*.err $O/col_profile.txt | head
cat $O/six_full_sweep.txt | grep -v amdgpu.ids
