#!/bin/bash
# Regenerates the round-5 artefacts under profiles/ (run on the GPU box: gpurun -- 'bash tools/exp/r5final.sh [part]').
# Everything goes to gpurun_out/r5final/; tools/exp/collect_r5.py copies what is wanted into profiles/ afterwards.
# Builds wanted in gpurun_exp/: asyncstats.so (-DTTSWEEP_ASYNC_STATS), colprof.so (-DTTSWEEP_COL_PROFILE), stripprof.so (-DTTSWEEP_PROFILE).
# part: a (bench line, kernel stats, small shards, phases), b (HBM regime, column profile, counters), c (other configs), d (CPU leg B2, 190 s); default a, b, c
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
part=${1:-abc}
O=gpurun_out/r5final; mkdir -p $O
B="python bench.py --no-cpu --no-traffic --no-host --no-hbm-regime"
if [[ $part == *a* ]]; then
echo "== default bench line (traffic passes, host program, CPU legs, HBM-regime run)"
timeout -k 10 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "rc=$?"
echo "== kernel stats of the default workload"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-traffic --no-host --no-hbm-regime > $O/prof_line.json 2> $O/prof.err; echo "rc=$?"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
rm -rf $O/prof
echo "== the small shards (818-FS)"
$B --steps 5 --warmup 1 --nstarts 3 > $O/n3_line.json 2> $O/n3.err; echo "n3 rc=$?"
$B --steps 5 --warmup 1 --nstarts 2 > $O/n2_line.json 2> $O/n2.err; echo "n2 rc=$?"
$B --steps 5 --warmup 1 --nstarts 1 > $O/n1_line.json 2> $O/n1.err; echo "n1 rc=$?"
$B --steps 5 --warmup 1 --starts 4 > $O/start4_line.json 2> $O/start4.err; echo "start4 rc=$?"
$B --steps 5 --warmup 1 --nstarts 3 --waves 4 > $O/n3_waves4_line.json 2> $O/n3w4.err; echo "n3 waves 4 rc=$?"
$B --steps 5 --warmup 1 --nstarts 1 --waves 4 > $O/n1_waves4_line.json 2> $O/n1w4.err; echo "n1 waves 4 rc=$?"
$B --steps 5 --warmup 1 --nstarts 3 --handoff 3 > $O/n3_handoff3_line.json 2> $O/n3h3.err; echo "n3 handoff 3 rc=$?"
echo "== units that improved nothing (-DTTSWEEP_ASYNC_STATS build)"
$B --steps 2 --warmup 1 --lib gpurun_exp/asyncstats.so > $O/asyncstats_line.json 2> $O/asyncstats.err; echo "asyncstats rc=$?"
$B --steps 2 --warmup 1 --nstarts 3 --lib gpurun_exp/asyncstats.so > $O/asyncstats_n3_line.json 2> $O/asyncstats_n3.err; echo "asyncstats n3 rc=$?"
echo "== phase stamps of the unit kernel (-DTTSWEEP_PROFILE build)"
: > $O/strip_phases.txt
for n in 24 3 1; do
  echo "== $n starts" >> $O/strip_phases.txt
  timeout -k 10 200 $B --steps 2 --warmup 1 --nstarts $n --lib gpurun_exp/stripprof.so > $O/stripprof_$n.json 2> $O/stripprof_$n.err
  grep "^prof" $O/stripprof_$n.err | tail -2 >> $O/strip_phases.txt
  python3 -c "import json;d=json.loads(open('$O/stripprof_$n.json').read().strip().splitlines()[-1]);print('ms',d['ms_per_step'])" >> $O/strip_phases.txt
done
fi
if [[ $part == *b* ]]; then
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
echo "== starts resident per launch"
: > $O/col_batch.txt
for n in 7 14 28 56; do timeout -k 10 400 python tools/exp/col_probe.py 1024,1024,512 $n 2 1 2>&1 | grep "^mode 1 order" | tail -1 >> $O/col_batch.txt; done
echo "== SQ counters of the unit kernel and of the column kernel"
bash tools/exp/pmc.sh r5sq1 "--steps 2 --warmup 1 --no-hbm-regime" sweep_units SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU > $O/pmc_sq1.txt 2>&1
bash tools/exp/pmc.sh r5sq2 "--steps 2 --warmup 1 --no-hbm-regime" sweep_units SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA > $O/pmc_sq2.txt 2>&1
bash tools/exp/pmc.sh r5sq3 "--steps 2 --warmup 1 --no-hbm-regime --star six --grid 1024,1024,512 --starts 111 --nstarts 14" column_solve SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU > $O/pmc_sq3.txt 2>&1
bash tools/exp/pmc.sh r5sq4 "--steps 2 --warmup 1 --no-hbm-regime --star six --grid 1024,1024,512 --starts 111 --nstarts 14" column_solve SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR > $O/pmc_sq4.txt 2>&1
for f in 1 2 3 4; do tail -n 1 $O/pmc_sq$f.txt; done
rm -rf gpurun_out/pmc_r5sq1 gpurun_out/pmc_r5sq2 gpurun_out/pmc_r5sq3 gpurun_out/pmc_r5sq4
fi
if [[ $part == *c* ]]; then
echo "== the other configurations (818-FS) with HBM-side traffic"
B2="python bench.py --no-cpu --no-host --no-hbm-regime"
$B2 --steps 2 --warmup 1 --grid 512,512,256 --starts 111 --nstarts 8 > $O/g512_818.json 2> $O/g512_818.err; echo "g512 rc $?"
timeout -k 10 500 $B2 --steps 1 --warmup 1 --grid 1024,1024,512 --starts 111 --nstarts 14 > $O/g1024_818.json 2> $O/g1024_818.err; echo "g1024 rc $?"
fi
if [[ $part == *d* ]]; then
echo "== BASELINE.md leg B2 in a bench line (start-1 to convergence on one host core: about 190 s)"
timeout -k 10 900 $B --steps 3 --warmup 1 --nstarts 1 --cpu-b2 > $O/cpu_b2_line.json 2> $O/cpu_b2.err; echo "b2 rc $?"
fi
echo "== summary"
for f in $O/*line.json $O/g*_818.json; do
python3 - "$f" <<'PY'
import json,sys,os
f=sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d["roofline"]; e=d.get("end_to_end_host_program") or {}
    print(os.path.basename(f).ljust(28), "ms %.2f"%d["ms_per_step"], r["bound"], "frac %.3f"%r["frac"], "over solve", r.get("frac_over_solve"), "launch_ms %.4f"%r["avg_launch_ms"], "n", r["launches"], "traffic", r.get("traffic"), "fallbacks", d["config"].get("fallbacks"), "eq %.2f"%d["config"]["full_sweep_equivalents_per_start_mean"], "loop_s", e.get("sweep_loop_wall_seconds"))
    h=d.get("roofline_hbm_regime")
    if h: print("   hbm regime: frac %.3f over solve %s ms %.1f launch_ms %.4f traffic %s cpu %s"%(h["frac"],h.get("frac_over_solve"),h["ms_per_solve"],h["avg_launch_ms"],h.get("traffic"),h.get("cpu_baseline",{}).get("value")))
    b=d.get("cpu_baseline_b2")
    if b: print("   B2:", b["seconds"], "s", b["sweeps_incl_confirming"], "sweeps, digest matches:", b["matches_reference_digest"])
except Exception as ex: print(os.path.basename(f), "FAILED", ex)
PY
done
grep -h "one-launch solve\|column prof" $O/*.err $O/col_profile.txt 2>/dev/null | head
cat $O/six_full_sweep.txt 2>/dev/null | grep -v amdgpu.ids
cat $O/strip_phases.txt 2>/dev/null
