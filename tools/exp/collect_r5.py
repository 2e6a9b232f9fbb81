#!/usr/bin/env python3
"""Copies what tools/exp/r5final.sh left in gpurun_out/r5final/ into profiles/r05_* (run in the
container after the gpurun call)."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(ROOT, "gpurun_out", "r5final")
P = os.path.join(ROOT, "profiles")


def line(name):
    return json.loads(open(os.path.join(O, name)).read().strip().splitlines()[-1])


def text(name):
    return "".join(l for l in open(os.path.join(O, name)) if "amdgpu.ids" not in l)


json.dump(line("bench_line.json"), open(os.path.join(P, "r05_bench_line.json"), "w"), indent=1)
shutil.copy(os.path.join(O, "bench_kernel_stats.csv"), os.path.join(P, "r05_bench_kernel_stats.csv"))
json.dump(line("six1024_bench_line.json"), open(os.path.join(P, "r05_six1024_bench_line.json"), "w"), indent=1)
shutil.copy(os.path.join(O, "six1024_kernel_stats.csv"), os.path.join(P, "r05_six1024_kernel_stats.csv"))
json.dump(line("six512_bench_line.json"), open(os.path.join(P, "r05_six512_bench_line.json"), "w"), indent=1)
open(os.path.join(P, "r05_six_full_sweep.txt"), "w").write(
    "ONE ordering sweep of the column kernel with every tile due: a converged box solved again (tools/exp/one_sweep.py\n"
    "1024,1024,512 14; every column is one run of 16 tiles, nothing improves, nothing is stored; algorithmic = 12 B x cells)\n\n"
    + text("six_full_sweep.txt"))
open(os.path.join(P, "r05_col_profile.txt"), "w").write(
    "column_solve_kernel, -DTTSWEEP_COL_PROFILE build (s_memtime stamps summed over all wavefronts; shares of the wavefronts'\n"
    "resident time; tools/exp/col_probe.py 1024,1024,512 14 2 1 = two solves of six-FS 1024x1024x512 x 14 and the second solve\n"
    "of the converged boxes: one sweep with every tile due).  cycles per block = 16 steps.\n\n" + text("col_profile.txt"))
small = {k: line(f"{k}.json") for k in ("n3_line", "n2_line", "n1_line", "start4_line", "n3_waves4_line", "n1_waves4_line", "n3_handoff3_line")}
small["what"] = ("bench.py --no-cpu --no-traffic --no-host --no-hbm-regime --steps 5 --warmup 1 + --nstarts 3 / 2 / 1, --starts 4 "
                 "(defaults: the eight-wave latency instance up to 3 starts), and for comparison --waves 4 (the four-wave instance: "
                 "round 4's kernel) and --handoff 3 (direct hand-off)  (tools/exp/r5final.sh a)")
json.dump(small, open(os.path.join(P, "r05_small_shard_lines.json"), "w"), indent=1)
stats = {"what": "-DTTSWEEP_ASYNC_STATS build (gpurun_exp/asyncstats.so, bench.py --lib): units and staged planes relaxed by one "
                 "one-launch solve WITH the deferral (default margin), and those of them that improved no cell; one line per solve",
         "24 starts": [l.strip() for l in text("asyncstats.err").splitlines() if "one-launch solve" in l],
         "3 starts": [l.strip() for l in text("asyncstats_n3.err").splitlines() if "one-launch solve" in l]}
json.dump(stats, open(os.path.join(P, "r05_async_stats.json"), "w"), indent=1)
open(os.path.join(P, "r05_strip_phases.txt"), "w").write(
    "Phase stamps of the unit kernel in a one-launch solve (-DTTSWEEP_PROFILE build, gpurun_exp/stripprof.so; tools/exp/r5final.sh a):\n"
    "wave 0, clock64 cycles per unit; fetch = claim incl. the wait for an entry (the workers' idle time); wait = at the barriers between\n"
    "groups of staged planes (of it: for the wave's own loads); 24 starts: four-wave two-plane units, 3 and 1 starts: the eight-wave\n"
    "latency instance (one-plane units, three planes per barrier group).\n\n" + text("strip_phases.txt"))
open(os.path.join(P, "r05_col_batch_size.txt"), "w").write(
    "Column kernel, six-FS, 1024x1024x512: starts resident per launch (tools/exp/col_probe.py 1024,1024,512 N 2 1, second solve; 8 TB/s roof)\n\n"
    + text("col_batch.txt"))
if os.path.exists(os.path.join(O, "g512_818.json")):     # (part c)
  other = {"g512_818": line("g512_818.json"), "g1024_818": line("g1024_818.json"),
         "what": "bench.py --no-cpu --no-host --no-hbm-regime (traffic measured live) --grid 512,512,256 --starts 111 --nstarts 8 --steps 2 --warmup 1 / "
                 "--grid 1024,1024,512 --starts 111 --nstarts 14 --steps 1 --warmup 1, 818-FS (tools/exp/r5final.sh c)"}
  json.dump(other, open(os.path.join(P, "r05_other_config_lines.json"), "w"), indent=1)
if os.path.exists(os.path.join(O, "cpu_b2_line.json")):       # (part d of r5final.sh: 190 s of CPU; not in every regeneration)
    json.dump(line("cpu_b2_line.json"), open(os.path.join(P, "r05_cpu_b2_line.json"), "w"), indent=1)


def counters(name):
    txt = open(os.path.join(O, name)).read().strip().splitlines()[-1]
    d = eval(txt[txt.index("{"):])
    return {k: int(v[0]) for k, v in d.items()}, max(v[1] for v in d.values())


c1, n1 = counters("pmc_sq1.txt")
c2, _ = counters("pmc_sq2.txt")
c3, n3 = counters("pmc_sq3.txt")
c4, _ = counters("pmc_sq4.txt")
u = dict(c1, **c2)
w = dict(c3, **c4)
out = {
    "command": "rocprofv3 --pmc <8 SQ counters per run> --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 "
               "--no-cpu --no-traffic --no-host --no-hbm-regime [...] (tools/exp/pmc.sh, called by tools/exp/r5final.sh b)",
    "sweep_units_kernel": {
        "kernel": f"sweep_units_kernel<16, 2, true> (one launch per solve), sums over all launches of the run ({n1} launches)",
        "counters": u,
        "derived": {
            "wave time waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": round(u["SQ_WAIT_ANY"] / u["SQ_WAVE_CYCLES"], 3),
            "vector ALU busy per SIMD (SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES)": round(u["SQ_INSTS_VALU"] / u["SQ_BUSY_CU_CYCLES"], 3),
            "scalar per vector instruction": round(u["SQ_INSTS_SALU"] / u["SQ_INSTS_VALU"], 3),
            "LDS busy (SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES)": round(u["SQ_LDS_IDX_ACTIVE"] / u["SQ_BUSY_CU_CYCLES"], 3),
            "LDS bank conflict share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)": round(u["SQ_LDS_BANK_CONFLICT"] / u["SQ_LDS_IDX_ACTIVE"], 3),
        },
    },
    "column_solve_kernel": {
        "command": "same, --star six --grid 1024,1024,512 --starts 111 --nstarts 14",
        "kernel": f"column_solve_kernel (one launch per solve, 1024 single-wavefront workgroups), sums over {n3} launches",
        "counters": w,
        "derived": {
            "wave time waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": round(w["SQ_WAIT_ANY"] / w["SQ_WAVE_CYCLES"], 3),
            "wave time issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)": round(w["SQ_ACTIVE_INST_ANY"] / w["SQ_WAVE_CYCLES"], 3),
            "vector ALU busy per SIMD (SQ_INSTS_VALU / SQ_WAVE_CYCLES: one wavefront per SIMD, the cycle counters count 4-cycle units)":
                round(w["SQ_INSTS_VALU"] / w["SQ_WAVE_CYCLES"], 3),
            "scalar per vector instruction": round(w["SQ_INSTS_SALU"] / w["SQ_INSTS_VALU"], 3),
            "LDS busy (SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES)": round(w["SQ_LDS_IDX_ACTIVE"] / w["SQ_BUSY_CU_CYCLES"], 3),
            "LDS bank conflict share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)": round(w["SQ_LDS_BANK_CONFLICT"] / w["SQ_LDS_IDX_ACTIVE"], 3),
            "round 4 for comparison": "waiting 0.47, issuing 0.40, vector ALU 0.24, LDS busy 0.33, bank conflict share 0.49 (profiles/r04_pmc_sq.json)",
        },
    },
}
json.dump(out, open(os.path.join(P, "r05_pmc_sq.json"), "w"), indent=1)
print("profiles/r05_* written")
