#!/usr/bin/env python3
"""Copies what tools/exp/r3final.sh left in gpurun_out/r3final/ into profiles/r03_* (run in the
container after the gpurun call)."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(ROOT, "gpurun_out", "r3final")
P = os.path.join(ROOT, "profiles")


def line(name):
    return json.loads(open(os.path.join(O, name)).read().strip().splitlines()[-1])


json.dump(line("bench_line.json"), open(os.path.join(P, "r03_bench_line.json"), "w"), indent=1)
shutil.copy(os.path.join(O, "bench_kernel_stats.csv"), os.path.join(P, "r03_bench_kernel_stats.csv"))
json.dump(line("six1024_bench_line.json"), open(os.path.join(P, "r03_six1024_bench_line.json"), "w"), indent=1)
shutil.copy(os.path.join(O, "six1024_kernel_stats.csv"), os.path.join(P, "r03_six1024_kernel_stats.csv"))
txt = open(os.path.join(O, "six1024_launch_classes.txt")).read()
txt = txt[:txt.index("sweeps (all solves)")] if "sweeps (all solves)" in txt else txt
open(os.path.join(P, "r03_six1024_launch_classes.txt"), "w").write(
    "tile_six_kernel launches of FOUR solves (warm-up, two timed, one instrumented) of six-FS 1024x1024x512 x 14, by duration\n"
    "(rocprofv3 --kernel-trace, tools/exp/trace_six.py; 5188 launches per solve: the hyperplanes in front of the first\n"
    "due tile of a sweep are not launched; per-sweep wall times of the dealing variants: r03_six1024_dealing.txt)\n\n" + txt)
json.dump(line("six512_bench_line.json"), open(os.path.join(P, "r03_six512_bench_line.json"), "w"), indent=1)
others = {k: line(f"{k}.json") for k in ("n3_line", "start4_line", "818_512_line", "818_1024_line", "prepass146_line",
                                         "prepass98_line", "n3_prepass98_line")}
json.dump(others, open(os.path.join(P, "r03_other_config_lines.json"), "w"), indent=1)
two = {k: line(f"{k}.json") for k in ("gloo2_512_line", "gloo2_241_line")}
two["note"] = ("two ranks on the ONE GPU of the box (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 "
               "--backend gloo): a rehearsal of the sharded path - cost-balanced shards, one context per rank, the host "
               "gather through shared memory - not a scaling measurement (both processes share the GPU).  The same with "
               "--backend nccl cannot run on one GPU (RCCL refuses two ranks on one device).")
json.dump(two, open(os.path.join(P, "r03_two_rank_lines.json"), "w"), indent=1)

# SQ counters
def counters(name):
    txt = open(os.path.join(O, name)).read().strip().splitlines()[-1]
    d = eval(txt[txt.index("{"):])
    return {k: int(v[0]) for k, v in d.items()}, max(v[1] for v in d.values())


c1, n1 = counters("pmc_sq1.txt")
c2, _ = counters("pmc_sq2.txt")
c3, n3 = counters("pmc_sq3.txt")
u = dict(c1, **c2)
out = {
    "command": "rocprofv3 --pmc <8 SQ counters per run> --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 "
               "--no-cpu --no-traffic --no-host --no-hbm-regime (tools/exp/pmc.sh, called by tools/exp/r3final.sh)",
    "kernel": f"sweep_units_kernel<16, 2, true> (one launch per solve), sums over all launches of the run ({n1} launches)",
    "counters": u,
    "derived": {
        "wave time waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": round(u["SQ_WAIT_ANY"] / u["SQ_WAVE_CYCLES"], 3),
        "wave time issuing vector ALU instructions (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)": round(u["SQ_ACTIVE_INST_VALU"] / u["SQ_WAVE_CYCLES"], 3),
        "vector ALU busy per SIMD (4 cycles x SQ_INSTS_VALU / (4 SIMDs x SQ_BUSY_CU_CYCLES))": round(u["SQ_INSTS_VALU"] / u["SQ_BUSY_CU_CYCLES"], 3),
        "scalar per vector instruction": round(u["SQ_INSTS_SALU"] / u["SQ_INSTS_VALU"], 3),
        "scalar memory instructions (round 2: 2.62e9)": u["SQ_INSTS_SMEM"],
        "LDS busy (SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES)": round(u["SQ_LDS_IDX_ACTIVE"] / u["SQ_BUSY_CU_CYCLES"], 3),
        "round 2 for comparison": "waiting 0.34, issuing VALU 0.34, VALU busy 0.66, scalar per vector 0.25, LDS busy 0.15 (profiles/r02_pmc_sq.json)",
    },
    "tile_six_kernel": {
        "command": "same, --star six --grid 1024,1024,512 --starts 111 --nstarts 14",
        "launches": n3, "counters": c3,
        "derived": {
            "wave time waiting": round(c3["SQ_WAIT_ANY"] / c3["SQ_WAVE_CYCLES"], 3),
            "vector ALU busy per SIMD, CU average (SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES)": round(c3["SQ_INSTS_VALU"] / c3["SQ_BUSY_CU_CYCLES"], 3),
            "LDS busy (SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES)": round(c3["SQ_LDS_IDX_ACTIVE"] / c3["SQ_BUSY_CU_CYCLES"], 3),
            "note": "six single-wavefront workgroups sit 2-2-1-1 on a CU's SIMDs (profiles/r03_placeprobe.txt): the CU average hides "
                    "that two SIMDs carry twice the instruction stream of the other two",
        },
    },
}
json.dump(out, open(os.path.join(P, "r03_pmc_sq.json"), "w"), indent=1)
print("profiles/r03_* written")
