#!/usr/bin/env python3
"""Launch-by-launch picture of a TILE solve from a rocprofv3 kernel trace
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --star six ...):
durations of the tile kernel by size class, gaps between consecutive launches, time per ordering
sweep.  python tools/exp/trace_six.py DIR [hyperplanes_per_sweep]"""
import csv
import glob
import statistics
import sys

d = sys.argv[1]
per_sweep = int(sys.argv[2]) if len(sys.argv) > 2 else 270
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tile = [r for r in rows if "tile_six_kernel" in r["Kernel_Name"] or "tile_sweep_kernel" in r["Kernel_Name"]]
print("tile launches", len(tile))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tile]
gap = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(tile, tile[1:])]
print("kernel us: total %.0f  mean %.2f  median %.2f  min %.2f  max %.1f" % (sum(dur), statistics.mean(dur), statistics.median(dur), min(dur), max(dur)))
print("gap us   : total %.0f  mean %.2f  median %.2f  p90 %.2f" % (sum(gap), statistics.mean(gap), statistics.median(gap), sorted(gap)[int(0.9 * len(gap))]))
edges = [0, 3, 5, 8, 12, 20, 40, 80, 160, 1e9]
for lo, hi in zip(edges, edges[1:]):
    sel = [x for x in dur if lo <= x < hi]
    print("  %6.0f .. %6.0f us: %5d launches, %8.1f ms" % (lo, min(hi, 99999), len(sel), sum(sel) / 1e3))
# per solve: the last solve's launches, by ordering sweep
nsweeps = len(tile) // per_sweep
print("sweeps (all solves):", nsweeps)
last = tile[-(len(tile) % (10**9)):]
for k in range(max(0, nsweeps - 20), nsweeps):
    blk = tile[k * per_sweep:(k + 1) * per_sweep]
    t = (int(blk[-1]["End_Timestamp"]) - int(blk[0]["Start_Timestamp"])) / 1e6
    kd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in blk]
    print("  sweep %3d: %7.2f ms wall, kernels %7.2f ms, launches < 8 us: %3d, median %.1f us" % (k, t, sum(kd) / 1e3, sum(1 for x in kd if x < 8), statistics.median(kd)))
