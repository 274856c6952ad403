#!/usr/bin/env python3
"""Roofline table of the transform kernels from the committed profiles alone.
usage: python tools/roofline_table.py [tag, default r3]   -> profiles/<tag>_roofline.csv (also printed)

Inputs, all under profiles/: <tag>_bench_under_trace.json (bench.py's line of the traced run: algorithmic FLOP per launch
of every kernel = its TFLOP/s x its average duration), <tag>_kernel_stats_timed.csv (average duration of every kernel
INSIDE the timed region of that run, from the rocprofv3 kernel trace), <tag>_pmc_counters.csv (one --pmc pass per
counter group).  Per kernel: achieved = FLOP / timed-region duration; frac = achieved / the dense MFMA peak of the
operand type it issues (fp16 2 500 TFLOP/s; a split-precision kernel issues three MFMAs per product, so issued = 3 x
achieved); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) and clock = GRBM_GUI_ACTIVE /
8 / duration, both under the counters (kernels run one at a time there and the clock is lower than in the bench)."""
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
prof = os.path.join(REPO, "profiles")
line = json.load(open(os.path.join(prof, f"{tag}_bench_under_trace.json")))
timed = {r["Name"]: r for r in csv.DictReader(open(os.path.join(prof, f"{tag}_kernel_stats_timed.csv"), newline=""))}
pmc = list(csv.DictReader(open(os.path.join(prof, f"{tag}_pmc_counters.csv"), newline="")))
rows = [["kernel", "launches_per_step", "gflop_per_launch", "timed_region_avg_ms", "achieved_tflops", "peak_tflops", "frac",
         "mfma_per_product", "issued_tflops", "mfma_busy_under_counters", "clock_ghz_under_counters", "hbm_traffic_mb_per_launch"]]
for name, k in line["kernels"].items():
    if "tflops" not in k:
        continue
    gflop = k["tflops"] * k["avg_ms"]                        # TFLOP/s x ms = GFLOP per launch
    t = next((r for n, r in timed.items() if name in n or name.replace("_kernel", "") in n), None)
    c = next((r for r in pmc if name in r["Kernel"] or name.replace("_kernel", "") in r["Kernel"]), None)
    if t is None:
        continue
    ms = float(t["AverageNs"]) / 1e6
    peak, passes = bench.kernel_peak(name)
    ach = gflop / ms
    busy = clock = traffic = ""
    if c and c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
        cyc = float(c["GRBM_GUI_ACTIVE"]) / bench.NUM_XCD
        busy = f"{float(c['SQ_VALU_MFMA_BUSY_CYCLES']) / (cyc * bench.NUM_SIMD):.3f}"
        clock = f"{cyc / float(c['AvgNs']):.2f}"
    if c and c.get("FETCH_SIZE") and c.get("WRITE_SIZE"):
        traffic = f"{(2.0 * float(c['FETCH_SIZE']) + float(c['WRITE_SIZE'])) / 1024.0:.0f}"
    rows.append([name, f"{k['calls_per_step']:.0f}", f"{gflop:.1f}", f"{ms:.3f}", f"{ach:.1f}", f"{peak:.0f}", f"{ach / peak:.3f}", passes,
                 f"{ach * passes:.1f}", busy, clock, traffic])
with open(os.path.join(prof, f"{tag}_roofline.csv"), "w", newline="") as f:
    csv.writer(f).writerows(rows)
for r in rows:
    print(",".join(str(v) for v in r))
