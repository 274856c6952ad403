#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED REGION of a profiled bench.py run.

usage: python tools/region_stats.py <rocprofv3 ..._kernel_trace.csv> [timed steps] [> profiles/rN_kernel_stats_timed.csv]

bench.py launches a one-thread kernel (mpc_profile_mark_kernel) right before its timed region starts and right after it
ends.  This script keeps the launches that start between the first two such marks of the trace -- so the warm-up steps
(ramping clocks) and the one-batch-at-a-time pass that follows the timed region (per-kernel HIP events, an idle GPU
between phases) are left out -- and prints what `rocprofv3 --stats` prints for the whole run: calls, total, average,
minimum and maximum duration and the share of the region's kernel time, one row per kernel, plus a last row with the
region's wall time between the marks."""
import csv
import sys
from collections import defaultdict

MARK = "mpc_profile_mark_kernel"


def main(path, steps=0):
    rows = list(csv.DictReader(open(path, newline="")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [r for r in rows if MARK in r["Kernel_Name"]]
    if len(marks) < 2:
        sys.exit(f"{path}: fewer than two {MARK} launches -- not a trace of bench.py's timed region")
    t_begin, t_end = int(marks[0]["End_Timestamp"]), int(marks[1]["Start_Timestamp"])
    per = defaultdict(list)
    for r in rows:
        s = int(r["Start_Timestamp"])
        if t_begin <= s < t_end and MARK not in r["Kernel_Name"]:
            per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - s)
    total = sum(sum(v) for v in per.values())
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for k in sorted(per, key=lambda k: -sum(per[k])):
        v = per[k]
        w.writerow([k, len(v), sum(v), f"{sum(v) / len(v):.1f}", min(v), max(v), f"{100.0 * sum(v) / total:.2f}"])
    w.writerow(["(wall time between the two marks)", 1, t_end - t_begin, t_end - t_begin, t_end - t_begin, t_end - t_begin, ""])
    # a kernel launched several times per step (the same layer kernel for different layers): the average per position in the step
    if steps:
        for k in sorted(per, key=lambda k: -sum(per[k])):
            v = per[k]
            m = len(v) // steps
            if m > 1 and len(v) == m * steps:
                w.writerow([k + " -- average by position in the step"] + [f"{sum(v[i::m]) / steps:.1f}" for i in range(m)])


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
