#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 counter-collection CSVs (`rocprofv3 --pmc ... --output-format csv`).
usage: python profiles/summarize_pmc.py X_counter_collection.csv [more.csv ...] > profiles/NAME.csv
One row per kernel: dispatches, average duration and the per-dispatch average of every counter found."""
import csv
import sys
from collections import defaultdict

vals = defaultdict(lambda: defaultdict(float))     # kernel -> counter -> sum
disp = defaultdict(set)
dur = defaultdict(float)
counters = []
for path in sys.argv[1:]:
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k, c = r["Kernel_Name"], r["Counter_Name"]
            key = (path, r["Dispatch_Id"])
            if key not in disp[k]:
                disp[k].add(key)
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            vals[k][c] += float(r["Counter_Value"])
            if c not in counters:
                counters.append(c)
ndisp = {k: len({d for d in v}) for k, v in disp.items()}
# a kernel appears once per input file; counters from different files are averaged over that file's dispatches
per_file = defaultdict(lambda: defaultdict(int))
for k, v in disp.items():
    for path, _ in v:
        per_file[k][path] += 1
print(",".join(["Kernel", "Dispatches", "AvgNs"] + counters))
for k in sorted(vals, key=lambda k: -dur[k]):
    n = max(per_file[k].values())
    row = [f'"{k}"', str(n), f"{dur[k] / ndisp[k]:.0f}"]
    for c in counters:
        row.append(f"{vals[k][c] / n:.0f}" if c in vals[k] else "")
    print(",".join(row))
