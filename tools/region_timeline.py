#!/usr/bin/env python3
"""Timeline of ONE step from the middle of bench.py's timed region (between its two marks) of a rocprofv3 kernel trace:
start, duration, queue, how long no kernel at all had been running before it, name.
usage: python tools/region_timeline.py <..._kernel_trace.csv> [milliseconds to show, default 8.5]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1], newline="")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 8.5e6
marks = [r for r in rows if "mpc_profile_mark_kernel" in r["Kernel_Name"]]
t_begin, t_end = int(marks[0]["End_Timestamp"]), int(marks[1]["Start_Timestamp"])
mid = (t_begin + t_end) // 2
# start the window at the first conv1 launch after the middle
first = next(int(r["Start_Timestamp"]) for r in rows if int(r["Start_Timestamp"]) >= mid and "conv1_rgb" in r["Kernel_Name"])
queues, busy_until, idle_total = {}, 0, 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < first - 2e6 or s > first + span:
        if s < first:
            busy_until = max(busy_until, e)
        continue
    gap = max((s - busy_until) / 1e3, 0.0) if busy_until else 0.0
    if s >= first:
        q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
        idle_total += gap
        print(f"{(s - first) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q{q}  idle-before {gap:6.1f}  {r['Kernel_Name'][:70]}")
    busy_until = max(busy_until, e)
print(f"GPU idle (no kernel running) inside the window: {idle_total:.1f} us")
