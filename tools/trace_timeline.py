"""Prints the last N kernels of a rocprofv3 kernel trace as a timeline: start, duration, queue, name.
usage: python tools/trace_timeline.py <dir with *kernel_trace.csv> [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-n:]
t0 = int(tail[0]["Start_Timestamp"])
queues = {}
busy_until = 0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = queues.setdefault(r["Queue_Id"], len(queues))
    gap = (s - busy_until) / 1e3 if busy_until else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  q{q}  idle-before {max(gap, 0):7.1f}  {r['Kernel_Name'][:60]}")
    busy_until = max(busy_until, e)
