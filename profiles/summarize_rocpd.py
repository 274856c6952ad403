#!/usr/bin/env python3
"""Turns a rocprofv3 (ROCm 7.2) results database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`
writes DIR/NAME_results.db) into the per-kernel statistics table that is committed under profiles/.
usage: python profiles/summarize_rocpd.py gpurun_out/prof_r1/r1_results.db > profiles/r1_kernel_stats.csv"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
    "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size) from kernels group by name "
    "order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,AGPRs,LDS,Scratch")
for name, calls, tot, avg, mn, mx, vg, ag, lds, scr in rows:
    print(f"\"{name}\",{calls},{tot},{avg:.0f},{mn},{mx},{100.0 * tot / total:.2f},{vg},{ag},{lds},{scr}")
