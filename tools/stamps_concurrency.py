#!/usr/bin/env python3
"""From an in-kernel timestamp dump (-DMPC_STAMPS build, MPC_DUMP=file): how many workgroups of the kernel were
resident on a CU at a time (sum of workgroup lifetimes / the CU's active span), the mean lifetime of a workgroup and its
phases.  usage: python tools/stamps_concurrency.py dump.txt"""
import collections
import sys

rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip()]
rows = [r for r in rows if r[0] and r[5]]
cus = collections.defaultdict(list)
for i, r in enumerate(rows):
    cus[(i % 8, (r[7] >> 8) & 0xff)].append((r[0], r[5]))
conc = []
for v in cus.values():
    span = max(e for _, e in v) - min(s for s, _ in v)
    conc.append(sum(e - s for s, e in v) / max(span, 1))
life = sum(r[5] - r[0] for r in rows) / len(rows) / 100
seg = [sum(r[i + 1] - r[i] for r in rows) / len(rows) / 100 for i in range(5)]
t0, t1 = min(r[0] for r in rows), max(r[5] for r in rows)
conc.sort()
print(f"{len(rows)} workgroups on {len(cus)} CUs, kernel span {(t1 - t0) / 100:.1f} us; workgroups resident per CU: mean {sum(conc) / len(conc):.2f}, "
      f"min {conc[0]:.2f}, median {conc[len(conc) // 2]:.2f}, max {conc[-1]:.2f}; lifetime {life:.1f} us = " + " + ".join(f"{s:.1f}" for s in seg))
