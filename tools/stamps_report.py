#!/usr/bin/env python3
"""Digest of an in-kernel timestamp dump (diagnostic build -DMPC_F16_STAMPS, MPC_DUMP=file): per phase, mean microseconds of
prologue / K loop / GDN mix / final math / flush, the K loop's clock, and per CU the idle gaps between workgroups."""
import sys
import collections
rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip()]
rows = [r for r in rows if r[0]]
t0 = min(r[0] for r in rows)
print(f"{len(rows)} workgroups, span {(max(r[5] for r in rows) - t0) / 100:.1f} us")
by_phase = collections.defaultdict(list)
for r in rows:
    by_phase[r[8]].append(r)
for ph, rs in sorted(by_phase.items()):
    n = len(rs)
    seg = [sum(r[i + 1] - r[i] for r in rs) / n / 100 for i in range(5)]
    clk = sum(r[6] / max(r[2] - r[1], 1) for r in rs) / n * 100 / 1000
    if len(rs[0]) > 9:
        steps = {0: 54, 1: 36, 2: 36, 3: 24}.get(ph, 1)
        sg = [sum(r[9 + i] for r in rs) / n / steps for i in range(4)]
        print(f"   per step (cycles of wave 0, {steps} steps): dma issue {sg[0]:.0f}, reads+mfma {sg[1]:.0f}, vmcnt wait {sg[2]:.0f}, barrier {sg[3]:.0f}")
    print(f"phase {ph}: n={n} prologue {seg[0]:.2f} K {seg[1]:.2f} gdn {seg[2]:.2f} math {seg[3]:.2f} flush {seg[4]:.2f} total {sum(seg):.2f} us; K-loop clock {clk:.2f} GHz")
# per CU timeline: key = (xcd = index % 8, hw cu/sh/se bits)
cus = collections.defaultdict(list)
for i, r in enumerate(rows):
    cus[(i % 8, (r[7] >> 8) & 0xff)].append((r[0], r[5]))
gaps, busy = [], []
for k, v in cus.items():
    v.sort()
    for a, b in zip(v, v[1:]):
        gaps.append((b[0] - a[1]) / 100)
    busy.append(sum(e - s for s, e in v) / 100)
gaps.sort()
print(f"{len(cus)} CUs; gap between consecutive workgroups on a CU: median {gaps[len(gaps)//2]:.2f} us, mean {sum(gaps)/len(gaps):.2f} us, p90 {gaps[int(len(gaps)*0.9)]:.2f}")
print(f"mean busy time per CU {sum(busy)/len(busy):.1f} us")
