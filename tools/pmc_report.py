#!/usr/bin/env python3
"""Prints a per-kernel digest of a profiles/summarize_pmc.py table (derived ratios of the SQ counters,
effective clock, fabric read/write megabytes).  usage: python tools/pmc_report.py table.csv"""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    g = lambda c: float(r.get(c) or 0)
    wc = g("SQ_WAVE_CYCLES")
    if not wc or float(r["AvgNs"]) < 2e4:
        continue
    print(f"{r['Kernel'][:58]:58s} n={r['Dispatches']:>3} {float(r['AvgNs']) / 1e3:8.1f}us clk={g('GRBM_GUI_ACTIVE') / 8 / float(r['AvgNs']):.2f}GHz "
          f"wait={g('SQ_WAIT_ANY') / wc:.2f} stall={g('SQ_WAIT_INST_ANY') / wc:.2f} act={g('SQ_ACTIVE_INST_ANY') / wc:.2f} "
          f"ldsstall={g('SQ_WAIT_INST_LDS') / wc:.2f} mfma_cyc={g('SQ_VALU_MFMA_BUSY_CYCLES') / 1e6:.0f}M bankconf={g('SQ_LDS_BANK_CONFLICT') / 1e6:.1f}M "
          f"busy={g('SQ_BUSY_CYCLES') / 1e6:.1f}M fetch={g('FETCH_SIZE') / 1e3:.0f}MB write={g('WRITE_SIZE') / 1e3:.0f}MB")
