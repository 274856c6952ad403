#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of a GPU pass (tools/gpu_profile.sh) into profiles/ (tracked) and records what they
were taken on.
usage: python tools/save_profiles.py gpurun_out/r3p gpurun_out/r3p_pmc [round tag, default r3]
Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --stats of the whole run), <tag>_kernel_stats_timed.csv (the same
statistics restricted to bench.py's timed region, tools/region_stats.py), <tag>_bench_under_trace.json (the bench line
of that traced run), <tag>_pmc_counters.csv, <tag>_pmc_meta.json (commit, digest of the dominant kernel's sources --
bench.py reports roofline.traffic only while that digest matches -- and the check of the FETCH_SIZE x2 rule on a kernel
whose read bytes are known exactly) and <tag>_roofline.csv (tools/roofline_table.py: per transform kernel, the
algorithmic FLOP per launch, the timed region's average duration, TFLOP/s against the MFMA peak of its operand type, and
matrix-pipe busy share and clock from the counter pass)."""
import csv
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

trace_dir, pmc_dir = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "r3"
prof = os.path.join(REPO, "profiles")
shutil.copy(os.path.join(trace_dir, "trace", "trace_kernel_stats.csv"), os.path.join(prof, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(pmc_dir, "pmc.csv"), os.path.join(prof, f"{tag}_pmc_counters.csv"))
for name in ("kernel_stats_timed.csv", "bench_under_trace.json"):
    if os.path.exists(os.path.join(trace_dir, name)):
        shutil.copy(os.path.join(trace_dir, name), os.path.join(prof, f"{tag}_{name}"))
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "status", "--porcelain", "--", "compression_amd/csrc"], cwd=REPO, capture_output=True, text=True).stdout.strip()
meta = {"commit": commit + ("+uncommitted kernel changes" if dirty else ""), "sources": list(bench.TRAFFIC_SOURCES),
        "sources_sha256_16": bench.sources_digest(),
        "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (one rocprofv3 --pmc pass per counter group, tools/pmc_run.sh)"}
# clamp_sse reads two fp32 tensors of 24 x 3 x 512 x 768 elements with 16-byte loads and nothing else
known = 2 * 24 * 3 * 512 * 768 * 4
for r in csv.DictReader(open(os.path.join(prof, f"{tag}_pmc_counters.csv"))):
    if r["Kernel"].startswith("clamp_sse_kernel") and r.get("FETCH_SIZE"):
        kb = float(r["FETCH_SIZE"])
        meta["fetch_size_check"] = {"kernel": "clamp_sse_kernel", "bytes_read": known, "FETCH_SIZE_KB": kb,
                                    "bytes_per_reported_byte": known / (kb * 1024.0)}
json.dump(meta, open(os.path.join(prof, f"{tag}_pmc_meta.json"), "w"), indent=1)
print(json.dumps(meta, indent=1))
if os.path.exists(os.path.join(prof, f"{tag}_kernel_stats_timed.csv")):
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "roofline_table.py"), tag], check=True)
