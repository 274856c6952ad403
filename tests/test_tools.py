"""The profile digest tools on synthetic inputs (no GPU): the timed-region statistics keep exactly the launches between
bench.py's two marks, and the roofline table reproduces its figures from the three committed files."""
import csv
import io
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for name, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", name, s, e])


def test_region_stats_keeps_the_timed_region_only(tmp_path):
    mark = "mpc_profile_mark_kernel(int, int*)"
    rows = [("conv(Params)", 0, 500),                      # warm-up: slower launches that must not count
            (mark, 600, 601),
            ("conv(Params)", 700, 800), ("deconv(Params)", 820, 850), ("conv(Params)", 900, 1020), ("deconv(Params)", 1030, 1050),
            (mark, 1100, 1101),
            ("conv(Params)", 1200, 2200)]                  # the pass behind the timed region
    p = tmp_path / "t_kernel_trace.csv"
    _trace(p, rows)
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "region_stats.py"), str(p), "2"], capture_output=True, text=True, check=True).stdout
    got = {r["Name"]: r for r in csv.DictReader(io.StringIO(out)) if r.get("Calls") and r["Calls"].isdigit()}
    assert got["conv(Params)"]["Calls"] == "2" and float(got["conv(Params)"]["AverageNs"]) == 110.0
    assert got["deconv(Params)"]["Calls"] == "2" and got["deconv(Params)"]["MinNs"] == "20"
    assert got["(wall time between the two marks)"]["TotalDurationNs"] == str(1100 - 601)
    assert mark not in got
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "region_stats.py"), str(p)], capture_output=True, text=True)
    assert r.returncode == 0
    _trace(p, rows[:1])
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "region_stats.py"), str(p)], capture_output=True, text=True)
    assert r.returncode != 0 and "fewer than two" in r.stderr


def test_committed_roofline_table_follows_from_the_committed_profiles():
    """profiles/r3_roofline.csv is what tools/roofline_table.py makes of r3_bench_under_trace.json,
    r3_kernel_stats_timed.csv and r3_pmc_counters.csv; spot-check the dominant kernel's row by hand."""
    prof = os.path.join(REPO, "profiles")
    need = [os.path.join(prof, n) for n in ("r3_roofline.csv", "r3_bench_under_trace.json", "r3_kernel_stats_timed.csv", "r3_pmc_counters.csv")]
    assert all(os.path.exists(n) for n in need)
    line = json.load(open(need[1]))
    dom = line["roofline"]["kernel"]
    k = line["kernels"][dom]
    gflop = k["tflops"] * k["avg_ms"]
    timed = next(r for r in csv.DictReader(open(need[2], newline="")) if dom in r["Name"])
    ms = float(timed["AverageNs"]) / 1e6
    row = next(r for r in csv.DictReader(open(need[0], newline="")) if r["kernel"] == dom)
    assert abs(float(row["achieved_tflops"]) - gflop / ms) < 0.06
    assert abs(float(row["frac"]) - gflop / ms / 2500.0) < 1e-3
    pmc = next(r for r in csv.DictReader(open(need[3], newline="")) if dom in r["Kernel"])
    busy = float(pmc["SQ_VALU_MFMA_BUSY_CYCLES"]) / (float(pmc["GRBM_GUI_ACTIVE"]) / 8 * 1024)
    assert abs(float(row["mfma_busy_under_counters"]) - busy) < 1e-3
