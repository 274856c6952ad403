#!/bin/bash
# Profiles of the current tree: kernel trace + stats of the default bench (whole run and timed region only), then the
# counter passes.  usage: bash tools/gpu_profile.sh NAME
set -o pipefail
N=${1:-r3p}
O=gpurun_out/$N; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1 || { echo "trace failed"; tail -5 $O/trace.log; exit 1; }
ls $O/trace
python3 tools/region_stats.py $O/trace/trace_kernel_trace.csv 50 > $O/kernel_stats_timed.csv || { echo "region stats failed"; exit 1; }
head -12 $O/kernel_stats_timed.csv
grep '^{' $O/trace.log | tail -1 > $O/bench_under_trace.json
rm -f $O/trace/trace_kernel_trace.csv          # tens of MB; its digest is kept
bash tools/pmc_run.sh ${N}_pmc || { echo "pmc failed"; exit 1; }
python3 tools/pmc_report.py gpurun_out/${N}_pmc/pmc.csv
