#!/bin/bash
# Kernel trace of the default bench, digested to the timed region only.  usage: bash tools/gpu_trace.sh NAME [bench args]
set -o pipefail
N=${1:-tr}; shift
O=gpurun_out/$N; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o trace -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline "$@" > $O/trace.log 2>&1 || { echo "trace failed"; tail -5 $O/trace.log; exit 1; }
python3 tools/region_stats.py $O/trace/trace_kernel_trace.csv 50 > $O/kernel_stats_timed.csv || { echo "region stats failed"; exit 1; }
rm -f $O/trace/trace_kernel_trace.csv
cut -c1-60,200- $O/kernel_stats_timed.csv | head -14
grep '^{' $O/trace.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'MP/s',round(d['ms_per_step'],3),'ms')"
