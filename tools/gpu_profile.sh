#!/bin/bash
# Round-2 profiles at the current tree: kernel trace + stats of the default bench, then the counter passes.
set -o pipefail
O=gpurun_out/${1:-r2p}; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1 || { echo "trace failed"; tail -5 $O/trace.log; exit 1; }
ls $O/trace
bash tools/pmc_run.sh ${1:-r2p}_pmc || { echo "pmc failed"; exit 1; }
head -c 1500 gpurun_out/${1:-r2p}_pmc/pmc.csv
