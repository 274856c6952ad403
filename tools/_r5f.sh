set -o pipefail
O=gpurun_out/r6m; mkdir -p $O
export TMPDIR=/tmp
MPC_PIPE_MODE=3 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1 || { echo "trace failed"; tail -5 $O/trace.log; exit 1; }
python3 tools/trace_timeline.py $O/trace 700 > $O/timeline.txt
wc -l $O/timeline.txt
