#!/bin/bash
# Runs the named GPU tests (pytest -k expression) with their output kept.  usage: bash tools/gpu_quick.sh NAME "k-expr" [files...]
set -o pipefail
O=gpurun_out/$1; mkdir -p $O; K="$2"; shift 2
timeout -k 10 600 python -u -m pytest ${@:-tests} -m gpu -x -v -s --timeout 300 --timeout-method thread -k "$K" > $O/pytest.log 2>&1 || { echo "pytest failed"; tail -80 $O/pytest.log; exit 1; }
grep -E "PASSED|FAILED|passed|failed|^\[" $O/pytest.log | tail -40
