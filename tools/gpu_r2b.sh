#!/bin/bash
# Round-2 GPU pass b: new parity tests (configs at full size), crafted-header tests, the allocation-free decode.
set -o pipefail
O=gpurun_out/r2b; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v -s --timeout 120 --timeout-method thread > $O/noalloc_alone.log 2>&1 || { echo "noalloc failed"; tail -30 $O/noalloc_alone.log; exit 1; }
tail -3 $O/noalloc_alone.log
timeout -k 10 900 python -u -m pytest tests/test_gpu_configs.py -x -v -s --timeout 300 --timeout-method thread > $O/configs.log 2>&1 || { echo "configs failed"; tail -60 $O/configs.log; exit 1; }
grep -E "differ|error|MRS1|tables|PASSED|FAILED|passed|failed" $O/configs.log
timeout -k 10 600 python -u -m pytest tests tests/wip_no_alloc.py -m gpu -x -q -s --timeout 300 --timeout-method thread > $O/pytest_all.log 2>&1 || { echo "pytest all failed"; tail -40 $O/pytest_all.log; exit 1; }
tail -3 $O/pytest_all.log
