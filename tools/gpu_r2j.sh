#!/bin/bash
# Round-2 GPU pass j: patch kernel with LDS-DMA weight ring -- correctness, then layer times.
set -o pipefail
O=gpurun_out/r2j; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_conv.py tests/test_gpu_models.py tests/test_gpu_configs.py tests/test_gpu_full_size.py -x -q -s --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { echo "pytest failed"; tail -60 $O/pytest.log; exit 1; }
grep -E "symbols differ|passed|failed" $O/pytest.log | tail -14
timeout -k 10 120 python tools/layer_bench.py conv2 conv3 conv4 > $O/layers.log 2>&1 || { echo "layer_bench failed"; tail -5 $O/layers.log; exit 1; }
grep conv $O/layers.log
MPC_X3_PATCH=0 timeout -k 10 120 python tools/layer_bench.py conv2 conv3 conv4 > $O/layers_old.log 2>&1 || exit 1
grep conv $O/layers_old.log
