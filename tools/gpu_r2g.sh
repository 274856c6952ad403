#!/bin/bash
# Round-2 GPU pass g: the LDS-patch f16x3 kernel -- correctness first, then per-layer times against the old kernel.
set -o pipefail
O=gpurun_out/r2g; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_conv.py tests/test_gpu_models.py tests/test_gpu_configs.py tests/test_gpu_full_size.py -x -q -s --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { echo "pytest failed"; tail -60 $O/pytest.log; exit 1; }
grep -E "differ|passed|failed" $O/pytest.log | tail -30
for v in 0 16 8; do
  MPC_X3_PATCH=$v timeout -k 10 120 python tools/layer_bench.py conv2 conv3 conv4 > $O/layers_$v.log 2>&1 || { echo "layer_bench $v failed"; tail -5 $O/layers_$v.log; exit 1; }
  echo "MPC_X3_PATCH=$v"; grep conv $O/layers_$v.log
done
