#!/bin/bash
# Conv tests, then layer times.  usage: bash tools/gpu_layers.sh NAME "layers" [pytest -k expr]
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_conv.py tests/test_gpu_models.py -m gpu -x -q --timeout 300 --timeout-method thread ${3:+-k "$3"} > $O/pytest.log 2>&1 || { echo "pytest failed"; tail -60 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
timeout -k 10 200 python tools/layer_bench.py $2 > $O/layers.log 2>&1 || { echo "layer_bench failed"; tail -5 $O/layers.log; exit 1; }
grep conv $O/layers.log
