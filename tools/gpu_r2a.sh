#!/bin/bash
# Round-2 first GPU pass: tests at HEAD, per-layer times of the default build and of the block-major K walk, bench.
set -o pipefail
O=gpurun_out/r2a; mkdir -p $O
KBM=$PWD/build_ab/libmpc_kbm.so
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { echo "pytest failed"; tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 180 python tools/layer_bench.py > $O/layers_default.log 2>&1 || { echo "layer_bench default failed"; tail -5 $O/layers_default.log; exit 1; }
cat $O/layers_default.log
MPC_LIB=$KBM timeout -k 10 180 python tools/layer_bench.py conv2 conv3 conv4 > $O/layers_kbm.log 2>&1 || { echo "layer_bench kbm failed"; tail -5 $O/layers_kbm.log; exit 1; }
cat $O/layers_kbm.log
MPC_LIB=$KBM timeout -k 10 300 python -u -m pytest tests/test_gpu_models.py tests/test_gpu_conv.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest_kbm.log 2>&1 || { echo "pytest kbm failed"; tail -20 $O/pytest_kbm.log; exit 1; }
tail -2 $O/pytest_kbm.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
MPC_LIB=$KBM timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_kbm.json 2> $O/bench_kbm.err || { echo "bench kbm failed"; tail -5 $O/bench_kbm.err; exit 1; }
cat $O/bench_kbm.json
