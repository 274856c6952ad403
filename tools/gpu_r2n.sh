#!/bin/bash
# Round-2 GPU pass n: all-DMA fp16 kernel -- correctness, layer times, bench.
set -o pipefail
O=gpurun_out/r2n; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_conv.py tests/test_gpu_models.py tests/test_gpu_configs.py tests/test_gpu_full_size.py -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { echo "pytest failed"; tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 120 python tools/layer_bench.py > $O/layers.log 2>&1 || { echo "layer_bench failed"; tail -5 $O/layers.log; exit 1; }
grep conv $O/layers.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value'],1),'MP/s', round(d['ms_per_step'],3),'ms'); print({k:round(v['avg_ms']*v['calls_per_step'],3) for k,v in d['kernels'].items()})"
