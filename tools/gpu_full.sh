#!/bin/bash
# Full GPU pass: every -m gpu test, smoke(), the default bench line.  usage: bash tools/gpu_full.sh NAME
set -o pipefail
O=gpurun_out/${1:-full}; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { echo "pytest failed"; tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value'],1),'MP/s', round(d['ms_per_step'],3),'ms'); print(d['roofline']); print(d['cpu_baseline'])"
