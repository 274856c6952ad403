#!/bin/bash
# Round-2 GPU pass f: sync-free encode/decode path -- all GPU tests, then the bench.
set -o pipefail
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { echo "pytest failed"; tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench3.json 2> $O/bench3.err || { echo "bench 3 failed"; tail -20 $O/bench3.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench3.json"))
print(round(d["value"],1), "MP/s", round(d["ms_per_step"],3), "ms/step")
print({k:(round(v["avg_ms"]*v["calls_per_step"],3)) for k,v in d["kernels"].items()})
print("kernel sum", round(sum(v["avg_ms"]*v["calls_per_step"] for k,v in d["kernels"].items() if k!="gather_cdf"),3))
PY
