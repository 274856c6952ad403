#!/bin/bash
# Round-2 GPU pass c: statistics kernels, the reworked bench (configs 3, 1, 2, 4, 5), a 1-rank "multi-GPU" launch.
set -o pipefail
O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_stats.py tests/test_gpu_models.py -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { echo "pytest failed"; tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python bench.py > $O/bench3.json 2> $O/bench3.err || { echo "bench 3 failed"; tail -20 $O/bench3.err; exit 1; }
cat $O/bench3.json
for c in 1 2 4 5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/bench$c.json 2> $O/bench$c.err || { echo "bench $c failed"; tail -20 $O/bench$c.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/bench$c.json"))
print($c, d["metric"], round(d["value"],1), "MP/s", round(d["ms_per_step"],3), "ms/step", d["config"]["workload"][:60], "| dom", d["roofline"]["kernel"], round(d["roofline"]["achieved"],1), "| cdf", round(d.get("cdf_build",{}).get("achieved",0)), "| hbm", d["hbm_ceiling"], "| psnr", round(d["stats"]["psnr"],2), "msssim", round(d["stats"]["msssim"],4))
PY
done
timeout -k 10 300 python bench.py --config 5 --arch TestolinaManyPriors --no-cpu-baseline --steps 5 > $O/bench5t.json 2> $O/bench5t.err || { echo "bench 5t failed"; tail -20 $O/bench5t.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench5t.json')); print('5t', round(d['value'],1), d['ms_per_step'])"
