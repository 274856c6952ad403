#!/bin/bash
# Bench lines of every configuration (+ the smooth-data variant of config 3, + the Testolina decoder for config 5).
set -o pipefail
O=gpurun_out/${1:-bench_all}; mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }; python - <<PY
import json; d=json.load(open("$O/$name.json")); print("$name", round(d["value"],1), d["unit"], round(d["ms_per_step"],3), "ms/step", d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
PY
}
run config3 && run config3_serial --serial --no-cpu-baseline && run config3_smooth --data smooth --no-cpu-baseline && run config1 --config 1 --no-cpu-baseline --steps 50 && run config2 --config 2 --no-cpu-baseline --steps 50 \
 && run config4 --config 4 --no-cpu-baseline && run config5 --config 5 --no-cpu-baseline && run config5_testolina --config 5 --arch TestolinaManyPriors --no-cpu-baseline \
 && run config6_hyperprior --config 6 --steps 20 --no-cpu-baseline && run config6_hyperprior_serial --config 6 --steps 10 --serial --no-cpu-baseline \
 && run config6_balle2018 --config 6 --arch Balle2018PTTFExp --steps 20 --no-cpu-baseline
