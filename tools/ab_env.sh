#!/bin/bash
# A/B of whole-bench throughput under environment settings on ONE box, alternating:
#   bash tools/ab_env.sh NAME "tagA:VAR=1 VAR2=x" "tagB:VAR=0" [...]
set -o pipefail
O=gpurun_out/$1; mkdir -p $O; shift
for round in 1 2 3; do
  for spec in "$@"; do
    tag=${spec%%:*}; vars=${spec#*:}
    env $vars timeout -k 10 200 python bench.py --no-cpu-baseline > $O/${tag}_$round.json 2> $O/${tag}_$round.err || { echo "$tag failed"; tail -5 $O/${tag}_$round.err; exit 1; }
    python3 -c "import json,sys; d=json.load(open('$O/${tag}_$round.json')); print('$tag round $round', round(d['value'],1),'MP/s',round(d['ms_per_step'],3),'ms')"
  done
done
