#!/bin/bash
# A/B layer timings of library builds on ONE box, alternating: bash tools/ab.sh NAME "layers" libA.so libB.so [...]
set -o pipefail
O=gpurun_out/$1; mkdir -p $O; L="$2"; shift 2
for round in 1 2 3; do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    MPC_LIB=$PWD/$lib timeout -k 10 200 python tools/layer_bench.py $L --reps 30 > $O/${tag}_$round.log 2>&1 || { echo "$tag failed"; tail -5 $O/${tag}_$round.log; exit 1; }
    echo "== $tag round $round"; grep conv $O/${tag}_$round.log
  done
done
