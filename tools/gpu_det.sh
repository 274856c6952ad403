#!/bin/bash
# Determinism loop of deconv3 for several library builds on one box: bash tools/gpu_det.sh NAME lib1.so lib2.so ...
O=gpurun_out/$1; mkdir -p $O; shift
for lib in "$@"; do
  tag=$(basename $lib .so)
  MPC_LIB=$PWD/$lib timeout -k 10 120 python tools/dbg_gs.py 4 40 > $O/$tag.log 2>&1 || { echo "$tag failed"; tail -5 $O/$tag.log; exit 1; }
  echo "== $tag: $(tail -1 $O/$tag.log)"; grep differs $O/$tag.log | head -3
done
