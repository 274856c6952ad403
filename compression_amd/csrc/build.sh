#!/bin/bash
# Builds compression_amd/libmpc.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="${MPC_OUT:-$here/../libmpc.so}"        # MPC_OUT / MPC_BUILD_DIR: side-by-side experimental builds
bdir="${MPC_BUILD_DIR:-$here/build}"
srcs=(context.hip entropy_tables.hip prior_competition.hip rans.hip conv_igemm.hip conv_igemm_f16.hip deconv5s2_patch_f16.hip conv_igemm_x3.hip conv5s2_patch_x3.hip conv_rgb_out_f16.hip gdn.hip metrics.hip)
objs=()
mkdir -p "$bdir"
pids=()
for s in "${srcs[@]}"; do
  o="$bdir/${s%.hip}.o"
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$here/$s" -nt "$o" ] || [ "$here/mpc_internal.h" -nt "$o" ] || [ "$here/conv_common.h" -nt "$o" ] || [ "$here/conv_x3_epilogue.h" -nt "$o" ] || [ "$here/../../include/mpc.h" -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
        -c "$here/$s" -o "$o" ${MPC_EXTRA_FLAGS:-} &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
