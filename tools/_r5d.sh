set -o pipefail
O=gpurun_out/r7e; mkdir -p $O
for rep in 1 2; do for v in nocap cap; do
  lib=compression_amd/libmpc.so; [ $v = nocap ] && lib=build_ab/libmpc_nocap.so
  MPC_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 > $O/${v}_$rep.json 2>$O/$v.err || { tail $O/$v.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r7e/${v}_$rep.json').read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['kernels']['conv1_rgb_patch_x3_kernel<6>']['avg_ms'],3))"
done; done
