#!/usr/bin/env python3
"""Times the first analysis layer (24 x 768x512) alone and beside the rANS decoder of a 24-image batch running on
another stream -- the pairing of the codec pipeline -- with HIP events round the layer.
usage: [MPC_LIB=other/libmpc.so] python tools/corun_decoder.py [reps] [occupier]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compression_amd as ca  # noqa: E402
from compression_amd import ops  # noqa: E402
from tests import synth  # noqa: E402

import ctypes  # noqa: E402
from compression_amd import _lib  # noqa: E402
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
occupier = len(sys.argv) > 2 and sys.argv[2] == "occupier"
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
m = ca.init_model("ManyPriors", dev, out_channel_N=192, out_channel_M=256, num_distributions=64, analysis_precision="f16x3",
                  synthesis_precision="fp16")
m.load_state_dict(synth.manypriors_state_dict(234))
m.eval()
x = synth.rand_image(1000, 24, 512, 768).to(dev)
strings, idx, h, w = m.encode_batch(x)
x4 = ops.nchw_to_nhwc(x, 4)
conv1 = m.Encoder.layers()[0]
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
sink = torch.zeros(16, device=dev)


def run(beside):
    best = []
    for _ in range(reps):
        torch.cuda.synchronize()
        go = torch.cuda.Event()
        go.record()
        if beside:
            with torch.cuda.stream(sa):
                sa.wait_event(go)
                if occupier:                                # (a -DMPC_EXPERIMENT_OCCUPIER build: a sleeping kernel of the decoder's footprint)
                    ctx = _lib.context_for(dev)
                    ctx.check(ctx.lib.mpc_experiment_occupier(ctx.handle, 660, 18484, 370, ctypes.c_void_p(sink.data_ptr())))
                else:
                    job = m.decode_batch_begin(strings, idx, h, w)
        with torch.cuda.stream(sb):
            sb.wait_event(go)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            conv1(x4, out_split=True)
            e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1))
    best.sort()
    return best[len(best) // 2]


for _ in range(3):
    conv1(x4, out_split=True)
dump = os.environ.get("MPC_DUMP_PREFIX")                    # a -DMPC_STAMPS build: one dump of the layer's workgroups per mode
if dump:
    os.environ["MPC_DUMP"] = dump + "_alone.txt"
alone = run(False)
if dump:
    os.environ["MPC_DUMP"] = dump + "_beside.txt"
beside = run(True)
print(f"conv1 alone {alone:.3f} ms, beside the decoder {beside:.3f} ms  ({os.environ.get('MPC_LIB', 'libmpc.so')})")
