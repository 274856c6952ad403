#!/usr/bin/env python3
"""Times the six layers of h_a and h_s (exact f32 implicit GEMM) at the benchmark's hyperprior workload (24 images of
768x512: a 24 x 32 x 48 x 256 latent).  usage: python tools/hyper_layers.py [images]"""
import sys

import torch

sys.path.insert(0, ".")
import compression_amd as ca  # noqa: E402
from tests import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
m = ca.init_model("Balle2018PTTFExp", dev, out_channel_N=192, out_channel_M=256)
sd = synth.transform_weights(237, 192, 256)
sd.update(synth.hyperprior_weights(239, 192, 256))
m.load_state_dict(sd)
m.eval()
y = torch.randn(B, 32, 48, 256, device=dev) * 3


def timed(fn, x, reps=20):
    out = fn(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn(x)
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps


for name, net in (("h_a", m.priorEncoder), ("h_s", m.priorDecoder)):
    x = y if name == "h_a" else z
    total = 0.0
    for i, layer in enumerate(net.layers()):
        out, ms = timed(layer, x)
        px = out.shape[0] * out.shape[1] * out.shape[2]
        print(f"{name} layer {i + 1}: {tuple(x.shape)} -> {tuple(out.shape)}  {ms:.3f} ms  ({px} output pixels, kernel {layer.kernel_name})")
        total += ms
        x = out
    print(f"{name}: {total:.3f} ms")
    if name == "h_a":
        z = torch.round(x)
