#!/usr/bin/env python3
"""Times single transform layers at the benchmark's shapes (24 x 768x512 images).
usage: python tools/layer_bench.py [conv1 conv2 conv3 conv4 deconv1 deconv2 deconv3 deconv4] [--reps 20]
Prints average milliseconds (HIP events) and algorithmic TFLOP/s per layer."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compression_amd import _lib, ops  # noqa: E402
from tests import synth  # noqa: E402

LAYERS = {
    # name: (kind, cin, cout, stride, act, mode, input H, W)
    "conv1": (_lib.MPC_CONV, 3, 192, 2, _lib.MPC_ACT_GDN, "x3", 512, 768),
    "conv2": (_lib.MPC_CONV, 192, 192, 2, _lib.MPC_ACT_GDN, "x3", 256, 384),
    "conv3": (_lib.MPC_CONV, 192, 192, 2, _lib.MPC_ACT_GDN, "x3", 128, 192),
    "conv4": (_lib.MPC_CONV, 192, 256, 2, _lib.MPC_ACT_NONE, "x3", 64, 96),
    "deconv1": (_lib.MPC_DECONV, 256, 192, 2, _lib.MPC_ACT_IGDN, "f16", 32, 48),
    "deconv2": (_lib.MPC_DECONV, 192, 192, 2, _lib.MPC_ACT_IGDN, "f16", 64, 96),
    "deconv3": (_lib.MPC_DECONV, 192, 192, 2, _lib.MPC_ACT_IGDN, "f16", 128, 192),
    "deconv4": (_lib.MPC_DECONV, 192, 3, 2, _lib.MPC_ACT_NONE, "f16", 256, 384),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("layers", nargs="*", default=list(LAYERS))
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=24)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name in a.layers:
        kind, cin, cout, stride, act, mode, H, W = LAYERS[name]
        shape = (cout, cin, 5, 5) if kind == _lib.MPC_CONV else (cin, cout, 5, 5)
        w = (torch.randn(shape, generator=g) * 0.02).to(dev)
        b = (torch.randn(cout, generator=g) * 0.1).to(dev)
        beta = gamma = None
        if act in (_lib.MPC_ACT_GDN, _lib.MPC_ACT_IGDN):
            beta, gamma = (t.to(dev) for t in synth._gdn(g, cout))
        layer = ops.ConvLayer(kind, w, b, 5, stride, act, beta, gamma, f16=mode == "f16", x3=mode == "x3",
                              in_bound=1.0 if mode == "x3" else 0.0)
        B = a.batch
        if mode == "x3" and cin > 4:
            x = (torch.randn(B, H, W, cin // 8, 16, device=dev) * 40).half()
            x[..., 8:] *= 1e-3
        elif mode == "x3":
            x = torch.rand(B, H, W, 4, device=dev)
        else:
            x = torch.randn(B, H, W, cin, device=dev).half()
        kw = dict(out_split=True) if (mode == "x3" and act == _lib.MPC_ACT_GDN) else {}
        for _ in range(3):
            layer(x, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            layer(x, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        print(f"{name:8s} {layer.kernel_name:42s} {ms:7.3f} ms  {layer.flops(B, H, W) / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
