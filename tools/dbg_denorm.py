"""Does the fp16 deconv layer's speed depend on the values?  Times one layer on inputs of different magnitudes."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from compression_amd import _lib, ops
from tests import synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cin = cout = 192
w = (torch.randn(cin, cout, 5, 5, generator=g) * 0.02).to(dev)
b = (torch.randn(cout, generator=g) * 0.1).to(dev)
beta, gamma = (t.to(dev) for t in synth._gdn(g, cout))
B, H, W = 8, 256, 256
base = torch.randn(B, H, W, cin, device=dev)
for wscale in (1.0, 1e-3):
    layer = ops.ConvLayer(_lib.MPC_DECONV, w * wscale, b, 5, 2, _lib.MPC_ACT_IGDN, beta, gamma, f16=True)
    for scale in (1.0, 1e-2, 1e-4, 1e-5, 1e-6, 0.0, 1e3, 6e4):
        x = (base * scale).half()
        relu = x.clamp(min=0)
        for name, t in (("randn", x), ("relu", relu)):
            layer(t); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): out = layer(t)
            e1.record(); torch.cuda.synchronize()
            den = int(((t.abs() < 6.1e-5) & (t != 0)).sum())
            print(f"w x{wscale:g} in x{scale:g} {name}: {e0.elapsed_time(e1)/10:.3f} ms  denormal inputs {den/t.numel():.3f} out finite {bool(torch.isfinite(out.float()).all())}", flush=True)
