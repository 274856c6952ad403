"""Determinism check of the split-precision conv2 layer (patch kernel): N runs on one input, compared with the first."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from compression_amd import _lib, ops
from tests import synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cin = cout = 192
w = (torch.randn(cout, cin, 5, 5, generator=g) * 0.02).to(dev)
b = (torch.randn(cout, generator=g) * 0.1).to(dev)
beta, gamma = (t.to(dev) for t in synth._gdn(g, cout))
layer = ops.ConvLayer(_lib.MPC_CONV, w, b, 5, 2, _lib.MPC_ACT_GDN, beta, gamma, x3=True, in_bound=1.0)
B = int(sys.argv[1])
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (256, 384)
x = (torch.randn(B, H, W, cin // 8, 16, device=dev) * 40).half()
x[..., 8:] *= 1e-3
ref = layer(x, out_split=True).clone()
bad = 0
for it in range(int(sys.argv[2])):
    out = layer(x, out_split=True)
    d = (out.float() - ref.float()).abs()
    n = int((d > 0).sum())
    if n:
        bad += 1
        nz = (d > 0).nonzero()
        print(it, "differs:", n, "values, max", float(d.max()), "first", nz[0].tolist(), "last", nz[-1].tolist(), flush=True)
print(layer.kernel_name, "runs that differ from the first:", bad, flush=True)
