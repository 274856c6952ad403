"""Determinism check of one transform layer: runs it N times on one input and reports which runs differ from the first."""
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
layer = ops.ConvLayer(_lib.MPC_DECONV, w, b, 5, 2, _lib.MPC_ACT_IGDN, beta, gamma, f16=True)
B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 128, 192
x = torch.randn(B, H, W, cin, device=dev).half()
ref = layer(x).clone()
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    out = layer(x)
    d = (out.float() - ref.float()).abs()
    n = int((d > 0).sum())
    if n:
        bad += 1
        nz = (d > 0).nonzero()
        print(it, "differs:", n, "values, max", float(d.max()), "first", nz[0].tolist(), "last", nz[-1].tolist(), flush=True)
print("runs that differ from the first:", bad, flush=True)
