#!/usr/bin/env python3
"""Time of the Laplace-row decoder against the scale of the rows: 8 images x 1536 locations x 256 channels, every
element the same scale, symbols drawn from the rows' own distribution, 87 streams an image (what the stream budget gives
the benchmark's hyperprior configuration).  usage: python tools/laplace_decoder_rate.py [scale ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from compression_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
scales = [float(a) for a in sys.argv[1:]] or [0.3, 1.0, 2.0, 3.9, 4.1, 10.0, 30.0]
n_img, n_loc, M, L, first = 8, 1536, 256, 257, -127.5
n_streams = 87
lps = -(-n_loc // n_streams)
n_streams = -(-n_loc // lps)
g = torch.Generator().manual_seed(1)
for b in scales:
    n = n_img * n_loc * M
    u = torch.rand(n, generator=g) - 0.5
    y = torch.round(-b * torch.sign(u) * torch.log1p(-2 * u.abs()))
    sym = (y + 127).clamp(0, 255).to(torch.int16).to(dev)
    sigma = torch.full((n,), b, device=dev)
    out, nbytes, status = ops.rans_encode_laplace(sigma, sym, n_img, M, first, L, 16, n_streams, lps)
    assert status.cpu().tolist() == [0] * n_img
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got, st = ops.rans_decode_laplace(sigma, out, nbytes, n_img, M, first, L, 16, n_streams, lps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert st.cpu().tolist() == [0] * n_img and torch.equal(got.view(-1), sym)
    per_stream = lps * M
    print(f"scale {b:6.2f}: {dt * 1e3:7.3f} ms, {dt / per_stream * 1e9:6.1f} ns per symbol of a stream, "
          f"{nbytes.sum().item() * 8 / n:5.2f} bits per symbol, outside the window {(y.abs() > 7).float().mean().item():.4f}")
