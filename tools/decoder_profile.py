#!/usr/bin/env python3
"""The fast table decoder on the benchmark's workload (24 x 768x512, the synthetic model of bench.py's config 3): share
of the symbols that fall outside the compact rows' window, bits per symbol, renormalisations per symbol, and the time of
the decode call alone.  usage: python tools/decoder_profile.py [rand|smooth]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import compression_amd as ca  # noqa: E402
from compression_amd import ops  # noqa: E402
from tests import synth  # noqa: E402

data = sys.argv[1] if len(sys.argv) > 1 else "rand"
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
model = ca.init_model("ManyPriors", dev, out_channel_N=192, out_channel_M=256, num_distributions=64, analysis_precision="f16x3",
                      synthesis_precision="fp16")
model.load_state_dict(synth.manypriors_state_dict(234))
model.eval()
x = (synth.rand_image if data == "rand" else synth.smooth_image)(1000, 24, 512, 768).to(dev)
streams, idx, h, w = model.encode_batch(x)
sym, _, _, _ = model.encode_batch(x, entropy_coding=False)
B, M = 24, 256
table = model.entropy_table.to(torch.int32) & 0xffff                      # [K, M, 257]
heads = table[:, :, 0:256:16]                                             # [K, M, 16]
upper = torch.cat([heads[:, :, 4:], torch.full_like(heads[:, :, :1], 65536)], dim=2)   # head of segment t + 4 (2^16 at the end)
mass = upper[:, :, :13] - heads[:, :, :13]
w0 = mass.argmax(dim=2)                                                   # first of equal maxima, like the kernel
rows_w0 = w0[idx.view(-1).long()]                                         # [B*n_loc, M]
s = sym.view(-1, M).long()
rel = s - 16 * rows_w0
outside = ((rel < 0) | (rel > 62)).float().mean().item()
nbytes = sum(len(b) for b in streams)
print(f"{data}: {nbytes * 8 / s.numel():.3f} bits per symbol, {nbytes / 2 / s.numel():.3f} renormalisations per symbol, "
      f"{outside:.4%} of the symbols outside the window, geometry {model.last_stream_geometry}")
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    job = model.decode_batch_begin(streams, idx, h, w)
    job["done"].synchronize()
    dt = time.perf_counter() - t0
print(f"decode_batch_begin (upload + compact rows + decode): {dt * 1e3:.3f} ms")
n_streams, lps = model.last_stream_geometry
print(f"{lps * M} symbols per stream")
