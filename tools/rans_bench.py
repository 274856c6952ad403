#!/usr/bin/env python3
"""Times the MRS1 coder alone at the benchmark's geometry (24 images x 1536 locations x 256 channels, K = 64,
L = 257) on a synthetic table with peaked rows.  usage: python tools/rans_bench.py [--reps 10]
Prints milliseconds per call for encode and decode, bits per symbol, and the decoder's status / round trip."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compression_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--images", type=int, default=24)
    ap.add_argument("--n-loc", type=int, default=1536)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    K, M, L = 64, 256, 257
    rng = np.random.default_rng(0)
    # peaked pmfs (about 14 bits/symbol would be uniform; aim at ~3.5 bits) with every symbol possible
    centre = rng.integers(64, 192, size=(K, M, 1))
    width = rng.uniform(1.5, 6.0, size=(K, M, 1))
    s = np.arange(L - 1)[None, None, :]
    pmf = np.exp(-np.abs(s - centre) / width) + 1e-9
    freq = np.maximum(1, np.floor(pmf / pmf.sum(-1, keepdims=True) * (65536 - 256)).astype(np.int64))
    freq[..., 128] += 65536 - freq.sum(-1)
    cdf = np.concatenate([np.zeros((K, M, 1), np.int64), np.cumsum(freq, -1)], -1)
    assert (cdf[..., -1] == 65536).all() and (np.diff(cdf, axis=-1) > 0).all()
    table = torch.from_numpy(cdf.astype(np.uint16).view(np.int16)).to(dev)
    n_img, n_loc = a.images, a.n_loc
    idx = torch.from_numpy(rng.integers(0, K, size=n_img * n_loc).astype(np.int32)).to(dev)
    # symbols drawn from each row's own distribution
    u = rng.integers(0, 65536, size=(n_img * n_loc, M))
    rows = cdf[idx.cpu().numpy()]                                     # [N, M, L]
    sym_np = (rows[..., 1:] <= u[..., None]).sum(-1).astype(np.int16)
    sym = torch.from_numpy(sym_np.reshape(-1)).to(dev)
    est = n_loc * M * 0.5
    ns, lps = ops.choose_streams(n_loc, M, est)
    out, nbytes, st = ops.rans_encode(table, idx, sym, n_img, ns, lps)
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0
    bits = float(nbytes.sum()) * 8 / sym.numel()
    dec, st2 = ops.rans_decode(table, idx, out, nbytes, n_img, ns, lps)
    torch.cuda.synchronize()
    ok = bool((dec.reshape(-1) == sym).all())

    def timeit(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    t_enc = timeit(lambda: ops.rans_encode(table, idx, sym, n_img, ns, lps))
    t_dec = timeit(lambda: ops.rans_decode(table, idx, out, nbytes, n_img, ns, lps))
    print(f"streams/image {ns}  symbols/stream {lps * M}  {bits:.2f} bits/symbol  encode {t_enc:.3f} ms  decode {t_dec:.3f} ms  "
          f"status {st2.cpu().tolist()[:4]}  round trip {'ok' if ok else 'MISMATCH'}", flush=True)
    nsym = lps * M
    print(f"decode: {t_dec * 1e6 / nsym:.0f} ns per symbol of one stream", flush=True)


if __name__ == "__main__":
    main()
