"""The C-ABI library loads without a GPU and exports exactly what include/mpc.h declares (no compute
calls here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "mpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from compression_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from compression_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/mpc.h but not exported by libmpc.so"
    lib.mpc_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.mpc_version()


def test_no_cpu_fallback():
    """A CPU tensor must be refused loudly rather than computed some other way."""
    import torch
    from compression_amd import _lib, ops
    with pytest.raises(_lib.MpcError):
        ops.renorm_cast_cdf(torch.zeros(4, 257))
    with pytest.raises(_lib.MpcError):
        _lib.context_for(torch.device("cpu"))


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(REPO, "compression_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f


def test_archive_container_round_trip(tmp_path):
    """MPA1: main string + cheapest side-string coding in one file; its side string costs what get_num_bits says."""
    import numpy as np
    import torch
    from compression_amd import pt_helpers
    rng = np.random.default_rng(3)
    for h, w, hi in ((32, 48, 64), (4, 6, 3), (8, 8, 1), (16, 24, 300)):
        idx = torch.from_numpy(rng.integers(0, hi, size=(h, w)))
        if hi == 1:
            idx[:] = 0
        main = rng.integers(0, 256, size=1000, dtype=np.uint8).tobytes()
        p = tmp_path / f"a{h}.mpa"
        n = pt_helpers.write_archive(str(p), main, idx, 16 * h, 16 * w)
        m2, idx2, H, W = pt_helpers.read_archive(str(p))
        assert m2 == main and torch.equal(idx2, idx.to(torch.int64)) and (H, W) == (16 * h, 16 * w)
        side_bits = pt_helpers.get_num_bits(idx.view(1, h, w), integers=True, try_png=True)
        assert 8 * (n - pt_helpers._ARCHIVE_HEAD.size - len(main)) == (side_bits + 7) // 8 * 8
    (tmp_path / "bad").write_bytes(b"MPA1" + b"\0" * 10)
    import pytest
    with pytest.raises(ValueError):
        pt_helpers.read_archive(str(tmp_path / "bad"))


def test_stream_geometry_invariants():
    """choose_streams: every location is covered, the 16-bit stream-length field is respected, the header overhead
    stays within the budget unless the length field forces more streams, and empty streams are dropped."""
    from compression_amd import ops
    for n_loc, M, est in ((1536, 256, 172e3), (96, 256, 8e3), (31680, 256, 3.5e6), (6, 192, 400.0), (1, 256, 50.0),
                          (1536, 320, 10.0), (4096, 8, 1e4)):
        n, lps = ops.choose_streams(n_loc, M, est)
        assert n >= 1 and lps >= 1 and n * lps >= n_loc and (n - 1) * lps < n_loc
        assert lps * M + 2 <= 65535
        min_streams = -(-n_loc // max(1, ops.MRS1_MAX_SYMBOLS_PER_STREAM // M))
        if n > min_streams:
            assert ops.MRS1_FIXED_HEADER + n * ops.MRS1_STREAM_OVERHEAD <= est * 0.004 + ops.MRS1_STREAM_OVERHEAD


def test_statistics_accumulator_with_msssim():
    import math
    from compression_amd.distributed import StatsAccumulator
    acc = StatsAccumulator()
    acc.add_image(bits=8000, pixels=100 * 100, sq_err=3.0, nbytes=1000, msssim=0.99)
    acc.add_image(bits=4000, pixels=100 * 100, sq_err=6.0, nbytes=500, msssim=0.9)
    r = acc.all_reduce().result()
    assert r["images"] == 2 and r["bpp"] == 12000 / 20000
    assert r["msssim"] == (0.99 + 0.9) / 2
    assert abs(r["msssimDB"] - (-10 * math.log10(0.01) - 10 * math.log10(0.1)) / 2) < 1e-9
    assert abs(r["psnr"] - (10 * math.log10(30000 / 3.0) + 10 * math.log10(30000 / 6.0)) / 2) < 1e-9


def test_hyperprior_latent_sizes_and_window():
    """Host arithmetic of the hyperprior codec and of the statistics module (no kernel is called)."""
    import torch
    from compression_amd import pt_ms_ssim
    from compression_amd.abstract_model import AbstractHyperpriorImageCompressor as A
    assert A._latent_hw(None, 128, 192) == ((8, 12), (2, 3))
    assert A._latent_hw(None, 512, 768) == ((32, 48), (8, 12))
    from oracle import manypriors_oracle as O
    assert torch.equal(pt_ms_ssim._fspecial_gauss_1d(11, 1.5), O.gauss_window(11, 1.5))


def test_mrs1_header_validation_on_the_host():
    """parse_mrs1_headers refuses everything the device kernels would derive a wrong address from."""
    import struct
    from compression_amd import ops
    n_loc, M = 120, 256
    good = struct.pack("<4I", 0x3153524D, 4, n_loc * M, 30 * M) + b"\0" * 64
    assert ops.parse_mrs1_headers([good, good], n_loc, M) == (4, 30)

    def forged(**kw):
        f = dict(magic=0x3153524D, n_streams=4, n_symbols=n_loc * M, sps=30 * M, tail=64)
        f.update(kw)
        return struct.pack("<4I", f["magic"], f["n_streams"], f["n_symbols"], f["sps"]) + b"\0" * f["tail"]

    for bad in (b"MRS1", forged(magic=1), forged(n_streams=0), forged(n_streams=2 ** 31 - 8), forged(n_streams=n_loc + 1),
                forged(n_symbols=n_loc * M + 1), forged(sps=30 * M + 1), forged(sps=0), forged(sps=10 * M),
                forged(sps=65536 // M * M + M), forged(n_streams=100, sps=2 * M, tail=8)):
        with pytest.raises(ValueError):
            ops.parse_mrs1_headers([bad], n_loc, M)
    with pytest.raises(ValueError, match="share one stream geometry"):
        ops.parse_mrs1_headers([good, forged(n_streams=5, sps=24 * M)], n_loc, M)
    with pytest.raises(ValueError):
        ops.parse_mrs1_headers([], n_loc, M)


def test_host_authority_float_tables_equal_the_reference_capture(golden):
    """The host-side evaluation of the learned CDF tables (table_authority="host": the reference's own ATen calls on
    CPU copies of the parameters) against rows captured from the reference itself in the build container
    (tests/golden/tables_std.npz, forward_hyperprior.npz): bit for bit.  No GPU is involved in that evaluation."""
    import hashlib
    import numpy as np
    import torch  # noqa: F401
    from compression_amd.bitEstimator import BitEstimator, MultiHeadBitEstimator
    from tests import synth
    N, M, K = 192, 256, 64
    g = golden("tables_std")
    sd = synth.manypriors_state_dict(234, N, M, K, spread=True)
    est = MultiHeadBitEstimator(M, nb_head=K)
    est.load_state_dict({k[len("bitEstimators."):]: v for k, v in sd.items() if k.startswith("bitEstimators.")})
    tf = est.table_host(-127.5, 128.5, 257)
    assert tuple(tf.shape) == (K, M, 257)
    assert hashlib.sha256(tf.contiguous().numpy().tobytes()).hexdigest() == str(g["sha_table_f"])
    gh = golden("forward_hyperprior")
    sdh = synth.hyperprior_state_dict(236, N, M)
    zest = BitEstimator(N)
    zest.load_state_dict({k[len("bitEstimator_z."):]: v for k, v in sdh.items() if k.startswith("bitEstimator_z.")})
    assert np.array_equal(zest.table_host(-127.5, 128.5, 257).numpy(), gh["z_table_f"])
