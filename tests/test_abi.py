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
