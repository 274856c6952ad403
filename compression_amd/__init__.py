"""compression_amd: the Manypriors / JDC codec inference path on AMD MI355X (gfx950).

Python mirrors of the reference's model classes (same names, constructor arguments, methods and
state_dict keys) that run every numeric step as a hand-written HIP kernel through the C ABI of
libmpc.so (include/mpc.h).  PyTorch tensors are used as device-memory containers only.
"""
from . import _lib
from ._lib import MpcError, build  # noqa: F401


def _registry():
    from .Balle2018PT_compressor import Balle2018PTTFExp_ImageCompressor
    from .manynets_compressor import Balle2017ManyPriors_ImageCompressor, ManyPriors_ImageCompressor
    from .manynets_dc import ManyPriors_DC
    from .testolina_compressor import TestolinaHP_ImageCompressor, TestolinaManyPriors_ImageCompressor
    return {c.__name__: c for c in (Balle2018PTTFExp_ImageCompressor, Balle2017ManyPriors_ImageCompressor,
                                    ManyPriors_DC, TestolinaHP_ImageCompressor, TestolinaManyPriors_ImageCompressor)} | \
        {"ManyPriors_ImageCompressor": ManyPriors_ImageCompressor}


def init_model(arch, device, std_suffix="ImageCompressor", **kwargs):
    """Class lookup by `arch` + suffix, construct, move to `device` -- the naming convention of the
    reference's model_ops.init_model (src/compression/libs/model_ops.py:187-208)."""
    reg = _registry()
    for name in (arch, f"{arch}_{std_suffix}", f"{arch}_DC"):
        if name in reg:
            return reg[name](device=device, **kwargs).to(device)
    raise NotImplementedError(f"unknown architecture {arch!r}; available: {sorted(reg)}")
