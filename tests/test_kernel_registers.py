"""The register budget the codec pipeline is built round, checked on the built library's own metadata (no GPU): the
decoder runs beside the first analysis layer, and two waves of each fit a SIMD's 512 registers only while the layer
stays at 208 and the decoder at 48 (a decoder wave beside two waves of 216 registers or more evicts one of them: the
layer then runs 20 % slower)."""
import os
import shutil
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import kernel_registers  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(kernel_registers.LLVM, "llvm-objdump"))
                                or not os.path.exists(os.path.join(REPO, "compression_amd", "libmpc.so")),
                                reason="needs the ROCm LLVM tools and a built libmpc.so")


@pytest.fixture(scope="module")
def kernels():
    return kernel_registers.kernels()


def _one(kernels, *parts):
    hits = [v for k, v in kernels.items() if all(p in k for p in parts)]
    assert len(hits) == 1, f"{parts}: {len(hits)} kernels match"
    return hits[0]


def test_decoder_fits_beside_the_first_analysis_layer(kernels):
    mangled = shutil.which("c++filt") is None
    conv1 = _one(kernels, "conv1_rgb_patch_x3_kernelILi6E" if mangled else "conv1_rgb_patch_x3_kernel<6>")
    decoder = _one(kernels, "decode_streams_wave_kernelILb1E" if mangled else "decode_streams_wave_kernel<true>")
    assert conv1["agpr"] == 0 and decoder["agpr"] == 0
    # registers are handed out in blocks of 8
    blocks = lambda n: -(-n // 8) * 8
    assert 2 * blocks(conv1["vgpr"]) + 2 * blocks(decoder["vgpr"]) <= 512, (conv1, decoder)
    assert decoder["lds"] == 0                              # (the layer's two workgroups take the CU's LDS)


def test_every_kernel_is_found_with_its_metadata(kernels):
    assert len(kernels) > 50
    for name, k in kernels.items():
        assert 0 < k["vgpr"] + k["agpr"] <= 512 and k["scratch"] <= 64, (name, k)
