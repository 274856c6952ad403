"""GPU parity tests of the transform layers (f32-MFMA implicit GEMM with fused GDN) against
PyTorch-CPU fp32 convolutions, layer by layer and end to end."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import manypriors_oracle as O
from tests import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from compression_amd import ops as _ops
    return _ops


def nhwc(x, cpad=None):
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad and cpad > x.shape[-1]:
        x = F.pad(x, (0, cpad - x.shape[-1]))
    return x


def rel_err(got, want):
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-12)).item()


def _w(g, *shape, scale=0.05):
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("cin,cout,k,stride,hw", [
    (192, 192, 5, 2, (20, 28)),      # g_a conv2/conv3, h_a conv2/3
    (192, 256, 5, 2, (12, 20)),      # g_a conv4 (two cout groups)
    (256, 192, 3, 1, (7, 9)),        # h_a conv1
    (192, 192, 5, 1, (9, 6)),        # Testolina conv
    (64, 128, 5, 2, (17, 13)),       # odd sizes, other channel counts
])
def test_conv2d_plain(dev, ops, cin, cout, k, stride, hw):
    from compression_amd import _lib
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(2, cin, *hw, generator=g)
    w, b = _w(g, cout, cin, k, k), _w(g, cout)
    want = F.conv2d(x, w, b, stride=stride, padding=(k - 1) // 2)
    layer = ops.ConvLayer(_lib.MPC_CONV, w.to(dev), b.to(dev), k, stride)
    got = layer(nhwc(x).to(dev)).cpu().permute(0, 3, 1, 2)
    assert got.shape == want.shape
    assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize("cin,cout,k,stride,hw", [
    (256, 192, 5, 2, (6, 9)),        # g_s deconv1
    (192, 192, 5, 2, (11, 7)),       # g_s deconv2/3, h_s deconv1/2
    (192, 256, 3, 1, (8, 12)),       # h_s deconv3
])
def test_conv_transpose2d_plain(dev, ops, cin, cout, k, stride, hw):
    from compression_amd import _lib
    g = torch.Generator().manual_seed(cin * 3 + cout + k)
    x = torch.randn(2, cin, *hw, generator=g)
    w, b = _w(g, cin, cout, k, k), _w(g, cout)
    want = F.conv_transpose2d(x, w, b, stride=stride, padding=(k - 1) // 2, output_padding=stride - 1)
    layer = ops.ConvLayer(_lib.MPC_DECONV, w.to(dev), b.to(dev), k, stride)
    got = layer(nhwc(x).to(dev)).cpu().permute(0, 3, 1, 2)
    assert got.shape == want.shape
    assert rel_err(got, want) < 1e-5


def test_first_layer_rgb_with_gdn(dev, ops):
    from compression_amd import _lib
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 34, 50, generator=g)
    w, b = _w(g, 192, 3, 5, 5, scale=0.3), _w(g, 192)
    beta, gamma = synth._gdn(g, 192)
    want = O.gdn(F.conv2d(x, w, b, stride=2, padding=2), beta, gamma, inverse=False)
    layer = ops.ConvLayer(_lib.MPC_CONV, w.to(dev), b.to(dev), 5, 2, _lib.MPC_ACT_GDN, beta.to(dev), gamma.to(dev))
    got = layer(ops.nchw_to_nhwc(x.to(dev), 4)).cpu().permute(0, 3, 1, 2)
    assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize("inverse", [False, True])
def test_fused_gdn_uses_gamma_rows_not_columns(dev, ops, inverse):
    """gamma is deliberately asymmetric: a transposed channel mix fails this test."""
    from compression_amd import _lib
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 192, 10, 14, generator=g)
    beta = 1.0 + torch.rand(192, generator=g)
    gamma = 0.3 * torch.rand(192, 192, generator=g) * (torch.arange(192).view(-1, 1) / 192.0 + 0.1)
    if inverse:
        w, b = _w(g, 192, 192, 5, 5), _w(g, 192)
        pre = F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1)
        kind, act = _lib.MPC_DECONV, _lib.MPC_ACT_IGDN
    else:
        w, b = _w(g, 192, 192, 5, 5), _w(g, 192)
        pre = F.conv2d(x, w, b, stride=2, padding=2)
        kind, act = _lib.MPC_CONV, _lib.MPC_ACT_GDN
    want = O.gdn(pre, beta, gamma, inverse=inverse)
    layer = ops.ConvLayer(kind, w.to(dev), b.to(dev), 5, 2, act, beta.to(dev), gamma.to(dev))
    got = layer(nhwc(x).to(dev)).cpu().permute(0, 3, 1, 2)
    assert rel_err(got, want) < 1e-5
    wrong = O.gdn(pre, beta, gamma.t().contiguous(), inverse=inverse)
    assert rel_err(wrong, want) > 1e-2


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("shape", [(2, 192, 13, 21), (1, 128, 5, 7), (1, 256, 33, 4), (3, 32, 1, 1), (2, 64, 3, 5, 6)])
def test_gdn_layer_called_directly(dev, inverse, shape):
    """The GDN module as a layer of its own (GDN.py:65-95) against the oracle's restatement, 4-D and 5-D inputs (the
    5-D tensor is folded to [B, C, D*W, H] and unfolded again), ragged pixel counts, asymmetric gamma, parameters below
    their lower bounds (they are clamped, :75-81).  The channel mix is an fp32 fma chain in another order than ATen's
    1x1 convolution: agreement to fp32 rounding."""
    from compression_amd.GDN import GDN
    g = torch.Generator().manual_seed(60 + len(shape))
    C = shape[1]
    layer = GDN(C, inverse=inverse)
    layer.beta.data = 0.5 + torch.rand(C, generator=g)
    layer.beta.data[::7] = 0.0                                     # below the bound: beta_eff = 1e-6
    gamma = 0.3 * torch.rand(C, C, generator=g) * (torch.arange(C).view(-1, 1) / C + 0.1)
    gamma[3::5, 1::4] = -1.0                                       # below the bound: gamma_eff = 0
    layer.gamma.data = gamma
    x = torch.randn(*shape, generator=g) * 3
    x4 = x if x.dim() == 4 else x.view(shape[0], C, shape[2] * shape[3], shape[4])
    want = O.gdn(x4, layer.beta.data, layer.gamma.data, inverse=inverse).view(shape)
    got = layer.to(dev)(x.to(dev)).cpu()
    assert got.shape == x.shape
    assert rel_err(got, want) < 2e-6
    assert rel_err(O.gdn(x4, layer.beta.data.cpu(), layer.gamma.data.cpu().t().contiguous(), inverse=inverse).view(shape), want) > 1e-3


def test_last_layer_to_rgb_nchw_clamped(dev, ops):
    from compression_amd import _lib
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 192, 9, 13, generator=g)
    w, b = _w(g, 192, 3, 5, 5, scale=0.02), _w(g, 3, scale=0.5)
    want = F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1)
    layer = ops.ConvLayer(_lib.MPC_DECONV, w.to(dev), b.to(dev), 5, 2)
    got = layer(nhwc(x).to(dev)).cpu()
    assert got.shape == want.shape
    assert rel_err(got, want) < 1e-5
    got_c = layer(nhwc(x).to(dev), clamp01=True).cpu()
    assert torch.allclose(got_c, want.clamp(0, 1), atol=1e-5)


def test_relu_exp_abs_round_and_nchw_variants(dev, ops):
    from compression_amd import _lib
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, 256, 6, 8, generator=g)
    w, b = _w(g, 192, 256, 3, 3), _w(g, 192)
    want = F.relu(F.conv2d(x.abs(), w, b, padding=1))
    layer = ops.ConvLayer(_lib.MPC_CONV, w.to(dev), b.to(dev), 3, 1, _lib.MPC_ACT_RELU, in_abs=True)
    assert rel_err(layer(nhwc(x).to(dev)).cpu().permute(0, 3, 1, 2), want) < 1e-5
    x2 = torch.randn(1, 192, 6, 8, generator=g)
    w2, b2 = _w(g, 192, 256, 3, 3, scale=0.02), _w(g, 256)
    want2 = torch.exp(F.conv_transpose2d(x2, w2, b2, stride=1, padding=1))
    layer2 = ops.ConvLayer(_lib.MPC_DECONV, w2.to(dev), b2.to(dev), 3, 1, _lib.MPC_ACT_EXP)
    assert rel_err(layer2(nhwc(x2).to(dev)).cpu().permute(0, 3, 1, 2), want2) < 1e-5
    w3, b3 = _w(g, 256, 192, 5, 5, scale=0.2), _w(g, 256)
    want3 = F.conv2d(x2, w3, b3, stride=2, padding=2)
    layer3 = ops.ConvLayer(_lib.MPC_CONV, w3.to(dev), b3.to(dev), 5, 2)
    y, yr = layer3(nhwc(x2).to(dev), round_out=True, nchw_out=True)
    assert rel_err(y.cpu(), want3) < 1e-5
    assert torch.equal(yr.cpu(), torch.round(y.cpu()))


def test_layer_create_rejects_unsupported_shapes(dev, ops):
    from compression_amd import _lib
    w, b = torch.zeros(100, 192, 5, 5, device=dev), torch.zeros(100, device=dev)
    with pytest.raises(_lib.MpcError):
        ops.ConvLayer(_lib.MPC_CONV, w, b, 5, 2)
    with pytest.raises(_lib.MpcError):
        ops.ConvLayer(_lib.MPC_CONV, torch.zeros(192, 192, 5, 5), torch.zeros(192), 5, 2)     # CPU tensors


@pytest.mark.parametrize("cin,act", [(256, "igdn"), (192, "igdn"), (192, "relu_conv"), (192, "rgb")])
def test_fp16_layers_against_fp32_reference(dev, ops, cin, act):
    """fp16-operand layers of the synthesis transform: relative error of a single layer ~1e-3."""
    from compression_amd import _lib
    g = torch.Generator().manual_seed(cin + len(act))
    x = torch.randn(2, cin, 9, 12, generator=g)
    xh = nhwc(x).to(dev).half()
    if act == "igdn":
        w, b = _w(g, cin, 192, 5, 5), _w(g, 192)
        beta = 1.0 + torch.rand(192, generator=g)
        gamma = 0.3 * torch.rand(192, 192, generator=g) * (torch.arange(192).view(-1, 1) / 192.0 + 0.1)
        want = O.gdn(F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1), beta, gamma, inverse=True)
        layer = ops.ConvLayer(_lib.MPC_DECONV, w.to(dev), b.to(dev), 5, 2, _lib.MPC_ACT_IGDN, beta.to(dev), gamma.to(dev), f16=True)
        got = layer(xh).float().cpu().permute(0, 3, 1, 2)
        got2 = layer(xh, wide_block=True).float().cpu().permute(0, 3, 1, 2)
    elif act == "relu_conv":
        w, b = _w(g, 192, cin, 5, 5), _w(g, 192)
        want = F.relu(F.conv2d(x, w, b, stride=1, padding=2))
        layer = ops.ConvLayer(_lib.MPC_CONV, w.to(dev), b.to(dev), 5, 1, _lib.MPC_ACT_RELU, f16=True)
        got = layer(xh).float().cpu().permute(0, 3, 1, 2)
        got2 = layer(xh, wide_block=True).float().cpu().permute(0, 3, 1, 2)
    else:
        w, b = _w(g, cin, 3, 5, 5, scale=0.02), _w(g, 3, scale=0.5)
        want = F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1)
        layer = ops.ConvLayer(_lib.MPC_DECONV, w.to(dev), b.to(dev), 5, 2, f16=True)
        got = layer(xh).cpu()
        got2 = layer(xh, wide_block=True).cpu()
        assert got.dtype == torch.float32
    assert got.shape == want.shape
    assert torch.equal(got, got2), "8-wave workgroups must give identical results"
    err = rel_err(got, want)
    print(f"fp16 layer {act} cin={cin}: rel err {err:.2e}")
    assert err < 4e-3


def test_fp16_transposed_layer_with_128_channels(dev, ops):
    """The narrow variant (128 channels: 8 accumulator tiles of 16 channels per wave) of the fp16 transposed layer with
    IGDN, on a grid that is not a multiple of the 16 x 8 tile."""
    from compression_amd import _lib
    g = torch.Generator().manual_seed(128)
    x = torch.randn(2, 128, 11, 19, generator=g)
    w, b = _w(g, 128, 128, 5, 5), _w(g, 128)
    beta = 1.0 + torch.rand(128, generator=g)
    gamma = 0.3 * torch.rand(128, 128, generator=g) * (torch.arange(128).view(-1, 1) / 128.0 + 0.1)
    want = O.gdn(F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1), beta, gamma, inverse=True)
    layer = ops.ConvLayer(_lib.MPC_DECONV, w.to(dev), b.to(dev), 5, 2, _lib.MPC_ACT_IGDN, beta.to(dev), gamma.to(dev), f16=True)
    got = layer(nhwc(x).to(dev).half()).float().cpu().permute(0, 3, 1, 2)
    assert got.shape == want.shape
    err = rel_err(got, want)
    print(f"fp16 transposed layer, 128 channels: rel err {err:.2e}")
    assert err < 4e-3


@pytest.mark.parametrize("hw", [(5, 64), (3, 32), (1, 96)])
def test_fp16_last_layer_row_tiled_kernel(dev, ops, hw):
    """The fp16 192 -> 3 layer on inputs whose width is a multiple of 32 takes the row-tiled kernel (horizontal
    taps staged through LDS); odd widths take the direct one.  Both must agree with the fp32 reference."""
    from compression_amd import _lib
    g = torch.Generator().manual_seed(hw[0] * 100 + hw[1])
    x = torch.randn(2, 192, *hw, generator=g)
    w, b = _w(g, 192, 3, 5, 5, scale=0.02), _w(g, 3, scale=0.5)
    want = F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1)
    layer = ops.ConvLayer(_lib.MPC_DECONV, w.to(dev), b.to(dev), 5, 2, f16=True)
    got = layer(nhwc(x).to(dev).half()).cpu()
    assert got.shape == want.shape
    err = rel_err(got, want)
    print(f"fp16 last layer {hw}: rel err {err:.2e}")
    assert err < 4e-3
    clamped = layer(nhwc(x).to(dev).half(), clamp01=True).cpu()
    assert torch.equal(clamped, got.clamp(0, 1))


def test_fp16_last_layer_segments_longer_than_the_minimum(dev, ops):
    """The row-tiled kernel walks column strips in segments of rows sized so that all segments fit one round of
    its persistent grid: small inputs get the 2-row minimum, this one (128 strips x 170 rows) 6-row segments with
    a ragged last one.  Checked against the fp32 reference like the small cases."""
    from compression_amd import _lib
    g = torch.Generator().manual_seed(170)
    x = torch.randn(2, 192, 170, 2048, generator=g).half()
    w, b = _w(g, 192, 3, 5, 5, scale=0.02), _w(g, 3, scale=0.5)
    layer = ops.ConvLayer(_lib.MPC_DECONV, w.to(dev), b.to(dev), 5, 2, f16=True)
    got = layer(x.to(dev).permute(0, 2, 3, 1).contiguous()).cpu()
    want = F.conv_transpose2d(x.float(), w.half().float(), b, stride=2, padding=2, output_padding=1)
    assert got.shape == want.shape
    err = ((got - want).abs().max() / want.abs().max()).item()
    print(f"fp16 last layer 170 x 2048: max err / max {err:.2e}")
    assert err < 1e-5                      # same fp16 operands, fp32 accumulation on both sides


def to_split(x_nhwc, scale=256.0):
    """fp32 NHWC -> S16 split tensor [B,H,W,C/8,16] (hi x8 | lo x8), the format f16x3 layers read."""
    v = (x_nhwc * scale).float()
    hi = v.half()
    lo = (v - hi.float()).half()
    B, H, W, C = x_nhwc.shape
    return torch.cat([hi.view(B, H, W, C // 8, 8), lo.view(B, H, W, C // 8, 8)], dim=-1).contiguous()


def from_split(s, scale=256.0):
    B, H, W, nb, _ = s.shape
    return ((s[..., :8].float() + s[..., 8:].float()) / scale).reshape(B, H, W, nb * 8)


@pytest.mark.parametrize("cout,gdn", [(192, True), (256, False)])
def test_split_precision_layer_is_fp32_grade(dev, ops, cout, gdn):
    """f16x3 (hi/lo fp16, three MFMAs per product): error against an fp64 reference must be at the level of
    the f32 kernel's, i.e. ~1e-6 relative, not fp16's 1e-3."""
    from compression_amd import _lib
    g = torch.Generator().manual_seed(cout)
    x = torch.randn(2, 192, 20, 28, generator=g)
    w, b = _w(g, cout, 192, 5, 5, scale=0.02), _w(g, cout)
    beta, gamma = synth._gdn(g, cout) if gdn else (None, None)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=2)
    if gdn:
        be, ga = O.gdn_effective(beta, gamma)
        ref = ref / torch.sqrt(F.conv2d(ref ** 2, ga.double().view(cout, cout, 1, 1), be.double()))
    act = _lib.MPC_ACT_GDN if gdn else _lib.MPC_ACT_NONE
    args = (beta.to(dev), gamma.to(dev)) if gdn else (None, None)
    l32 = ops.ConvLayer(_lib.MPC_CONV, w.to(dev), b.to(dev), 5, 2, act, *args)
    # in_bound > 0 puts the GDN channel mix on the split-precision MFMA too
    lx3 = ops.ConvLayer(_lib.MPC_CONV, w.to(dev), b.to(dev), 5, 2, act, *args, x3=True, in_bound=float(x.abs().max()))
    xs = to_split(nhwc(x)).to(dev)
    x_rt = from_split(xs.cpu())                        # what the split format actually holds (22 bits)
    assert (x_rt - nhwc(x)).abs().max().item() < 1e-6
    got32 = l32(nhwc(x).to(dev)).cpu().permute(0, 3, 1, 2).double()
    gotx3 = lx3(xs).cpu().permute(0, 3, 1, 2).double()
    assert torch.equal(lx3(xs, wide_block=True).cpu().permute(0, 3, 1, 2).double(), gotx3)
    e32 = ((got32 - ref).abs().max() / ref.abs().max()).item()
    ex3 = ((gotx3 - ref).abs().max() / ref.abs().max()).item()
    print(f"cout={cout} gdn={gdn}: rel err vs fp64  f32-MFMA {e32:.2e}   f16x3 {ex3:.2e}")
    assert ex3 < 5e-6 and ex3 < 2 * e32 + 1e-7
    # split output feeds the next layer: round trip through the S16 writer
    if gdn:
        outs = lx3(xs, out_split=True)
        assert torch.equal(lx3(xs, out_split=True, wide_block=True), outs)
        back = from_split(outs.cpu()).permute(0, 3, 1, 2).double()
        assert ((back - ref).abs().max() / ref.abs().max()).item() < 5e-6
        outs32 = l32(nhwc(x).to(dev), out_split=True)
        back32 = from_split(outs32.cpu()).permute(0, 3, 1, 2).double()
        assert ((back32 - ref).abs().max() / ref.abs().max()).item() < 5e-6


def test_split_precision_first_layer_rgb(dev, ops):
    """conv1 (3 -> 192, 5x5 s2) + GDN on the split-precision path: the im2col K axis (75 -> 96) is gathered from
    the padded NHWC image on the fly."""
    from compression_amd import _lib
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 3, 34, 50, generator=g)
    w, b = _w(g, 192, 3, 5, 5, scale=0.3), _w(g, 192)
    beta, gamma = synth._gdn(g, 192)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=2)
    be, ga = O.gdn_effective(beta, gamma)
    ref = ref / torch.sqrt(F.conv2d(ref ** 2, ga.double().view(192, 192, 1, 1), be.double()))
    x4 = ops.nchw_to_nhwc(x.to(dev), 4)
    l32 = ops.ConvLayer(_lib.MPC_CONV, w.to(dev), b.to(dev), 5, 2, _lib.MPC_ACT_GDN, beta.to(dev), gamma.to(dev))
    lx3 = ops.ConvLayer(_lib.MPC_CONV, w.to(dev), b.to(dev), 5, 2, _lib.MPC_ACT_GDN, beta.to(dev), gamma.to(dev),
                        x3=True, in_bound=1.0)
    e32 = ((l32(x4).cpu().permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max()).item()
    got = lx3(x4)
    assert torch.equal(lx3(x4, wide_block=True), got)
    ex3 = ((got.cpu().permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max()).item()
    gots = from_split(lx3(x4, out_split=True).cpu()).permute(0, 3, 1, 2).double()
    exs = ((gots - ref).abs().max() / ref.abs().max()).item()
    print(f"first layer: rel err vs fp64  f32-MFMA {e32:.2e}   f16x3 {ex3:.2e}   f16x3 split-out {exs:.2e}")
    assert ex3 < 3e-6 and exs < 3e-6


@pytest.mark.parametrize("which", ["deconv_f16", "conv_x3", "conv1_x3"])
def test_big_layers_are_bit_identical_from_launch_to_launch(dev, ops, which):
    """The LDS-DMA kernels at a size that fills the chip (two workgroups per CU, thousands of workgroups): 12 launches on
    one input must give one result.  A transfer that is read before it has landed shows up here as a few tiles that
    change from run to run (it does not show at small sizes, where a CU holds one workgroup)."""
    from compression_amd import _lib
    g = torch.Generator().manual_seed(11)
    c = 192
    beta, gamma = (t.to(dev) for t in synth._gdn(g, c))
    b = (torch.randn(c, generator=g) * 0.1).to(dev)
    if which == "deconv_f16":
        w = (torch.randn(c, c, 5, 5, generator=g) * 0.02).to(dev)
        layer = ops.ConvLayer(_lib.MPC_DECONV, w, b, 5, 2, _lib.MPC_ACT_IGDN, beta, gamma, f16=True)
        x = torch.randn(4, 128, 192, c, generator=g).half().to(dev)
        run = lambda: layer(x)
    elif which == "conv1_x3":
        w = (torch.randn(c, 3, 5, 5, generator=g) * 0.1).to(dev)
        layer = ops.ConvLayer(_lib.MPC_CONV, w, b, 5, 2, _lib.MPC_ACT_GDN, beta, gamma, x3=True, in_bound=1.0)
        x = torch.rand(4, 512, 768, 4, generator=g).to(dev)
        x[..., 3] = 0
        run = lambda: layer(x, out_split=True)
    else:
        w = (torch.randn(c, c, 5, 5, generator=g) * 0.02).to(dev)
        layer = ops.ConvLayer(_lib.MPC_CONV, w, b, 5, 2, _lib.MPC_ACT_GDN, beta, gamma, x3=True, in_bound=1.0)
        xs = torch.randn(4, 256, 384, c // 8, 16, generator=g) * 40
        xs[..., 8:] *= 1e-3
        x = xs.half().to(dev)
        run = lambda: layer(x, out_split=True)
    first = run().clone()
    differing = sum(int(not torch.equal(run(), first)) for _ in range(12))
    print(f"{layer.kernel_name}: {differing} of 12 launches differ from the first")
    assert differing == 0
