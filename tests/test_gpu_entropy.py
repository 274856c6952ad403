"""GPU parity tests of the entropy-side kernels (tables, competition, CDF rows, rANS) against the CPU
oracle and the golden vectors captured from the reference.  Everything goes through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import coders, manypriors_oracle as O
from tests import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
M, K, L = 256, 64, 257
PARAM_ORDER = ["f1.h", "f1.b", "f1.a", "f2.h", "f2.b", "f2.a", "f3.h", "f3.b", "f3.a", "f4.h", "f4.b"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from compression_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def priors():
    sd = synth.prior_weights(235, K, M, spread=True)
    tf = O.entropy_table_float(sd, M).contiguous()
    return sd, tf


def test_float_table_close_and_int_tables_exact(dev, ops, priors):
    sd, tf = priors
    params = [sd["bitEstimators." + k].to(dev) for k in PARAM_ORDER]
    tf_gpu = ops.bitparm_cdf_table(params, K * M, -127.5, L).view(K, M, L).cpu()
    # float chain: same rounding points as the reference, transcendental functions correctly rounded
    # instead of Sleef's <=1 ulp -> a few entries differ in the last bits
    assert (tf_gpu - tf).abs().max().item() < 1e-6
    frac_equal = (tf_gpu == tf).float().mean().item()
    print(f"float table: {frac_equal:.4f} of entries bit-identical to the CPU build")
    assert frac_equal > 0.5
    # integer stage: bit-exact in both rounding orders when fed the same float table
    for order in (O.ORDER_ROUND_THEN_ADD, O.ORDER_ADD_THEN_ROUND):
        want = O.renorm_cast_cdf(tf.clone(), 16, order)
        got = ops.renorm_cast_cdf(tf.to(dev), 16, order).cpu()
        assert torch.equal(got, want), f"int16 table differs (order {order})"
    # end to end (device float table -> int table): only entries whose scaled value sits at a rounding tie move
    want = O.renorm_cast_cdf(tf.clone(), 16, O.ORDER_ROUND_THEN_ADD)
    got = ops.renorm_cast_cdf(tf_gpu.to(dev), 16, O.ORDER_ROUND_THEN_ADD).cpu()
    diff = (got.int() - want.int())
    mism = (diff != 0).float().mean().item()
    print(f"int table built from the device float table: {mism:.2e} of entries differ, max |diff| {diff.abs().max().item()}")
    assert diff.abs().max().item() <= 1 and mism < 2e-3


def test_int_table_wraparound_cast(dev, ops):
    c = torch.tensor([[0.0, 0.25, 0.5, 0.61274512, 0.999, 1.0, 1.0]], dtype=torch.float32)
    for order in (0, 1):
        want = O.renorm_cast_cdf(c.clone(), 16, order)
        got = ops.renorm_cast_cdf(c.to(dev), 16, order).cpu()
        assert torch.equal(got, want)


def test_bits_lut(dev, ops, priors):
    _, tf = priors
    for form, name in ((0, "log2"), (1, "ln")):
        want = O.bits_lut(tf, name).permute(1, 2, 0).contiguous()          # [M, S, K]
        got = ops.bits_lut(tf.to(dev), form).cpu()
        assert (got - want).abs().max().item() <= 4e-6
        print(f"bits LUT ({name}): {(got == want).float().mean().item():.5f} bit-identical")
        assert (got == want).float().mean().item() > 0.99


def test_bits_lut_nan_and_clamp_semantics(dev, ops):
    tf = torch.zeros(1, 1, 5)
    tf[0, 0] = torch.tensor([0.0, 0.0, 0.5, 0.4, 1.0])       # p = 0, 0.5, -0.1 (NaN), 0.6
    want = O.bits_lut(tf, "log2").permute(1, 2, 0)
    got = ops.bits_lut(tf.to(dev), 0).cpu()
    assert torch.isnan(want[0, 2, 0]) and torch.isnan(got[0, 2, 0])
    assert torch.allclose(got[0, [0, 1, 3], 0], want[0, [0, 1, 3], 0], atol=1e-6)
    assert abs(got[0, 0, 0].item() - 33.219) < 1e-2             # p = 0 -> -log2(1e-10)


def test_competition_bit_exact_with_reference_lut(dev, ops, golden):
    """With a CPU-built LUT uploaded, totals and indices must equal the CPU path's bit for bit (same fp32
    association as ATen's sum over the channel axis).  The CPU side is evaluated live because PyTorch's
    vectorised log2/tanh differ in the last bit between host CPU generations; the golden indices
    captured from the reference (on the build container's CPU) are compared up to exact ties."""
    g = golden("competition_latent")
    sd = synth.prior_weights(235, K, M, spread=True)
    tf = O.entropy_table_float(sd, M)
    lut = O.bits_lut(tf, "log2").permute(1, 2, 0).contiguous().to(dev)
    yl = synth.synthetic_latent(5, 2, M, 16, 16)
    sym = O.symbols_from_latent(yl).view(-1, M).to(dev)
    idx, minbits, total = ops.prior_argmin(sym, lut, want_total=True)
    want_total = O.prior_bits_lut(yl, tf).reshape(K, -1)
    want_min, want_idx = want_total.min(0)
    assert torch.equal(total.cpu(), want_total), "fp32 association differs from ATen's channel sum"
    assert torch.equal(idx.cpu().long(), want_idx)
    assert torch.equal(minbits.cpu(), want_min)
    gold_total = torch.from_numpy(g["total_bits"]).reshape(K, -1)
    print(f"totals bit-identical to the golden capture: {(total.cpu() == gold_total).float().mean().item():.4f}")
    assert (total.cpu() - gold_total).abs().max().item() < 2e-3
    mism = (idx.cpu().numpy() != g["idx"].reshape(-1)).mean()
    print(f"indices vs golden capture: {mism:.2e} differ")
    assert mism <= 1 / 512
    # with the device-built LUT only exact-tie locations may move
    lut_dev = ops.bits_lut(tf.to(dev), 0)
    idx2, _, _ = ops.prior_argmin(sym, lut_dev)
    mism = (idx2.cpu().numpy() != g["idx"].reshape(-1)).mean()
    print(f"competition with device LUT: {mism:.2e} of locations differ")
    assert mism <= 2 / 512


def test_competition_nan_and_ties(dev, ops):
    Kk, Mm, S = 8, 16, 4
    lut = torch.rand(Mm, S, Kk)
    lut[:, :, 5] = lut[:, :, 2]                   # prior 5 ties with prior 2 everywhere -> 2 must win over 5
    lut[3, 1, 6] = float("nan")                   # prior 6 is NaN whenever channel 3 carries symbol 1
    sym = torch.randint(0, S, (40, Mm), dtype=torch.int16)
    sym[:20, 3] = 1
    sym[20:, 3] = 0
    bits = torch.stack([lut[torch.arange(Mm), sym[i].long()] for i in range(40)])        # [loc, M, K]
    want_min, want_idx = bits.permute(2, 0, 1).sum(2).min(0)
    idx, minbits, _ = ops.prior_argmin(sym.to(dev), lut.to(dev))
    assert (idx.cpu()[:20] == 6).all() and torch.isnan(minbits.cpu()[:20]).all()
    assert torch.equal(idx.cpu()[20:].long(), want_idx[20:])
    assert not (idx.cpu()[20:] == 5).any()


def test_gather_cdf_rows(dev, ops, priors):
    _, tf = priors
    table = O.renorm_cast_cdf(tf, 16, 0)
    idx = torch.randint(0, K, (37,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
    want = O.gather_cdf(table, idx)
    got = ops.gather_cdf(table.to(dev), idx.to(dev)).cpu()
    assert torch.equal(got, want)
    # many locations (several chunks of the prior-ordering pass), uneven prior usage, one prior only
    g3 = torch.Generator().manual_seed(4)
    big = torch.randint(0, K, (3001,), generator=g3, dtype=torch.int32)
    big[torch.rand(3001, generator=g3) < 0.7] = 5
    for case in (big, torch.full((1500,), K - 1, dtype=torch.int32)):
        assert torch.equal(ops.gather_cdf(table.to(dev), case.to(dev)).cpu(), O.gather_cdf(table, case))
    # a shape whose per-prior block is not a multiple of 16 bytes takes the scalar path
    t2 = table[:, :3, :].contiguous()
    assert torch.equal(ops.gather_cdf(t2.to(dev), idx.to(dev)).cpu(), O.gather_cdf(t2, idx))


@pytest.fixture(scope="module")
def coder_case(priors):
    _, tf = priors
    table = O.renorm_cast_cdf(tf, 16, 0)
    n_img, h, w = 3, 12, 10
    yl = synth.synthetic_latent(11, n_img, M, h, w)
    total = O.prior_bits_lut(yl, tf)
    _, idx = O.competition(total)                      # [B,h,w]
    sym = O.symbols_from_latent(yl)                    # [B, hw*M]
    return table, idx.reshape(n_img, -1).to(torch.int32), sym, h * w


@pytest.mark.parametrize("n_streams", [1, 5, 40, 120])
def test_rans_encode_matches_spec_and_roundtrips(dev, ops, coder_case, n_streams):
    table, idx, sym, n_loc = coder_case
    n_img = idx.shape[0]
    lps = -(-n_loc // n_streams)
    n_streams = -(-n_loc // lps)
    out, nbytes, status = ops.rans_encode(table.to(dev), idx.reshape(-1).to(dev), sym.reshape(-1).to(dev), n_img, n_streams, lps)
    assert status.cpu().tolist() == [0] * n_img
    nb = nbytes.cpu().tolist()
    tnp = table.numpy()
    for i in range(n_img):
        want = coders.mrs1_encode(tnp, idx[i].numpy(), sym[i].numpy(), n_streams, lps * M)
        got = out[i, :nb[i]].cpu().numpy().tobytes()
        assert got == want, f"image {i}: GPU bytes differ from the scalar specification"
        assert np.array_equal(coders.mrs1_decode(tnp, idx[i].numpy(), got), sym[i].numpy())
    dec, st = ops.rans_decode(table.to(dev), idx.reshape(-1).to(dev), out, nbytes, n_img, n_streams, lps)
    assert st.cpu().tolist() == [0] * n_img
    assert torch.equal(dec.cpu(), sym)
    # the decoder writing the fp16 latent itself == its symbols put through symbols_to_latent_f16
    lat, st = ops.rans_decode_latent_f16(table.to(dev), idx.reshape(-1).to(dev), out, nbytes, n_img, n_streams, lps, -127, 1)
    assert st.cpu().tolist() == [0] * n_img
    assert torch.equal(lat.view(-1), ops.symbols_to_latent_f16(dec.view(-1), -127, 1).view(-1))
    assert torch.equal(lat.cpu().float(), sym.float() - 127.0)


def test_rans_size_vs_range_coder(dev, ops, coder_case):
    table, idx, sym, n_loc = coder_case
    tnp = table.numpy()
    est = coders.ideal_bits(tnp, idx[0].numpy(), sym[0].numpy()) / 8
    n_streams, lps = ops.choose_streams(n_loc, M, est)
    out, nbytes, _ = ops.rans_encode(table.to(dev), idx[:1].reshape(-1).to(dev), sym[:1].reshape(-1).to(dev), 1, n_streams, lps)
    ref = len(coders.range_encode_table(tnp, idx[0].numpy(), sym[0].numpy()))
    got = int(nbytes.cpu()[0])
    print(f"rANS {got} B in {n_streams} streams vs range coder {ref} B: {100 * (got - ref) / ref:+.3f} %")
    assert abs(got - ref) / ref < 0.005


def test_rans_decode_flags_corruption(dev, ops, coder_case):
    table, idx, sym, n_loc = coder_case
    out, nbytes, _ = ops.rans_encode(table.to(dev), idx[:1].reshape(-1).to(dev), sym[:1].reshape(-1).to(dev), 1, 4, 30)
    bad = out.clone()
    bad[0, 0] ^= 0xFF                                  # magic
    _, st = ops.rans_decode(table.to(dev), idx[:1].reshape(-1).to(dev), bad, nbytes, 1, 4, 30)
    assert st.cpu()[0] < 0
    bad = out.clone()
    bad[0, 40:60] = 0                                   # payload
    dec, st = ops.rans_decode(table.to(dev), idx[:1].reshape(-1).to(dev), bad, nbytes, 1, 4, 30)
    assert st.cpu()[0] < 0 or not torch.equal(dec.cpu(), sym[:1])


def test_rans_decode_refuses_crafted_geometry(dev, ops, coder_case):
    """A header can claim anything: a stream count that would wrap the 32-bit header size, or more bytes than the
    buffer holds, must end in an error or a negative status, never in reads outside the buffer."""
    from compression_amd import MpcError
    table, idx, sym, n_loc = coder_case
    out, nbytes, _ = ops.rans_encode(table.to(dev), idx[:1].reshape(-1).to(dev), sym[:1].reshape(-1).to(dev), 1, 4, 30)
    args = (table.to(dev), idx[:1].reshape(-1).to(dev), out, nbytes, 1)
    with pytest.raises(MpcError, match="more streams than locations"):
        ops.rans_decode(*args, 2 ** 31 - 8, 30)
    with pytest.raises(MpcError):
        ops.rans_decode(*args, n_loc, 1 << 20)                   # stream longer than the 16-bit length field allows
    lying = torch.full_like(nbytes, out.shape[1] + 4096)           # claims more bytes than one image's slot
    _, st = ops.rans_decode(args[0], args[1], out, lying, 1, 4, 30)
    assert st.cpu().tolist() == [-20]
    streams = [out[0, :int(nbytes.cpu()[0])].cpu().numpy().tobytes()]
    forged = bytearray(streams[0]); forged[4:8] = (2 ** 31 - 8).to_bytes(4, "little")
    with pytest.raises(ValueError, match="streams for"):
        ops.rans_decode_from_bytes(args[0], args[1], [bytes(forged)])


def test_rans_rejects_bad_symbols(dev, ops, coder_case):
    table, idx, sym, n_loc = coder_case
    s = sym[:1].clone().reshape(-1)
    s[7] = 300
    _, _, status = ops.rans_encode(table.to(dev), idx[:1].reshape(-1).to(dev), s.to(dev), 1, 4, 30)
    assert status.cpu()[0] < 0


def test_symbols_latent_roundtrip_and_clamp(dev, ops):
    y = torch.tensor([-130.0, -127.0, 0.0, 1.0, 128.0, 140.0]).to(dev)
    sym, nclamp = ops.latent_to_symbols(y, -127, 1, 256)
    assert sym.cpu().tolist() == [0, 0, 127, 128, 255, 255] and int(nclamp.cpu()) == 2
    back = ops.symbols_to_latent(sym, -127, 1)
    assert back.cpu().tolist() == [-127.0, -127.0, 0.0, 1.0, 128.0, 128.0]


def test_laplace_bits_and_rows(dev, ops):
    g = torch.Generator().manual_seed(9)
    sigma = torch.exp(torch.randn(4000, generator=g) * 1.5)
    y = torch.round(torch.randn(4000, generator=g) * sigma)
    want, _ = O.laplace_bits(y, sigma)
    got = ops.laplace_bits_total(y.to(dev), sigma.to(dev)).cpu().item()
    assert got == pytest.approx(want.item(), rel=2e-6)
    rows_want = O.laplace_cdf_rows(sigma[:64])
    rows_got = ops.laplace_cdf_rows(sigma[:64].to(dev), -127.5, 257).cpu()
    d = (rows_got.int() - rows_want.int()).abs()
    print(f"laplace rows: {(d != 0).float().mean().item():.2e} entries differ")
    assert d.max().item() <= 1 and (d != 0).float().mean().item() < 2e-3


@pytest.mark.parametrize("Lrow,Mch,n_img,n_loc,n_streams", [(257, 32, 2, 96, 5), (257, 8, 1, 40, 1), (200, 16, 3, 33, 4)])
def test_laplace_coder_equals_the_coder_over_materialised_rows(dev, ops, Lrow, Mch, n_img, n_loc, n_streams):
    """The coder that evaluates the per-element Laplace(0, sigma) rows where it needs them (mpc_rans_encode_laplace /
    mpc_rans_decode_laplace) against the coder run over the same rows materialised as a [n_img*n_loc, M, L] table with
    one "prior" per location (what the reference hands to torchac, abstract_model.py:409-423): the containers are the
    same bytes, both decoders return the coded symbols.  Ragged cases: a row shorter than 257, a last stream that is
    shorter than the others, one stream only, scales from 1e-3 (all mass on one symbol) to 60."""
    g = torch.Generator().manual_seed(Lrow + n_loc)
    n = n_img * n_loc * Mch
    sigma = torch.exp(torch.rand(n, generator=g) * 11.0 - 6.9)
    sigma[::97] = 0.0                                       # clamped to 1e-10: a step function
    S = Lrow - 1
    centre = S // 2 - 1                                     # symbol of the value 0; targets first + s are the bin edges,
    first = -centre - 0.5                                   # as in the model (min_feat - 0.5 with min_feat = -127)
    y = torch.round(torch.randn(n, generator=g) * sigma.clamp(max=40.0))
    sym = (y + centre).clamp(0, S - 1).to(torch.int16)
    sym[5], sym[6] = 0, S - 1                               # both ends of the row
    sig_d, sym_d = sigma.to(dev), sym.to(dev)
    lps = -(-n_loc // n_streams)
    out, nbytes, status = ops.rans_encode_laplace(sig_d, sym_d, n_img, Mch, first, Lrow, 16, n_streams, lps)
    assert status.cpu().tolist() == [0] * n_img
    # the same rows, materialised, through the table coder: one table row set per location of the whole batch
    rows = ops.laplace_cdf_rows(sig_d, first, Lrow, 16).view(n_img * n_loc, Mch, Lrow)
    idx = torch.arange(n_img * n_loc, dtype=torch.int32, device=dev)
    out_t, nbytes_t, status_t = ops.rans_encode(rows, idx, sym_d, n_img, n_streams, lps)
    assert status_t.cpu().tolist() == [0] * n_img
    nb = nbytes.cpu().tolist()
    assert nb == nbytes_t.cpu().tolist()
    for i in range(n_img):
        assert torch.equal(out[i, :nb[i]], out_t[i, :nb[i]])
    got, st = ops.rans_decode_laplace(sig_d, out, nbytes, n_img, Mch, first, Lrow, 16, n_streams, lps)
    assert st.cpu().tolist() == [0] * n_img and torch.equal(got.view(-1), sym_d)
    got_t, st_t = ops.rans_decode(rows, idx, out_t, nbytes_t, n_img, n_streams, lps)
    assert torch.equal(got_t.view(-1), sym_d)
    # a damaged payload is reported, not decoded silently
    bad = out.clone()
    bad[0, 16 + 2 * n_streams + 3] ^= 0x5A
    _, st_bad = ops.rans_decode_laplace(sig_d, bad, nbytes, n_img, Mch, first, Lrow, 16, n_streams, lps)
    got_bad, _ = ops.rans_decode_laplace(sig_d, bad, nbytes, n_img, Mch, first, Lrow, 16, n_streams, lps)
    assert st_bad.cpu()[0].item() < 0 or not torch.equal(got_bad.view(-1), sym_d)
    # a symbol outside the row is refused by the encoder
    sym_bad = sym_d.clone()
    sym_bad[3] = S
    _, _, st_enc = ops.rans_encode_laplace(sig_d, sym_bad, n_img, Mch, first, Lrow, 16, n_streams, lps)
    assert st_enc.cpu()[0].item() == -10
    with pytest.raises(Exception, match="257"):
        ops.rans_encode_laplace(sig_d, sym_d, n_img, Mch, first, 300, 16, n_streams, lps)


def test_fast_decoder_outside_its_row_window(dev, ops):
    """The 256-symbol decoder works on compact rows: the 16 segment heads and the 64 entries of the four heaviest
    consecutive segments; a state that falls into any other segment takes the segment from the full table.  Here most
    symbols do: wide rows (Laplace scales 20 .. 200, and exactly uniform rows) coded with symbols drawn uniformly from
    all 256 values, the window at either end of the row for some rows.  The GPU bytes equal the scalar specification's,
    the independent C decoder and the GPU decoder both return the symbols."""
    K2, M2, n_img, n_loc, n_streams = 4, 32, 2, 50, 7
    g = torch.Generator().manual_seed(77)
    t = torch.linspace(-127.5, 128.5, 257)
    scale = torch.exp(torch.rand(K2, M2, 1, generator=g) * 2.3 + 3.0)
    centre = (torch.rand(K2, M2, 1, generator=g) - 0.5) * 300.0          # modes beyond both ends of the row as well
    tf = 0.5 - 0.5 * torch.sign(t - centre) * torch.expm1(-(t - centre).abs() / scale)
    tf[0, :4] = torch.linspace(0.0, 1.0, 257)                            # uniform rows: every window holds the same mass
    table = O.renorm_cast_cdf(tf.contiguous(), 16, 0)
    idx = torch.randint(0, K2, (n_img, n_loc), generator=g, dtype=torch.int32)
    sym = torch.randint(0, 256, (n_img, n_loc * M2), generator=g, dtype=torch.int16)
    lps = -(-n_loc // n_streams)
    td, idxd = table.to(dev), idx.reshape(-1).to(dev)
    out, nbytes, status = ops.rans_encode(td, idxd, sym.reshape(-1).to(dev), n_img, n_streams, lps)
    assert status.cpu().tolist() == [0] * n_img
    nb = nbytes.cpu().tolist()
    tnp = table.numpy()
    for i in range(n_img):
        got = out[i, :nb[i]].cpu().numpy().tobytes()
        assert got == coders.mrs1_encode(tnp, idx[i].numpy(), sym[i].numpy(), n_streams, lps * M2)
        assert np.array_equal(coders.mrs1_decode(tnp, idx[i].numpy(), got), sym[i].numpy())
    dec, st = ops.rans_decode(td, idxd, out, nbytes, n_img, n_streams, lps)
    assert st.cpu().tolist() == [0] * n_img
    assert torch.equal(dec.cpu(), sym)


@pytest.mark.parametrize("M2,n_loc,n_streams", [(10, 37, 1), (24, 64, 3), (48, 41, 2), (16, 200, 1), (256, 9, 2)])
def test_fast_decoder_channel_counts_and_long_streams(dev, ops, M2, n_loc, n_streams):
    """The 256-symbol decoder loads the rows of 16 symbols per pass: with a channel count that is a multiple of 16 a pass
    lies inside one location (one address, fixed offsets), otherwise every symbol's row is addressed on its own and the
    last pass of a stream is short.  Streams of up to 3 200 symbols: several hundred renormalisation words, which go
    through the decoder's three word registers many times.  Narrow and wide rows mixed, symbols drawn from the rows'
    own distributions.  Bytes against the scalar specification, symbols from both decoders."""
    K2, n_img = 5, 2
    g = torch.Generator().manual_seed(1000 + M2)
    t = torch.linspace(-127.5, 128.5, 257)
    scale = torch.exp(torch.rand(K2, M2, 1, generator=g) * 5.0 - 2.0)    # 0.14 .. 20
    centre = torch.round((torch.rand(K2, M2, 1, generator=g) - 0.5) * 60.0)
    tf = 0.5 - 0.5 * torch.sign(t - centre) * torch.expm1(-(t - centre).abs() / scale)
    table = O.renorm_cast_cdf(tf.contiguous(), 16, 0)
    idx = torch.randint(0, K2, (n_img, n_loc), generator=g, dtype=torch.int32)
    u = torch.rand(n_img, n_loc, M2, generator=g) - 0.5
    rows_scale, rows_centre = scale[idx.long(), :, 0], centre[idx.long(), :, 0]
    y = rows_centre + torch.round(-rows_scale * torch.sign(u) * torch.log1p(-2 * u.abs()))
    sym = (y + 127).clamp(0, 255).to(torch.int16).reshape(n_img, -1)
    lps = -(-n_loc // n_streams)
    n_streams = -(-n_loc // lps)
    td, idxd = table.to(dev), idx.reshape(-1).to(dev)
    out, nbytes, status = ops.rans_encode(td, idxd, sym.reshape(-1).to(dev), n_img, n_streams, lps)
    assert status.cpu().tolist() == [0] * n_img
    nb = nbytes.cpu().tolist()
    tnp = table.numpy()
    for i in range(n_img):
        got = out[i, :nb[i]].cpu().numpy().tobytes()
        assert got == coders.mrs1_encode(tnp, idx[i].numpy(), sym[i].numpy(), n_streams, lps * M2)
        assert np.array_equal(coders.mrs1_decode(tnp, idx[i].numpy(), got), sym[i].numpy())
    dec, st = ops.rans_decode(td, idxd, out, nbytes, n_img, n_streams, lps)
    assert st.cpu().tolist() == [0] * n_img
    assert torch.equal(dec.cpu(), sym)
    # one word less than the stream needs: reported, whatever the state does afterwards
    short = nbytes.clone()
    short[0] -= 2
    _, st = ops.rans_decode(td, idxd, out, short, n_img, n_streams, lps)
    assert st.cpu()[0].item() < 0 and st.cpu()[1].item() == 0
