"""Full-size cases of the benchmark configurations, checked through size-independent properties (the CPU oracle
would take minutes here): lossless encode -> bytes -> decode of every image, agreement of the fast arithmetic
(f16x3 analysis, fp16 synthesis, row-tiled last layer, 8-wave workgroups) with the exact f32 path, and
batch-composition independence (an image codes to the same bytes alone and inside a batch)."""
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
N, M, K = 192, 256, 64


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _model(dev, **kw):
    import compression_amd as ca
    m = ca.init_model("ManyPriors", dev, out_channel_N=N, out_channel_M=M, num_distributions=K, **kw)
    m.load_state_dict(synth.manypriors_state_dict(234, N, M, K, spread=True))
    return m.eval()


def test_kodak_batch_24_images(dev):
    """24 x 768x512 (the benchmark's step): fast path vs exact path."""
    from compression_amd import ops
    fast = _model(dev, analysis_precision="f16x3", synthesis_precision="fp16")
    exact = _model(dev, analysis_precision="fp32", synthesis_precision="fp32")
    x = synth.smooth_image(77, 24, 512, 768).to(dev)
    sym_f, idx_f, h, w = fast.encode_batch(x, entropy_coding=False)
    sym_e, idx_e, _, _ = exact.encode_batch(x, entropy_coding=False)
    flips = (sym_f != sym_e).float().mean().item()
    print(f"f16x3 vs f32 analysis: {flips:.2e} of the symbols differ")
    assert flips <= 1e-5                                           # rounding ties only
    assert (idx_f != idx_e).float().mean().item() <= 1e-3          # an index can move only where a symbol flipped
    streams, idx, h, w = fast.encode_batch(x)
    assert len(streams) == 24 and all(s[:4] == b"MRS1" for s in streams)
    # lossless: the decoder's symbols are the encoder's
    table = fast.entropy_table
    dec = ops.rans_decode_from_bytes(table, idx.reshape(-1).to(torch.int32), streams)
    assert torch.equal(dec.view(-1), sym_f.reshape(-1))
    # the coded size follows the competition's estimate (ln-form LUT -> bits)
    recon = fast.decode_batch(streams, idx, h, w)
    recon_e = exact.decode_batch(streams, idx, h, w)
    err = (recon.clamp(0, 1) - recon_e.clamp(0, 1)).abs().max().item()
    print(f"fp16 vs f32 synthesis on 24 images: max |dx| = {err:.2e}")
    assert err < 1e-2
    # an image codes to the same bytes alone as inside the batch
    alone, _, _, _ = fast.encode_batch(x[5:6])
    assert alone[0] == streams[5]


def test_4k_image_round_trip(dev):
    """3840 x 2112 (the 4K configuration after the harness' crop to multiples of 64): one image, 31 680 locations,
    more streams than the 16-bit length field alone would need."""
    from compression_amd import ops
    m = _model(dev, analysis_precision="f16x3", synthesis_precision="fp16")
    x = synth.smooth_image(78, 1, 2112, 3840).to(dev)
    sym, idx, h, w = m.encode_batch(x, entropy_coding=False)
    assert (h, w) == (132, 240)
    streams, idx2, _, _ = m.encode_batch(x)
    assert torch.equal(idx, idx2)
    dec = ops.rans_decode_from_bytes(m.entropy_table, idx.reshape(-1).to(torch.int32), streams)
    assert torch.equal(dec.view(-1), sym.reshape(-1))
    recon = m.decode_batch(streams, idx, h, w)
    assert recon.shape == (1, 3, 2112, 3840) and bool(torch.isfinite(recon).all())
    out = m(x)
    bits = 8 * len(streams[0])
    est = float(out["bpp_feature"]) * 2112 * 3840
    print(f"4K: coded {bits} bits, estimated {est:.0f}")
    assert bits < est * 1.01 + 4096


def test_two_batches_in_flight_repeat_the_same_bits(dev):
    """The benchmark's step with two batches in flight (CodecPipeline), 120 times on one batch: every result equals
    the one-batch-at-a-time result, byte strings and reconstruction.  This is the regression test of the LDS-DMA
    completion wait: with the transforms' LDS-DMA transfers only waited for by vmcnt(0), 6 % of such steps had one
    wrong tile in a transposed layer once the coders of the other batch shared the CUs."""
    from compression_amd.pipeline import CodecPipeline
    m = _model(dev, analysis_precision="f16x3", synthesis_precision="fp16")
    x = synth.rand_image(1000, 24, 512, 768).to(dev)
    strings, idx, h, w = m.encode_batch(x)
    recon = m.decode_batch(strings, idx, h, w).clone()
    torch.cuda.synchronize()
    pipe = CodecPipeline(m, dev)
    bad = []

    def check(r):
        pipe.synchronize()
        if r[0] != strings or not torch.equal(r[1], recon):
            bad.append(r[3])

    for it in range(120):
        r = pipe.submit(x, tag=it)
        if r is not None:
            check(r)
    check(pipe.flush())
    assert not bad, f"steps {bad} differ from the one-batch-at-a-time result"


def test_transforms_side_by_side_give_the_same_bits(dev):
    """g_s of one batch started behind the SECOND layer of the next batch's g_a, so that the transposed layers share the
    CUs with the third and the last analysis layer (CodecPipeline starts it behind the third by default), 600 steps:
    every result equals the one-batch-at-a-time result.  Regression test of the patch kernels' barriers: when a wave
    could arrive at the tap's s_barrier with its ds_write of the next weight chunk still queued (no lgkmcnt(0) in front of
    the barrier), the waves that left the barrier first read the old chunk, and 3 of 1 000 steps had a few wrong symbols
    in one wave's 32 pixels of the last analysis layer -- only when another transform kernel kept the CU's LDS busy."""
    from compression_amd import manynets_compressor
    from compression_amd.pipeline import CodecPipeline
    m = _model(dev, analysis_precision="f16x3", synthesis_precision="fp16")
    x = synth.rand_image(1000, 24, 512, 768).to(dev)
    strings, idx, h, w = m.encode_batch(x)
    recon = m.decode_batch(strings, idx, h, w).clone()
    torch.cuda.synchronize()
    bad = []
    default = manynets_compressor.SYNTHESIS_MAY_START_BEHIND_LAYER
    manynets_compressor.SYNTHESIS_MAY_START_BEHIND_LAYER = 2
    try:
        pipe = CodecPipeline(m, dev)

        def check(r):
            pipe.synchronize()
            if r[0] != strings or not torch.equal(r[1], recon):
                bad.append(r[3])

        for it in range(600):
            r = pipe.submit(x, tag=it)
            if r is not None:
                check(r)
        check(pipe.flush())
    finally:
        manynets_compressor.SYNTHESIS_MAY_START_BEHIND_LAYER = default
    assert not bad, f"steps {bad} differ from the one-batch-at-a-time result"
