"""Per-image squared error and the device-side statistics sums (what the multi-GPU driver all-reduces) against the
host implementation of the same sums (compression_amd.distributed.StatsAccumulator.add_image, which follows
src/compression/tests.py:393-460)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_per_image_sse_and_device_statistics():
    assert torch.cuda.is_available(), "this test needs the MI355X"
    from compression_amd import ops
    from compression_amd.distributed import StatsAccumulator
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, H, W = 5, 40, 53                                     # 3*H*W is not a multiple of 4: the scalar tail runs too
    recon = torch.rand(B, 3, H, W, generator=g) * 1.4 - 0.2
    target = torch.rand(B, 3, H, W, generator=g)
    xhat, sse = ops.clamp_sse_per_image(recon.to(dev), target.to(dev))
    assert torch.equal(xhat.cpu(), recon.clamp(0, 1))
    want = ((recon.clamp(0, 1) - target).double() ** 2).sum((1, 2, 3))
    assert torch.allclose(sse.cpu(), want, rtol=1e-6)
    _, sse_raw = ops.clamp_sse_per_image(recon.to(dev), target.to(dev), score_clamped=False)
    assert torch.allclose(sse_raw.cpu(), ((recon - target).double() ** 2).sum((1, 2, 3)), rtol=1e-6)
    _, one = ops.clamp_sse(recon.to(dev), target.to(dev))
    assert float(one) == pytest.approx(float(sse_raw.sum()), rel=1e-9)

    ms = torch.tensor([0.91, 0.95, 0.99, 1.0, 0.5], dtype=torch.float64)
    host = StatsAccumulator()
    for i in range(B):
        host.add_image(bits=8000.0 / B, pixels=H * W, sq_err=float(want[i]), nbytes=1000.0 / B, seconds=0.25 / B,
                       side_bits=40.0 / B, msssim=float(ms[i]))
    acc = StatsAccumulator(device=dev)
    acc.vec += 7.0                                           # reset must discard this
    acc.add_batch_device(sse, 3 * H * W, bits=8000.0, pixels=B * H * W, nbytes=1000.0, seconds=0.25, side_bits=40.0,
                         msssim=ms.to(dev), reset=True)
    got, ref = acc.result(), host.result()
    for k in ref:
        assert got[k] == pytest.approx(ref[k], rel=1e-9), k
    acc.add_batch_device(sse, 3 * H * W, bits=8000.0, pixels=B * H * W, nbytes=1000.0, seconds=0.25, side_bits=40.0,
                         msssim=ms.to(dev))
    twice = acc.result()
    assert twice["images"] == 2 * B and twice["psnr_sum"] == pytest.approx(2 * ref["psnr_sum"], rel=1e-9)
    assert twice["psnr"] == pytest.approx(ref["psnr"], rel=1e-9)
