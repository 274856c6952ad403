"""SSIM / MS-SSIM of the reference's test harness on the device (mirror of src/common/extlibs/pt_ms_ssim.py:
`ssim`, `ms_ssim`, same arguments and return values) plus the per-image statistics test_dir accumulates
(src/compression/tests.py:393-418).  The filtering and the reductions are HIP kernels (mpc_ssim_stats,
mpc_avg_pool2); the final combination of the 5 x B x C scale values is done on the host."""
import math

import numpy as np
import torch

from . import ops

DEFAULT_WEIGHTS = [0.0448, 0.2856, 0.3001, 0.2363, 0.1333]


def _fspecial_gauss_1d(size, sigma):
    """1-D Gaussian window, float32, as pt_ms_ssim.py:8-23 builds it."""
    coords = torch.arange(size).to(dtype=torch.float)
    coords -= size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g /= g.sum()
    return g


def _check(X, Y, win_size):
    if len(X.shape) != 4:
        raise ValueError("Input images should be 4-d tensors.")
    if X.dtype != Y.dtype:
        raise ValueError("Input images should have the same dtype.")
    if X.shape != Y.shape:
        raise ValueError("Input images should have the same dimensions.")
    if win_size % 2 != 1:
        raise ValueError("Window size should be odd.")


def _window(win, win_size, win_sigma):
    if win is not None:
        return win.reshape(-1, win.shape[-1])[0].float().cpu()      # the reference repeats one window per channel
    return _fspecial_gauss_1d(win_size, win_sigma)


def ssim(X, Y, data_range=1.0, size_average=True, win_size=11, win_sigma=1.5, win=None, K=(0.01, 0.03),
         nonnegative_ssim=True):
    """pt_ms_ssim.py:87-141."""
    if win is not None:
        win_size = win.shape[-1]
    _check(X, Y, win_size)
    w = _window(win, win_size, win_sigma)
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    ssim_pc, _ = ops.ssim_stats(X.contiguous(), Y.contiguous(), w.tolist(), C1, C2)
    ssim_pc = ssim_pc.to(torch.float32)
    if nonnegative_ssim:
        ssim_pc = torch.relu(ssim_pc)
    return ssim_pc.mean() if size_average else ssim_pc.mean(1)


def ms_ssim(X, Y, data_range=1.0, size_average=True, win_size=11, win_sigma=1.5, win=None, weights=None, K=(0.01, 0.03)):
    """pt_ms_ssim.py:144-216."""
    if win is not None:
        win_size = win.shape[-1]
    _check(X, Y, win_size)
    smaller_side = min(X.shape[-2:])
    assert smaller_side > (win_size - 1) * (2 ** 4), \
        "Image size should be larger than %d due to the 4 downsamplings in ms-ssim" % ((win_size - 1) * (2 ** 4))
    if weights is None:
        weights = DEFAULT_WEIGHTS
    wts = torch.tensor(weights, dtype=torch.float32, device=X.device)
    w = _window(win, win_size, win_sigma).tolist()
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    X, Y = X.contiguous(), Y.contiguous()
    levels = wts.shape[0]
    mcs = []
    ssim_pc = None
    for i in range(levels):
        ssim_pc, cs = ops.ssim_stats(X, Y, w, C1, C2)
        if i < levels - 1:
            mcs.append(torch.relu(cs.to(torch.float32)))
            X, Y = ops.avg_pool2(X), ops.avg_pool2(Y)
    ssim_pc = torch.relu(ssim_pc.to(torch.float32))
    mcs_and_ssim = torch.stack(mcs + [ssim_pc], dim=0)                      # (level, batch, channel)
    val = torch.prod(mcs_and_ssim ** wts.view(-1, 1, 1), dim=0)
    return val.mean() if size_average else val.mean(1)


def image_stats(reconstructed_image, gt, bpp=None, train_lambda=None, visual_loss=None):
    """The per-image numbers test_dir logs and sums (tests.py:393-418): psnr, msssim, msssimDB and, when bpp,
    train_lambda and visual_loss are given, combined_loss."""
    _, sse = ops.clamp_sse(reconstructed_image.contiguous(), gt.contiguous())
    mse = float(sse.item()) / reconstructed_image.numel()
    res = {"psnr": 10.0 * (math.log(1.0 / mse) / np.log(10))}
    try:
        m = float(ms_ssim(reconstructed_image, gt, data_range=1.0, size_average=True))
        res["msssim"] = m
        res["msssimDB"] = -10.0 * (math.log(1 - m) / np.log(10))
    except AssertionError as e:
        print(e)
    if bpp is not None and train_lambda is not None and visual_loss is not None:
        res["combined_loss"] = train_lambda * visual_loss + bpp
    return res
