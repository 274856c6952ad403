"""Multi-GPU driver pieces: image sharding and the statistics all-reduce.

The reference has no distributed code (one process, one device).  The codec path has no cross-image state, so N
GPUs run N independent shards; the only exchange is the sum of the evaluation statistics that the reference's
test harness accumulates per image (src/compression/tests.py:439-460): bits, pixels, squared error, PSNR,
MS-SSIM (and its dB form), image count, bytes, seconds.  One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" for CPU rehearsal).
"""
import math
from typing import Dict, List

import torch

STAT_KEYS = ("bits", "pixels", "sq_err", "psnr_sum", "images", "bytes", "seconds", "side_bits", "msssim_sum",
             "msssim_db_sum")
VECTOR_LEN = 16


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin: item i belongs to rank i % world."""
    return list(range(rank, n_items, world))


class StatsAccumulator:
    """Running sums of per-image statistics; `all_reduce()` makes them global."""

    def __init__(self, device="cpu"):
        self.vec = torch.zeros(VECTOR_LEN, dtype=torch.float64, device=device)

    def add_image(self, bits: float, pixels: int, sq_err: float, nbytes: int = 0, seconds: float = 0.0,
                  side_bits: float = 0.0, msssim: float = 0.0):
        mse = sq_err / max(pixels * 3, 1)
        psnr = 10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf")
        row = dict(bits=bits, pixels=pixels, sq_err=sq_err, psnr_sum=psnr, images=1, bytes=nbytes, seconds=seconds,
                   side_bits=side_bits, msssim_sum=msssim,
                   msssim_db_sum=-10.0 * math.log10(1.0 - msssim) if 0.0 < msssim < 1.0 else 0.0)
        for i, k in enumerate(STAT_KEYS):
            self.vec[i] += row[k]

    def add_batch_device(self, sse, values_per_image: int, bits: float, pixels: float, nbytes: float = 0.0,
                         seconds: float = 0.0, side_bits: float = 0.0, msssim=None, reset: bool = False):
        """The same sums for a whole batch without a device-to-host copy: `sse` (and `msssim`) are float64 [B] tensors
        on the GPU (ops.clamp_sse_per_image, pt_ms_ssim.ms_ssim(size_average=False)); PSNR and MS-SSIM dB are formed
        per image on the device.  The accumulator's vector must live on that GPU."""
        from . import ops
        ops.stats_accumulate(self.vec, sse, values_per_image, bits, pixels, nbytes, seconds, side_bits, msssim, reset)

    def all_reduce(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.vec, op=dist.ReduceOp.SUM)
        return self

    def result(self) -> Dict[str, float]:
        v = {k: float(self.vec[i]) for i, k in enumerate(STAT_KEYS)}
        n = max(v["images"], 1.0)
        return {**v, "bpp": (v["bits"] + v["side_bits"]) / max(v["pixels"], 1.0), "psnr": v["psnr_sum"] / n,
                "msssim": v["msssim_sum"] / n, "msssimDB": v["msssim_db_sum"] / n,
                "mse": v["sq_err"] / max(v["pixels"] * 3, 1.0)}
