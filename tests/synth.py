"""Deterministic synthetic weights and inputs shared by the golden-vector generator, the tests and
bench.py.  Everything is derived from explicit torch.Generator seeds on the CPU, so the reference
(in the build container) and the HIP path (on the GPU box) see bit-identical tensors.

Weight statistics follow the reference's initialisers (Xavier-normal gains of
Balle2018PT_compressor.py:82-96,286-300; biases 0.01; GDN beta/gamma of GDN.py:47-63) with two
deliberate departures that make the tests sharper:
  * GDN gamma gets positive off-diagonal noise (the default off-diagonal is exactly zero after the
    reparametrisation, which would hide a transposed or permuted gamma);
  * the 64 priors are spread (per-prior scale ramp on f1.h, random offsets), because with the default
    N(0, 0.01) init every prior is the same and one prior wins everywhere.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def _xavier_normal(g, shape, gain):
    rf = 1
    for d in shape[2:]:
        rf *= d
    fan_in, fan_out = shape[1] * rf, shape[0] * rf
    std = gain * math.sqrt(2.0 / (fan_in + fan_out))
    return torch.randn(shape, generator=g) * std


def _gdn(g, ch):
    pedestal = (2.0 ** -18) ** 2
    beta = torch.sqrt(torch.ones(ch) + pedestal) * (1.0 + 0.2 * torch.rand(ch, generator=g))
    gamma = torch.sqrt(0.1 * torch.eye(ch) + pedestal) + 0.02 * torch.rand(ch, ch, generator=g)
    return beta, gamma


def transform_weights(seed: int, N: int = 192, M: int = 256, testolina: bool = False) -> Dict[str, torch.Tensor]:
    """Encoder.* / Decoder.* tensors with the reference's state_dict names and shapes."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    enc = [(N, 3, math.sqrt(2 * (3 + N) / 6)), (N, N, math.sqrt(2)), (N, N, math.sqrt(2)),
           (M, N, math.sqrt(2 * (M + N) / (2 * N)))]
    for i, (co, ci, gain) in enumerate(enc, 1):
        sd[f"Encoder.conv{i}.weight"] = _xavier_normal(g, (co, ci, 5, 5), gain)
        sd[f"Encoder.conv{i}.bias"] = torch.full((co,), 0.01) + 0.01 * torch.randn(co, generator=g)
        if i < 4:
            sd[f"Encoder.gdn{i}.beta"], sd[f"Encoder.gdn{i}.gamma"] = _gdn(g, N)
    if testolina:
        for i, ci in enumerate((M, N, N, N), 1):
            sd[f"Decoder.convred{i}.weight"] = _xavier_normal(g, (N, ci, 5, 5), math.sqrt(2))
            sd[f"Decoder.convred{i}.bias"] = torch.full((N,), 0.01) + 0.01 * torch.randn(N, generator=g)
    d1_in = N if testolina else M
    dec = [(d1_in, N, math.sqrt(2 * (M + N) / (2 * M))), (N, N, math.sqrt(2)), (N, N, math.sqrt(2)),
           (N, 3, math.sqrt(2 * (N + 3) / (2 * N)))]
    for i, (ci, co, gain) in enumerate(dec, 1):
        sd[f"Decoder.deconv{i}.weight"] = _xavier_normal(g, (ci, co, 5, 5), gain)
        sd[f"Decoder.deconv{i}.bias"] = torch.full((co,), 0.01) + 0.01 * torch.randn(co, generator=g)
        if i < 4:
            sd[f"Decoder.igdn{i}.beta"], sd[f"Decoder.igdn{i}.gamma"] = _gdn(g, N)
    return sd


def prior_weights(seed: int, K: int = 64, M: int = 256, spread: bool = True) -> Dict[str, torch.Tensor]:
    """bitEstimators.f{1..4}.{h,b,a} of shape [K,1,M,1,1]."""
    g = torch.Generator().manual_seed(seed)
    shape = (K, 1, M, 1, 1)
    sd = {}
    for i in (1, 2, 3, 4):
        std = 0.5 if spread else 0.01
        sd[f"bitEstimators.f{i}.h"] = std * torch.randn(shape, generator=g)
        sd[f"bitEstimators.f{i}.b"] = std * torch.randn(shape, generator=g)
        if i < 4:
            sd[f"bitEstimators.f{i}.a"] = std * torch.randn(shape, generator=g)
    if spread:
        ramp = torch.linspace(-3.0, 1.5, K).view(K, 1, 1, 1, 1)
        sd["bitEstimators.f1.h"] = ramp + 0.3 * torch.randn(shape, generator=g)
        sd["bitEstimators.f1.b"] = 1.5 * torch.randn(shape, generator=g)
    return sd


def hyperprior_weights(seed: int, N: int = 192, M: int = 256) -> Dict[str, torch.Tensor]:
    """priorEncoder.*, priorDecoder.*, bitEstimator_z.* (Balle2018PT_compressor.py:105-127,223-251;
    bitEstimator.py:9-45)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    sd["priorEncoder.conv1.weight"] = _xavier_normal(g, (N, M, 3, 3), math.sqrt(2 * (M + N) / (2 * M)))
    sd["priorEncoder.conv2.weight"] = _xavier_normal(g, (N, N, 5, 5), math.sqrt(2))
    sd["priorEncoder.conv3.weight"] = _xavier_normal(g, (N, N, 5, 5), math.sqrt(2))
    sd["priorDecoder.deconv1.weight"] = _xavier_normal(g, (N, N, 5, 5), math.sqrt(2))
    sd["priorDecoder.deconv2.weight"] = _xavier_normal(g, (N, N, 5, 5), math.sqrt(2))
    sd["priorDecoder.deconv3.weight"] = _xavier_normal(g, (N, M, 3, 3), math.sqrt(2 * (M + N) / (2 * N)))
    for k in list(sd):
        co = sd[k].shape[1] if "deconv" in k else sd[k].shape[0]
        sd[k.replace("weight", "bias")] = torch.full((co,), 0.01) + 0.01 * torch.randn(co, generator=g)
    for i in (1, 2, 3, 4):
        sd[f"bitEstimator_z.f{i}.h"] = 0.3 * torch.randn(1, N, 1, 1, generator=g)
        sd[f"bitEstimator_z.f{i}.b"] = 0.3 * torch.randn(1, N, 1, 1, generator=g)
        if i < 4:
            sd[f"bitEstimator_z.f{i}.a"] = 0.3 * torch.randn(1, N, 1, 1, generator=g)
    sd["bitEstimator_z.f1.h"] = sd["bitEstimator_z.f1.h"] - 1.0
    return sd


def manypriors_state_dict(seed: int = 234, N: int = 192, M: int = 256, K: int = 64, spread: bool = True,
                          testolina: bool = False) -> Dict[str, torch.Tensor]:
    sd = transform_weights(seed, N, M, testolina)
    sd.update(prior_weights(seed + 1, K, M, spread))
    return sd


def hyperprior_state_dict(seed: int = 234, N: int = 192, M: int = 256) -> Dict[str, torch.Tensor]:
    sd = transform_weights(seed, N, M)
    sd.update(hyperprior_weights(seed + 2, N, M))
    return sd


def rand_image(seed: int, B: int, H: int, W: int) -> torch.Tensor:
    """Uniform noise in [0,1): the reference test harness' worst case (high-entropy latents)."""
    return torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(seed))


def smooth_image(seed: int, B: int, H: int, W: int) -> torch.Tensor:
    """Low-pass noise + a little grain: image-like latents and bit-rates."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(B, 3, max(H // 8, 1), max(W // 8, 1), generator=g)
    up = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)
    return (up + 0.02 * torch.randn(B, 3, H, W, generator=g)).clamp(0.0, 1.0)


def noisy_image(clean: torch.Tensor, seed: int, poisson_scale: float = 0.2 ** 2,
                uniform_scale: float = 0.04 ** 2) -> torch.Tensor:
    """Synthetic noisy input for the JDC config, following the recipe of src/dc/libs/dcdatasets.py:82-101
    (Poisson shot noise on a scaled image plus a uniform read-noise term, clipped to [0,1])."""
    g = torch.Generator().manual_seed(seed)
    chi = 1.0 / max(poisson_scale, 1e-8)
    shot = torch.poisson(torch.clamp_min(clean * chi, 0), generator=g) / chi
    return (shot + math.sqrt(uniform_scale) * torch.rand(clean.shape, generator=g)).clamp(0.0, 1.0)


def synthetic_latent(seed: int, B: int, M: int, h: int, w: int, min_feat: int = -127,
                     max_feat: int = 128) -> torch.Tensor:
    """Integer latents with a per-location scale between 0.3 and 30, for the entropy-side kernels."""
    g = torch.Generator().manual_seed(seed)
    sigma = torch.exp(torch.rand(B, 1, h, w, generator=g) * 4.6 - 1.2)
    return torch.round(torch.randn(B, M, h, w, generator=g) * sigma).clamp(min_feat, max_feat)
