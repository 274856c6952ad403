"""CPU oracle for the Manypriors / JDC codec inference path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement (PyTorch-CPU fp32 /
numpy) of the reference algorithm.  It is imported by `tests/`, by
`__graft_entry__.smoke()` and by the `cpu_baseline` leg of `bench.py`, always as
the checker or the timed CPU baseline -- never by the product package
`compression_amd`, which must fail loudly when its HIP library is missing.

Pinning: every function below is checked against tensors captured from the
reference itself (imported in the build container by `tests/golden/make_golden.py`;
the captured tensors are committed under `tests/golden/`).  The one part that has
no reference output to pin against is the range coder (`torchac` is a third-party
C++ extension that is absent from the reference tree and from this image): the
coder in `oracle/range_coder.c` restates the published algorithm and its bytes
and sizes are "parity unpinned".

All functions are functional: weights come in as a dict with the reference's
`state_dict` key names (e.g. `Encoder.conv1.weight`, `Encoder.gdn1.beta`,
`bitEstimators.f1.h`).
"""
from __future__ import annotations

import lzma
import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# ----------------------------------------------------------------------------
# GDN  (reference: src/compression/models/GDN.py:47-95)
# ----------------------------------------------------------------------------
GDN_REPARAM_OFFSET = 2.0 ** -18
GDN_PEDESTAL = GDN_REPARAM_OFFSET ** 2
GDN_BETA_MIN = 1e-6
GDN_BETA_BOUND = (GDN_BETA_MIN + GDN_PEDESTAL) ** 0.5
GDN_GAMMA_BOUND = GDN_REPARAM_OFFSET


def gdn_effective(beta: torch.Tensor, gamma: torch.Tensor):
    """Lower-bound + square reparametrisation of (beta, gamma), GDN.py:75-81."""
    b = torch.max(beta, torch.ones_like(beta) * GDN_BETA_BOUND) ** 2 - GDN_PEDESTAL
    g = torch.max(gamma, torch.ones_like(gamma) * GDN_GAMMA_BOUND) ** 2 - GDN_PEDESTAL
    return b, g


def gdn(x: torch.Tensor, beta: torch.Tensor, gamma: torch.Tensor, inverse: bool):
    """y_i = x_i / sqrt(beta_i + sum_j gamma_ij x_j^2)  (or times, when inverse). GDN.py:83-91."""
    b, g = gdn_effective(beta, gamma)
    ch = x.shape[1]
    norm = torch.sqrt(F.conv2d(x ** 2, g.view(ch, ch, 1, 1), b))
    return x * norm if inverse else x / norm


# ----------------------------------------------------------------------------
# Transforms  (reference: src/compression/models/Balle2018PT_compressor.py)
# ----------------------------------------------------------------------------
def analysis(x: torch.Tensor, sd: SD, prefix: str = "Encoder.") -> torch.Tensor:
    """g_a: 4 x (conv 5x5 s2 p2), GDN after the first three. Balle2018PT_compressor.py:82-102."""
    p = prefix
    for i in (1, 2, 3):
        x = F.conv2d(x, sd[f"{p}conv{i}.weight"], sd[f"{p}conv{i}.bias"], stride=2, padding=2)
        x = gdn(x, sd[f"{p}gdn{i}.beta"], sd[f"{p}gdn{i}.gamma"], inverse=False)
    return F.conv2d(x, sd[f"{p}conv4.weight"], sd[f"{p}conv4.bias"], stride=2, padding=2)


def synthesis(y: torch.Tensor, sd: SD, prefix: str = "Decoder.") -> torch.Tensor:
    """g_s: 4 x (deconv 5x5 s2 p2 op1), IGDN after the first three. Balle2018PT_compressor.py:286-308."""
    p = prefix
    x = y
    for i in (1, 2, 3):
        x = F.conv_transpose2d(x, sd[f"{p}deconv{i}.weight"], sd[f"{p}deconv{i}.bias"],
                               stride=2, padding=2, output_padding=1)
        x = gdn(x, sd[f"{p}igdn{i}.beta"], sd[f"{p}igdn{i}.gamma"], inverse=True)
    return F.conv_transpose2d(x, sd[f"{p}deconv4.weight"], sd[f"{p}deconv4.bias"],
                              stride=2, padding=2, output_padding=1)


def testolina_synthesis(y: torch.Tensor, sd: SD, prefix: str = "Decoder.") -> torch.Tensor:
    """Testolina g_s: 4 x (conv 5x5 s1 p2 + ReLU) at latent resolution, then the 4 deconvs with
    IGDN.  testolina_compressor.py:18-100."""
    p = prefix
    x = y
    for i in (1, 2, 3, 4):
        x = F.relu(F.conv2d(x, sd[f"{p}convred{i}.weight"], sd[f"{p}convred{i}.bias"], stride=1, padding=2))
    for i in (1, 2, 3):
        x = F.conv_transpose2d(x, sd[f"{p}deconv{i}.weight"], sd[f"{p}deconv{i}.bias"],
                               stride=2, padding=2, output_padding=1)
        x = gdn(x, sd[f"{p}igdn{i}.beta"], sd[f"{p}igdn{i}.gamma"], inverse=True)
    return F.conv_transpose2d(x, sd[f"{p}deconv4.weight"], sd[f"{p}deconv4.bias"],
                              stride=2, padding=2, output_padding=1)


def hyper_analysis(y: torch.Tensor, sd: SD, prefix: str = "priorEncoder.") -> torch.Tensor:
    """h_a: |y| -> conv3 s1 -> relu -> conv5 s2 -> relu -> conv5 s2. Balle2018PT_compressor.py:105-127."""
    p = prefix
    x = torch.abs(y)
    x = F.relu(F.conv2d(x, sd[f"{p}conv1.weight"], sd[f"{p}conv1.bias"], stride=1, padding=1))
    x = F.relu(F.conv2d(x, sd[f"{p}conv2.weight"], sd[f"{p}conv2.bias"], stride=2, padding=2))
    return F.conv2d(x, sd[f"{p}conv3.weight"], sd[f"{p}conv3.bias"], stride=2, padding=2)


def hyper_synthesis(z: torch.Tensor, sd: SD, prefix: str = "priorDecoder.") -> torch.Tensor:
    """h_s ("TFCodeExp"): deconv5 s2 -> relu -> deconv5 s2 -> relu -> deconv3 s1 -> exp.
    Balle2018PT_compressor.py:223-251."""
    p = prefix
    x = F.relu(F.conv_transpose2d(z, sd[f"{p}deconv1.weight"], sd[f"{p}deconv1.bias"],
                                  stride=2, padding=2, output_padding=1))
    x = F.relu(F.conv_transpose2d(x, sd[f"{p}deconv2.weight"], sd[f"{p}deconv2.bias"],
                                  stride=2, padding=2, output_padding=1))
    return torch.exp(F.conv_transpose2d(x, sd[f"{p}deconv3.weight"], sd[f"{p}deconv3.bias"],
                                        stride=1, padding=1))


# ----------------------------------------------------------------------------
# Learned CDFs  (reference: src/compression/models/bitEstimator.py:9-45, 87-141)
# ----------------------------------------------------------------------------
def _bitparm(x, h, b, a):
    """One Bitparm layer: affine with softplus'd slope, tanh-gated residual, sigmoid on the last."""
    t = x * F.softplus(h) + b
    if a is None:
        return torch.sigmoid(t)
    return t + torch.tanh(t) * torch.tanh(a)


def multihead_cdf(x: torch.Tensor, sd: SD, prefix: str = "bitEstimators.") -> torch.Tensor:
    """K-head factorised CDF. x broadcasts against params of shape [K,1,M,1,1] -> [K,B,M,h,w].
    bitEstimator.py:102-107,136-141."""
    for i in (1, 2, 3):
        x = _bitparm(x, sd[f"{prefix}f{i}.h"], sd[f"{prefix}f{i}.b"], sd[f"{prefix}f{i}.a"])
    return _bitparm(x, sd[f"{prefix}f4.h"], sd[f"{prefix}f4.b"], None)


def factorised_cdf(x: torch.Tensor, sd: SD, prefix: str = "bitEstimator_z.") -> torch.Tensor:
    """Single-head factorised CDF for the hyper-latent z, params [1,C,1,1]. bitEstimator.py:30-45."""
    return multihead_cdf(x, sd, prefix)


def entropy_table_float(sd: SD, M: int, min_feat: int = -127, max_feat: int = 128,
                        q_intv: int = 1, prefix: str = "bitEstimators.") -> torch.Tensor:
    """CDF of every prior/channel at the bin edges min_feat-.5 ... max_feat+.5 -> [K,M,L] fp32.
    manynets_compressor.py:142-152."""
    L = int((-min_feat + max_feat) / q_intv + 2)
    targets = torch.linspace(min_feat - 0.5, max_feat + 0.5, L)
    targets = targets.expand(1, M, L).unsqueeze(-1).contiguous()
    return multihead_cdf(targets, sd, prefix).squeeze(-1).squeeze(1)


def z_table_float(sd: SD, N: int, min_feat: int = -127, max_feat: int = 128,
                  q_intv: int = 1) -> torch.Tensor:
    """Factorised z CDF at the bin edges -> [N, L]. abstract_model.py:277-284."""
    L = int((-min_feat + max_feat) / q_intv + 2)
    targets = torch.linspace(min_feat - 0.5, max_feat + 0.5, L)
    targets = targets.expand(N, L).unsqueeze(0).unsqueeze(-1)
    return factorised_cdf(targets, sd).squeeze(-1).squeeze(0)


ORDER_ROUND_THEN_ADD = 0   # torchac's _renorm_cast_cdf_ (used for the Manypriors table)
ORDER_ADD_THEN_ROUND = 1   # the in-repo copy, abstract_model.py:59-73 (hyperprior tables)


def renorm_cast_cdf(cdf: torch.Tensor, precision: int = 16, order: int = ORDER_ROUND_THEN_ADD):
    """Float CDF [..., L] -> int16 CDF with strictly increasing entries.

    order 0 (published torchac algorithm; the module itself is absent, see header):
        round(cdf * (2^p - (L-1))) -> int16 (wraps) -> + arange(L)
    order 1 (abstract_model.py:59-73):
        round(cdf * (2^p - (L-1)) + arange(L)) -> int16 (wraps)
    """
    L = cdf.shape[-1]
    factor = torch.tensor(2, dtype=torch.float32).pow_(precision) - (L - 1)
    r = torch.arange(L, dtype=torch.int16)
    if order == ORDER_ROUND_THEN_ADD:
        return cdf.mul(factor).round().to(torch.int16) + r
    scaled = cdf.mul(factor)
    scaled.add_(r)
    return scaled.round().to(torch.int16)


# ----------------------------------------------------------------------------
# Quantisation and prior competition  (manynets_compressor.py:204-269, 533-554)
# ----------------------------------------------------------------------------
def quantise(y: torch.Tensor) -> torch.Tensor:
    return torch.round(y)


def symbols_from_latent(yhat: torch.Tensor, min_feat: int = -127, q_intv: int = 1) -> torch.Tensor:
    """int16 symbols, location-major / channel-minor: index = loc*M + c. manynets_compressor.py:264-269.
    Accepts [B,M,h,w]; returns [B, h*w*M]."""
    B, M = yhat.shape[:2]
    s = ((yhat.reshape(B, M, -1).transpose(1, 2) - min_feat) / q_intv).round().short()
    return s.reshape(B, -1)


def latent_from_symbols(sym: torch.Tensor, M: int, h: int, w: int, min_feat: int = -127,
                        q_intv: int = 1) -> torch.Tensor:
    """Inverse of symbols_from_latent. manynets_compressor.py:329-330."""
    B = sym.shape[0]
    f = sym.view(B, -1, M).transpose(1, 2).reshape(B, M, h, w)
    return (f * q_intv + min_feat).float()


def prior_bits_network(yhat: torch.Tensor, sd: SD) -> torch.Tensor:
    """bits[k,b,h,w] = sum_c clamp(-log2(P_k,c(y+.5)-P_k,c(y-.5)+1e-10),0,50).
    manynets_compressor.py:548-553."""
    probs = multihead_cdf(yhat + 0.5, sd) - multihead_cdf(yhat - 0.5, sd)
    return torch.sum(torch.clamp(-torch.log2(probs + 1e-10), 0, 50), dim=2)


def bits_lut(table_f: torch.Tensor, form: str = "log2") -> torch.Tensor:
    """Per-(k,c,s) code length from the float table: [K,M,L-1].
    form "log2": forward()'s expression (manynets_compressor.py:551-553);
    form "ln":   encode()'s expression  (manynets_compressor.py:247-248)."""
    p = table_f[..., 1:] - table_f[..., :-1]
    if form == "log2":
        return torch.clamp(-torch.log2(p + 1e-10), 0, 50)
    return torch.clamp(-1.0 * torch.log(p + 1e-10) / math.log(2.0), 0, 50)


def prior_bits_lut(yhat: torch.Tensor, table_f: torch.Tensor, min_feat: int = -127,
                   form: str = "log2") -> torch.Tensor:
    """Same quantity as prior_bits_network, through the float table (the LUT competition of
    manynets_compressor.py:238-249).  Returns [K,B,h,w]."""
    K, M, L = table_f.shape
    B, _, h, w = yhat.shape
    lut = bits_lut(table_f, form)                                   # [K,M,L-1]
    s = (yhat - min_feat).long().permute(1, 0, 2, 3).reshape(1, M, -1).expand(K, M, -1)
    bits = torch.gather(lut, 2, s).view(K, M, B, h, w).permute(0, 2, 1, 3, 4)
    return bits.sum(dim=2)


def competition(total_bits: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(minbits, idx) = first-min over the prior axis. manynets_compressor.py:554."""
    return total_bits.min(0)


def gather_cdf(table_i16: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """cdf[loc, c, :] = table[idx[loc], c, :]  -> [n_loc, M, L] int16. manynets_compressor.py:252-262."""
    return table_i16[idx.reshape(-1).long()]


# ----------------------------------------------------------------------------
# Hyperprior bits  (abstract_model.py:226-243)
# ----------------------------------------------------------------------------
def laplace_cdf(x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """CDF of Laplace(0, scale): 0.5 - 0.5*sign(x)*expm1(-|x|/scale) (torch.distributions.Laplace.cdf)."""
    return 0.5 - 0.5 * x.sign() * torch.expm1(-x.abs() / scale)


def laplace_bits(yhat: torch.Tensor, sigma: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Total bits of yhat under Laplace(0, clamp(sigma)). abstract_model.py:226-234."""
    s = sigma.clamp(1e-10, 1e10)
    probs = laplace_cdf(yhat + 0.5, s) - laplace_cdf(yhat - 0.5, s)
    bits = torch.clamp(-1.0 * torch.log(probs + 1e-10) / math.log(2.0), 0, 50)
    return bits.sum(), probs


def factorised_bits(zhat: torch.Tensor, sd: SD) -> Tuple[torch.Tensor, torch.Tensor]:
    """Total bits of zhat under the factorised prior. abstract_model.py:236-243."""
    prob = factorised_cdf(zhat + 0.5, sd) - factorised_cdf(zhat - 0.5, sd)
    bits = torch.clamp(-1.0 * torch.log(prob + 1e-10) / math.log(2.0), 0, 50)
    return bits.sum(), prob


def laplace_cdf_rows(sigma_flat: torch.Tensor, min_feat: int = -127, max_feat: int = 128,
                     precision: int = 16) -> torch.Tensor:
    """Per-element int16 CDF rows for the hyperprior main stream: [n, L].
    abstract_model.py:404-416 (in-repo rounding order)."""
    L = max_feat - min_feat + 2
    targets = torch.linspace(min_feat - 0.5, max_feat + 0.5, L)
    s = sigma_flat.clamp(1e-10, 1e10)
    cdf = laplace_cdf(targets.unsqueeze(-1).expand(-1, s.numel()), s).transpose(0, 1)
    return renorm_cast_cdf(cdf.contiguous(), precision, ORDER_ADD_THEN_ROUND)


# ----------------------------------------------------------------------------
# Side string  (pt_helpers.py:32-107, utilities.py:357-390)
# ----------------------------------------------------------------------------
def side_string_bits(idx: torch.Tensor, png_size_bytes: Optional[int] = None) -> int:
    """Bits needed for the prior-index map: min(raw, LZMA, [PNG], fixed-length); 0 if all zero.
    pt_helpers.py:86-107.  The PNG term needs pypng (absent here) and is taken from the caller."""
    a = idx.cpu().numpy()
    if a.min() == 0 and a.max() == 0:
        return 0
    if a.max() <= 255 and a.min() >= 0:
        a = a.astype(np.uint8)
    elif a.max() <= 32767 and a.min() >= 0:
        a = a.astype(np.int16)
    else:
        raise NotImplementedError("index map outside [0, 32767]")
    raw = a.tobytes()
    bits = len(raw) * 8
    bits = min(bits, len(lzma.compress(raw)) * 8)
    if png_size_bytes is not None:
        bits = min(bits, png_size_bytes * 8)
    mx = int(a.max())
    per_value = 0 if mx <= 0 else math.floor(math.log2(mx) + 1)
    return min(bits, a.size * per_value)


# ----------------------------------------------------------------------------
# Whole-model entry points
# ----------------------------------------------------------------------------
def manypriors_forward(x: torch.Tensor, sd: SD, min_feat: int = -127, use_lut: bool = False,
                       table_f: Optional[torch.Tensor] = None, synthesis_fn=synthesis) -> dict:
    """Eval-mode forward of the Manypriors model without entropy coding.
    manynets_compressor.py:502-644 (eval branch, dist_patch_size == 1)."""
    y = analysis(x, sd)
    yhat = quantise(y)
    recon = synthesis_fn(yhat, sd)
    if use_lut:
        total = prior_bits_lut(yhat, table_f, min_feat)
    else:
        total = prior_bits_network(yhat, sd)
    minbits, idx = competition(total)
    npix = x.shape[0] * x.shape[2] * x.shape[3]
    bits_feature = torch.gather(total, 0, idx.unsqueeze(0)).sum()
    bpp_side = torch.tensor(side_string_bits(idx), dtype=torch.float32) / npix
    bpp_feature = bits_feature / npix
    return {
        "y": y, "yhat": yhat, "recon": recon,
        "reconstructed_image": recon.clamp(0.0, 1.0),
        "visual_loss": F.mse_loss(recon, x),
        "total_bits": total, "minbits": minbits, "idx": idx,
        "bpp_feature": bpp_feature, "bpp_sidestring": bpp_side, "bpp": bpp_feature + bpp_side,
        "used_dists": idx.unique().tolist(),
    }


def manypriors_forward_train(x: torch.Tensor, sd: SD, noise: torch.Tensor, passthrough_ae: bool,
                             dists_last_use: np.ndarray, synthesis_fn=synthesis) -> dict:
    """Training-mode forward of the Manypriors model (values only), dist_patch_size == 1.
    manynets_compressor.py:514-530 (noise quantisation; passthrough_ae: the decoder sees round(y)), :547-561
    (network-form bits on the noisy latent), :579-602 (prior bookkeeping, forced training of priors idle for more
    than 50 calls; draws from numpy's global generator like the reference), :604-607, :621-625.
    `noise` is the U(-0.5, 0.5) tensor the reference draws; `dists_last_use` is updated in place."""
    K = sd["bitEstimators.f1.h"].shape[0]
    y = analysis(x, sd)
    y_entropy = y + noise
    y_ae = torch.round(y) if passthrough_ae else y_entropy
    recon = synthesis_fn(y_ae, sd)
    total = prior_bits_network(y_entropy, sd)
    minbits, idx = total.min(0)
    max_forced = int(y_entropy.size(0) * y_entropy.size(2) * y_entropy.size(3) * 0.1)
    used = idx.unique()
    dists_last_use[used.numpy()] = 0
    unused = np.setdiff1d(np.arange(K), used.numpy(), assume_unique=True)
    dists_last_use[unused] += 1
    num_forced = 0
    if K > 1:
        idle = np.argwhere(dists_last_use > 50).flatten()
        num_forced = min(idle.size, max_forced)
        victims = minbits.flatten().sort(descending=True).indices[:num_forced]
        idx.flatten()[victims] = torch.tensor(np.random.choice(unused, victims.size(), replace=False))
    npix = x.shape[0] * x.shape[2] * x.shape[3]
    bits_feature = torch.gather(total, 0, idx.unsqueeze(0)).sum()
    # the reference charges feature_batched_shape[0] = the BATCH size (not the location count) times log2 K here
    bpp_side = torch.tensor((y_entropy.size(0) * math.log2(K)) / npix)
    bpp_feature = bits_feature / npix
    return {
        "y": y, "recon": recon, "reconstructed_image": recon.clamp(0.0, 1.0), "visual_loss": F.mse_loss(recon, x),
        "total_bits": total, "minbits": minbits, "idx": idx,
        "bpp_feature": bpp_feature, "bpp_sidestring": bpp_side, "bpp": bpp_feature + bpp_side,
        "used_dists": used.tolist(), "num_forced_dists": num_forced,
    }


def prior_bits_lut_encode_layout(yhat: torch.Tensor, table_f: torch.Tensor, min_feat: int = -127):
    """LUT competition exactly as encode() lays it out: bits as a [K, h*w, M, 1] tensor summed over
    dim 2 (manynets_compressor.py:238-249).  ATen reduces this contiguous inner dimension in a
    different order from the [K,B,M,h,w] sum of forward(), so exact fp32 ties can resolve
    differently between the two forms."""
    K, M, L = table_f.shape
    _, _, h, w = yhat.shape
    lut = bits_lut(table_f, "ln")
    s = (yhat - min_feat).long().view(M, -1).transpose(0, 1)            # [hw, M]
    bits = torch.gather(lut.unsqueeze(1).expand(K, h * w, M, L - 1), 3,
                        s.view(1, h * w, M, 1).expand(K, h * w, M, 1))
    return bits.sum(dim=2)                                              # [K, hw, 1]


def manypriors_encode_symbols(x: torch.Tensor, sd: SD, table_f: torch.Tensor, min_feat: int = -127):
    """encode(entropy_coding=False): symbols + prior indices through the LUT competition.
    manynets_compressor.py:197-276 (batch size 1)."""
    yhat = quantise(analysis(x, sd))
    _, M, h, w = yhat.shape
    _, idx = prior_bits_lut_encode_layout(yhat, table_f, min_feat).min(0)
    return symbols_from_latent(yhat, min_feat).flatten(), idx.reshape(-1, 1), h, w


def hyperprior_forward(x: torch.Tensor, sd: SD) -> dict:
    """Eval-mode forward of the hyperprior model. abstract_model.py:168-272."""
    y = analysis(x, sd)
    z = hyper_analysis(y, sd)
    zhat = torch.round(z)
    sigma = hyper_synthesis(zhat, sd)
    yhat = torch.round(y)
    recon = synthesis(yhat, sd)
    bits_y, _ = laplace_bits(yhat, sigma)
    bits_z, _ = factorised_bits(zhat, sd)
    npix = x.shape[0] * x.shape[2] * x.shape[3]
    return {
        "y": y, "z": z, "zhat": zhat, "sigma": sigma, "yhat": yhat, "recon": recon,
        "reconstructed_image": recon.clamp(0.0, 1.0),
        "visual_loss": F.mse_loss(recon, x),
        "bpp_feature": bits_y / npix, "bpp_sidestring": bits_z / npix,
        "bpp": bits_y / npix + bits_z / npix,
    }

# ----------------------------------------------------------------------------
# Image-quality statistics  (src/common/extlibs/pt_ms_ssim.py:8-216, src/compression/tests.py:393-418)
# ----------------------------------------------------------------------------
def gauss_window(size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    """pt_ms_ssim.py:8-23."""
    coords = torch.arange(size).to(dtype=torch.float)
    coords -= size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _gaussian_filter(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    """Separable valid filtering, width first then height (pt_ms_ssim.py:26-38)."""
    C = x.shape[1]
    w = win.view(1, 1, 1, -1).repeat(C, 1, 1, 1)
    out = F.conv2d(x, w, stride=1, padding=0, groups=C)
    return F.conv2d(out, w.transpose(2, 3), stride=1, padding=0, groups=C)


def ssim_maps(X: torch.Tensor, Y: torch.Tensor, win: torch.Tensor, data_range: float = 1.0,
              K: Tuple[float, float] = (0.01, 0.03)) -> Tuple[torch.Tensor, torch.Tensor]:
    """(mean SSIM, mean contrast-structure) per (image, channel). pt_ms_ssim.py:41-84."""
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _gaussian_filter(X, win), _gaussian_filter(Y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = _gaussian_filter(X * X, win) - mu1_sq
    sigma2_sq = _gaussian_filter(Y * Y, win) - mu2_sq
    sigma12 = _gaussian_filter(X * Y, win) - mu1_mu2
    cs_map = (2 * sigma12 + C2) / (sigma1_sq + sigma2_sq + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return torch.flatten(ssim_map, 2).mean(-1), torch.flatten(cs_map, 2).mean(-1)


def ms_ssim(X: torch.Tensor, Y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """pt_ms_ssim.py:144-216 with the default window and weights, size_average=True."""
    weights = torch.FloatTensor([0.0448, 0.2856, 0.3001, 0.2363, 0.1333])
    win = gauss_window()
    mcs = []
    for i in range(5):
        ssim_pc, cs = ssim_maps(X, Y, win, data_range)
        if i < 4:
            mcs.append(torch.relu(cs))
            padding = (X.shape[2] % 2, X.shape[3] % 2)
            X = F.avg_pool2d(X, kernel_size=2, padding=padding)
            Y = F.avg_pool2d(Y, kernel_size=2, padding=padding)
    ssim_pc = torch.relu(ssim_pc)
    val = torch.prod(torch.stack(mcs + [ssim_pc], dim=0) ** weights.view(-1, 1, 1), dim=0)
    return val.mean()


def psnr(xhat: torch.Tensor, x: torch.Tensor) -> float:
    """tests.py:393-398."""
    mse = F.mse_loss(xhat, x)
    return (10 * (torch.log(1.0 / mse) / math.log(10))).item()
