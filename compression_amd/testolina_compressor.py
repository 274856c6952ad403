"""Testolina et al. (2021) extended decoder and the two models built on it (mirror of
src/compression/models/testolina_compressor.py)."""
import math

import torch
from torch import nn

from . import _lib, abstract_model, manynets_compressor
from .Balle2018PT_compressor import (Analysis_net, Analysis_prior_net, SynthesisTFCodeExp_prior_net, _FusedNet, _xavier)
from .GDN import GDN


class Testolina_Synthesis_net(_FusedNet):
    """Four extra 5x5 stride-1 conv + ReLU layers at latent resolution, then the usual four deconvs with IGDN
    (testolina_compressor.py:18-100).  deconv1 therefore takes N, not M, channels."""

    def __init__(self, out_channel_N=192, out_channel_M=320, out_channel_fin=3, precision="fp16", **kwargs):
        super().__init__()
        N, M = out_channel_N, out_channel_M
        self.precision = precision if (N % 64 == 0 and M % 64 == 0 and N <= 192) else "fp32"
        self.igdn1, self.igdn2, self.igdn3 = GDN(N, inverse=True), GDN(N, inverse=True), GDN(N, inverse=True)
        self.relu = nn.ReLU()
        self.convred1 = _xavier(nn.Conv2d(M, N, 5, stride=1, padding=2), math.sqrt(2 * (M + N) / (M + M)))
        self.convred2 = _xavier(nn.Conv2d(N, N, 5, stride=1, padding=2), math.sqrt(2))
        self.convred3 = _xavier(nn.Conv2d(N, N, 5, stride=1, padding=2), math.sqrt(2))
        self.convred4 = _xavier(nn.Conv2d(N, N, 5, stride=1, padding=2), math.sqrt(2))
        dc = lambda i, o: nn.ConvTranspose2d(i, o, 5, stride=2, padding=2, output_padding=1)
        self.deconv1 = _xavier(dc(N, N), math.sqrt(2))
        self.deconv2 = _xavier(dc(N, N), math.sqrt(2))
        self.deconv3 = _xavier(dc(N, N), math.sqrt(2))
        self.deconv4 = _xavier(dc(N, out_channel_fin), math.sqrt(2 * (N + 3) / (N + N)))

    def _spec(self):
        C, D, R, G = _lib.MPC_CONV, _lib.MPC_DECONV, _lib.MPC_ACT_RELU, _lib.MPC_ACT_IGDN
        return [(C, self.convred1, 1, R, None, False), (C, self.convred2, 1, R, None, False),
                (C, self.convred3, 1, R, None, False), (C, self.convred4, 1, R, None, False),
                (D, self.deconv1, 2, G, self.igdn1, False), (D, self.deconv2, 2, G, self.igdn2, False),
                (D, self.deconv3, 2, G, self.igdn3, False), (D, self.deconv4, 2, _lib.MPC_ACT_NONE, None, False)]

    def forward_nhwc(self, y, score_against=None, events=None):
        """events (a dict): receives "deconv2_done", recorded in front of the third transposed layer."""
        from . import ops
        from .Balle2018PT_compressor import _scored_last_layer
        ls = self.layers()
        t = ops.f32_to_f16(y) if (self.precision == "fp16" and y.dtype != torch.float16) else y
        for layer in ls[:-2]:
            t = layer(t)
        if events is not None:
            events["deconv2_done"] = torch.cuda.Event()
            events["deconv2_done"].record()
        t = ls[-2](t)
        if score_against is None:
            return ls[-1](t, nchw_out=True)
        return _scored_last_layer(ls[-1], t, score_against)

    def forward(self, x):
        from . import ops
        return self.forward_nhwc(ops.nchw_to_nhwc(x.contiguous()))

    def latent_from_symbols(self, sym, min_feat, q_intv):
        from . import ops
        return ops.symbols_to_latent_f16(sym, min_feat, q_intv) if self.precision == "fp16" else ops.symbols_to_latent(sym, min_feat, q_intv)


class TestolinaHP_ImageCompressor(abstract_model.AbstractHyperpriorImageCompressor):
    """Hyperprior model with the extended decoder (testolina_compressor.py:103-134)."""

    def __init__(self, out_channel_N=192, out_channel_M=320, lossf="mse", device="cuda:0", entropy_coding=False, **kwargs):
        super().__init__(out_channel_N=out_channel_N, out_channel_M=out_channel_M, lossf=lossf, device=device,
                         entropy_coding=entropy_coding, **kwargs)
        self.Encoder = Analysis_net(out_channel_N=out_channel_N, out_channel_M=out_channel_M,
                                    precision=kwargs.get("analysis_precision") or "fp32")
        self.Decoder = Testolina_Synthesis_net(out_channel_N=out_channel_N, out_channel_M=out_channel_M,
                                               precision=kwargs.get("synthesis_precision") or "fp16")
        self.priorDecoder = SynthesisTFCodeExp_prior_net(out_channel_N=out_channel_N, out_channel_M=out_channel_M)
        self.priorEncoder = Analysis_prior_net(out_channel_N=out_channel_N, out_channel_M=out_channel_M)


class TestolinaManyPriors_ImageCompressor(manynets_compressor.Balle2017ManyPriors_ImageCompressor):
    """Manypriors model with the extended decoder (testolina_compressor.py:137-185)."""

    def __init__(self, out_channel_N=192, out_channel_M=320, **kwargs):
        sp = kwargs.get("synthesis_precision", "fp16")
        super().__init__(out_channel_N=out_channel_N, out_channel_M=out_channel_M, **kwargs)
        self.Decoder = Testolina_Synthesis_net(out_channel_N=out_channel_N, out_channel_M=out_channel_M, precision=sp)
