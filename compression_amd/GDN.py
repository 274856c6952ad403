"""GDN / IGDN parameters (mirror of src/compression/models/GDN.py:28-95 of the reference).

On the MI355X path the normalisation is never run as a layer of its own: it is fused into the
epilogue of the producing convolution (compression_amd/csrc/conv_igemm.hip), which also applies the
lower-bound + square reparametrisation of (beta, gamma).  This module only owns the two parameters,
with the reference's names, shapes and initial values, so that reference checkpoints load unchanged.
"""
import torch
from torch import nn


class GDN(nn.Module):
    def __init__(self, ch, inverse=False, beta_min=1e-6, gamma_init=0.1, reparam_offset=2 ** -18):
        super().__init__()
        if beta_min != 1e-6 or reparam_offset != 2 ** -18:
            raise NotImplementedError("the fused kernel uses beta_min=1e-6 and reparam_offset=2**-18")
        self.inverse = inverse
        self.pedestal = reparam_offset ** 2
        self.beta = nn.Parameter(torch.sqrt(torch.ones(ch) + self.pedestal))
        self.gamma = nn.Parameter(torch.sqrt(gamma_init * torch.eye(ch) + self.pedestal))

    def forward(self, inputs):
        raise NotImplementedError(
            "GDN runs fused into the preceding convolution; call the owning Analysis_net / Synthesis_net")
