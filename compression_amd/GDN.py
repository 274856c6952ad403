"""GDN / IGDN layer (mirror of src/compression/models/GDN.py:28-95 of the reference).

Inside the transforms the normalisation never runs as a layer of its own: it is fused into the epilogue of the
producing convolution (compression_amd/csrc/conv_igemm.hip and the patch kernels), which also applies the lower-bound +
square reparametrisation of (beta, gamma); there this module only owns the two parameters, with the reference's names,
shapes and initial values, so that reference checkpoints load unchanged.  Called directly, forward() runs the same
arithmetic as one HIP kernel on the NCHW input (mpc_gdn_forward, exact fp32 matrix pipe for the channel mix).
"""
import torch
from torch import nn

from . import ops


class GDN(nn.Module):
    def __init__(self, ch, inverse=False, beta_min=1e-6, gamma_init=0.1, reparam_offset=2 ** -18):
        super().__init__()
        if beta_min != 1e-6 or reparam_offset != 2 ** -18:
            raise NotImplementedError("the kernels use beta_min=1e-6 and reparam_offset=2**-18")
        self.inverse = inverse
        self.pedestal = reparam_offset ** 2
        self.beta = nn.Parameter(torch.sqrt(torch.ones(ch) + self.pedestal))
        self.gamma = nn.Parameter(torch.sqrt(gamma_init * torch.eye(ch) + self.pedestal))

    def forward(self, inputs):
        """[B, C, H, W] (or [B, C, D, W, H], folded like the reference's :67-70) fp32 on the GPU -> same shape.  Values
        only: the library has no backward pass."""
        return ops.gdn_forward(inputs.contiguous(), self.beta.detach(), self.gamma.detach(), self.inverse)
