"""Learned factorised CDFs (mirror of src/compression/models/bitEstimator.py:9-45, 87-141).

The modules own the Bitparm parameters (same names/shapes as the reference) and evaluate the 4-layer
chain through libmpc (`mpc_bitparm_cdf_eval` / `mpc_bitparm_cdf_table`).
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops


class _Bitparm(nn.Module):
    def __init__(self, shape, final, init_std=0.01, init_mode="normal"):
        super().__init__()
        self.final = final

        def make():
            t = torch.empty(shape)
            if init_mode == "normal":
                torch.nn.init.normal_(t, 0, init_std)
            elif init_mode == "xavier_uniform":
                torch.nn.init.xavier_uniform_(t.view(shape[0], -1), init_std)
            else:
                raise NotImplementedError(init_mode)
            return nn.Parameter(t)

        self.h = make()
        self.b = make()
        self.a = None if final else make()


class _Chain(nn.Module):
    def chain_params(self):
        """[h1,b1,a1,h2,b2,a2,h3,b3,a3,h4,b4] in the order libmpc expects."""
        return [self.f1.h, self.f1.b, self.f1.a, self.f2.h, self.f2.b, self.f2.a,
                self.f3.h, self.f3.b, self.f3.a, self.f4.h, self.f4.b]

    def n_rows(self):
        return self.f1.h.numel()

    def table(self, first_target: float, L: int) -> torch.Tensor:
        """CDF at first_target + s, s < L, for every (head, channel) row -> [n_rows, L]."""
        return ops.bitparm_cdf_table(self.chain_params(), self.n_rows(), first_target, L)

    def _host_chain(self, x):
        """The four Bitparm layers on host copies of the parameters, with the reference's own ATen calls in the
        reference's order (bitEstimator.py:21-26 / 136-141): the result is the reference's, bit for bit, on the same
        host.  Load-time use only (a table has K*M*L entries)."""
        for f in (self.f1, self.f2, self.f3):
            h, b, a = f.h.detach().cpu(), f.b.detach().cpu(), f.a.detach().cpu()
            x = x * F.softplus(h) + b
            x = x + torch.tanh(x) * torch.tanh(a)
        h, b = self.f4.h.detach().cpu(), self.f4.b.detach().cpu()
        return torch.sigmoid(x * F.softplus(h) + b)


class BitEstimator(_Chain):
    """Single-head estimator used for the hyper-latent z (bitEstimator.py:30-45); params [1,C,1,1]."""

    def __init__(self, channel, **kwargs):
        super().__init__()
        shape = (1, channel, 1, 1)
        self.f1, self.f2, self.f3 = _Bitparm(shape, False), _Bitparm(shape, False), _Bitparm(shape, False)
        self.f4 = _Bitparm(shape, True)

    def table_host(self, lo: float, hi: float, L: int) -> torch.Tensor:
        """[C, L] float table evaluated on the host exactly as the reference does it (abstract_model.py:277-283:
        linspace targets, expanded -- not contiguous -- to [1, C, L, 1])."""
        C = self.f1.h.numel()
        targets = torch.linspace(lo, hi, L)
        targets = targets.expand(C, L).unsqueeze(0).unsqueeze(-1)
        return self._host_chain(targets).squeeze()

    def forward_nhwc(self, x, offset: float = 0.0):
        """x: NHWC [B,h,w,C] -> CDF(x + offset), same shape."""
        return ops.bitparm_cdf_eval(self.chain_params(), x, offset, row_minor=True)

    def forward(self, x, offset: float = 0.0):
        """x: [B,C,h,w] -> CDF(x + offset), same shape."""
        B, C, h, w = x.shape
        rows = x.permute(1, 0, 2, 3).reshape(C, -1).contiguous()
        out = ops.bitparm_cdf_eval(self.chain_params(), rows, offset)
        return out.view(C, B, h, w).permute(1, 0, 2, 3).contiguous()


class MultiHeadBitEstimator(_Chain):
    """K-head estimator of the Manypriors model (bitEstimator.py:87-141); params [K,1,C,1,1]."""

    def __init__(self, channel: int, nb_head: int, shape=("g", "bs", "ch", "h", "w"), bitparm_init_mode="normal",
                 bitparm_init_range=0.01, **kwargs):
        super().__init__()
        if tuple(shape) != ("g", "bs", "ch", "h", "w"):
            raise NotImplementedError(shape)
        self.channel, self.nb_head = channel, nb_head
        pshape = (nb_head, 1, channel, 1, 1)
        mk = lambda final: _Bitparm(pshape, final, bitparm_init_range, bitparm_init_mode)
        self.f1, self.f2, self.f3, self.f4 = mk(False), mk(False), mk(False), mk(True)

    def table_host(self, lo: float, hi: float, L: int) -> torch.Tensor:
        """[K, C, L] float table evaluated on the host exactly as the reference does it
        (manynets_compressor.py:142-152: linspace targets as a contiguous [1, C, L, 1] against [K, 1, C, 1, 1]
        parameters)."""
        targets = torch.linspace(lo, hi, L)
        targets = targets.expand(1, self.channel, L).unsqueeze(-1).contiguous()
        table = self._host_chain(targets).squeeze(-1).squeeze(1)
        return table.squeeze()

    def forward(self, x, offset: float = 0.0):
        """x: [B,C,h,w] -> [K,B,C,h,w] CDF values of every head (the network form of the competition)."""
        B, C, h, w = x.shape
        rows = x.permute(1, 0, 2, 3).reshape(1, C, -1).expand(self.nb_head, C, -1).reshape(self.nb_head * C, -1).contiguous()
        out = ops.bitparm_cdf_eval(self.chain_params(), rows, offset)
        return out.view(self.nb_head, C, B, h, w).permute(0, 2, 1, 3, 4).contiguous()
