"""Transform networks (mirror of src/compression/models/Balle2018PT_compressor.py).

Each network owns nn.Conv2d / nn.ConvTranspose2d / GDN modules purely as PARAMETER HOLDERS with the
reference's attribute names, shapes and initialisation, so `state_dict()` is interchangeable with the
reference's.  `forward` never calls them: it runs the fused libmpc layers (implicit-GEMM convolution +
bias + GDN/IGDN/ReLU/exp epilogue) on NHWC activations.
"""
import math

import torch
from torch import nn

from . import _lib, ops
from .GDN import GDN


def _xavier(mod, gain):
    torch.nn.init.xavier_normal_(mod.weight.data, gain)
    torch.nn.init.constant_(mod.bias.data, 0.01)
    return mod


class _FusedNet(nn.Module):
    """Caches the packed libmpc layers and rebuilds them when a parameter changes."""

    def _spec(self):
        """-> list of (kind, conv_module, stride, act, gdn_module|None, in_abs)."""
        raise NotImplementedError

    precision = "fp32"     # "fp16": fp16-operand MFMA layers with fp16 activations (synthesis transforms only)

    def _fingerprint(self):
        return (self.precision,) + tuple((p.data_ptr(), p._version, p.device) for p in self.parameters())

    def layers(self):
        fp = self._fingerprint()
        if getattr(self, "_fused_fp", None) != fp:
            built = []
            bound = 1.0          # |input| bound along the stack: image in [0,1], then 1/sqrt(min gamma_ii) after a GDN
            for kind, conv, stride, act, gdn, in_abs in self._spec():
                built.append(ops.ConvLayer(kind, conv.weight.data, conv.bias.data, conv.kernel_size[0], stride, act,
                                           None if gdn is None else gdn.beta.data,
                                           None if gdn is None else gdn.gamma.data, in_abs,
                                           f16=self.precision == "fp16", x3=self.precision == "f16x3",
                                           in_bound=bound if self.precision == "f16x3" else 0.0))
                if gdn is not None and not gdn.inverse:
                    gii = torch.diagonal(gdn.gamma.data).float().clamp_min(2.0 ** -18) ** 2 - 2.0 ** -36
                    bound = float(1.0 / torch.sqrt(gii.min().clamp_min(1e-12)))
                else:
                    bound = 0.0
            object.__setattr__(self, "_fused", built)
            object.__setattr__(self, "_fused_fp", fp)
        return self._fused

    def _apply(self, fn, *args, **kwargs):
        object.__setattr__(self, "_fused_fp", None)
        return super()._apply(fn, *args, **kwargs)


def _check_activation(name):
    if name != "GDN":
        raise NotImplementedError(f"activation_function={name}: the MI355X path implements the GDN transforms")


class Analysis_net(_FusedNet):
    """g_a: 4 x conv 5x5 s2, GDN after the first three (Balle2018PT_compressor.py:58-102)."""

    def __init__(self, out_channel_N=192, out_channel_M=320, activation_function="GDN", precision="fp32"):
        """precision "fp32": exact f32 MFMA; "f16x3": conv2-4 on split-precision fp16 MFMA (fp32-grade results,
        activations carried as fp16 hi/lo pairs scaled by 2^8 -- needs GDN outputs bounded by 255, which
        1/sqrt(min gamma_ii) guarantees for every sane model; checked when the layers are built)."""
        super().__init__()
        _check_activation(activation_function)
        N, M = out_channel_N, out_channel_M
        self.precision = precision
        self.gdn1, self.gdn2, self.gdn3 = GDN(N), GDN(N), GDN(N)
        self.conv1 = _xavier(nn.Conv2d(3, N, 5, stride=2, padding=2), math.sqrt(2 * (3 + N) / 6))
        self.conv2 = _xavier(nn.Conv2d(N, N, 5, stride=2, padding=2), math.sqrt(2))
        self.conv3 = _xavier(nn.Conv2d(N, N, 5, stride=2, padding=2), math.sqrt(2))
        self.conv4 = _xavier(nn.Conv2d(N, M, 5, stride=2, padding=2), math.sqrt(2 * (M + N) / (N + N)))

    def _spec(self):
        C, G = _lib.MPC_CONV, _lib.MPC_ACT_GDN
        return [(C, self.conv1, 2, G, self.gdn1, False), (C, self.conv2, 2, G, self.gdn2, False),
                (C, self.conv3, 2, G, self.gdn3, False), (C, self.conv4, 2, _lib.MPC_ACT_NONE, None, False)]

    def forward_nhwc(self, x4, round_out=False, keep_unrounded=True):
        """x4: NHWC image padded to 4 channels -> latent NHWC (or (y | None, round(y)))."""
        l1, l2, l3, l4 = self.layers()
        if self.precision == "f16x3":
            self._check_split_range()
            t = l3(l2(l1(x4, out_split=True), out_split=True), out_split=True)
        else:
            t = l3(l2(l1(x4)))
        return l4(t, round_out=round_out, keep_unrounded=keep_unrounded)

    def forward_symbols(self, x4, min_feat, n_sym_values, n_clamped=None, keep_rounded=False, events=None, first_layer_on=None):
        """x4 -> (round(y) | None, int16 symbols NHWC, clamped count): quantisation and symbolisation run in the last
        layer's epilogue.  events (a dict): receives "layer2_done" and "layer3_done", events recorded behind the second and the third layer.
        first_layer_on = (stream, event | None): the first layer runs on that stream (x4 must be complete there), behind
        the event, and the other layers on the current stream behind it: for a caller that lets the first layer, which
        leaves half of the matrix pipe idle, start before the work queued on the current stream has ended."""
        l1, l2, l3, l4 = self.layers()
        split = self.precision == "f16x3"
        if split:
            self._check_split_range()
        kw = dict(out_split=True) if split else {}
        if first_layer_on is None:
            t = l1(x4, **kw)
        else:
            early, after = first_layer_on
            main = torch.cuda.current_stream(x4.device)
            with torch.cuda.stream(early):
                if after is not None:
                    early.wait_event(after)
                t = l1(x4, **kw)
                first_done = torch.cuda.Event()
                first_done.record()
            main.wait_event(first_done)
            for part in (t if isinstance(t, (tuple, list)) else (t,)):
                if isinstance(part, torch.Tensor):
                    part.record_stream(main)
        t = l2(t, **kw)
        if events is not None:
            events["layer2_done"] = torch.cuda.Event()
            events["layer2_done"].record()
        t = l3(t, **kw)
        if events is not None:
            events["layer3_done"] = torch.cuda.Event()
            events["layer3_done"].record()
        return l4.forward_symbols(t, min_feat, n_sym_values, n_clamped, keep_rounded)

    def _check_split_range(self):
        """|GDN output_i| <= 1/sqrt(gamma_eff_ii); the split format holds |v| < 255.9."""
        fp = self._fused_fp
        if getattr(self, "_split_checked", None) == fp:
            return
        for gdn in (self.gdn1, self.gdn2, self.gdn3):
            g = torch.diagonal(gdn.gamma.data).float().clamp_min(2.0 ** -18) ** 2 - 2.0 ** -36
            if float(g.min()) <= 1.6e-5:
                raise _lib.MpcError("precision='f16x3' needs GDN gamma_ii > 1.6e-5 (activation bound 255); use 'fp32'")
        object.__setattr__(self, "_split_checked", fp)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x.contiguous(), 4)))


def _scored_last_layer(layer, t, target):
    """-> (clamp(x_hat, 0, 1), per-image squared error against target): fused into the layer's epilogue if it can."""
    fused = layer.forward_scored(t, target)
    return fused if fused is not None else ops.clamp_sse_per_image(layer(t, nchw_out=True), target)


class Synthesis_net(_FusedNet):
    """g_s: 4 x deconv 5x5 s2, IGDN after the first three (Balle2018PT_compressor.py:262-308)."""

    def __init__(self, out_channel_N=192, out_channel_M=320, out_channel_fin=3, activation_function="GDN",
                 precision="fp16"):
        super().__init__()
        _check_activation(activation_function)
        N, M = out_channel_N, out_channel_M
        self.precision = precision if (N % 64 == 0 and M % 64 == 0 and N <= 192) else "fp32"
        self.igdn1, self.igdn2, self.igdn3 = GDN(N, inverse=True), GDN(N, inverse=True), GDN(N, inverse=True)
        dc = lambda i, o: nn.ConvTranspose2d(i, o, 5, stride=2, padding=2, output_padding=1)
        self.deconv1 = _xavier(dc(M, N), math.sqrt(2 * (M + N) / (M + M)))
        self.deconv2 = _xavier(dc(N, N), math.sqrt(2))
        self.deconv3 = _xavier(dc(N, N), math.sqrt(2))
        self.deconv4 = _xavier(dc(N, out_channel_fin), math.sqrt(2 * (N + 3) / (N + N)))

    def _spec(self):
        D, G = _lib.MPC_DECONV, _lib.MPC_ACT_IGDN
        return [(D, self.deconv1, 2, G, self.igdn1, False), (D, self.deconv2, 2, G, self.igdn2, False),
                (D, self.deconv3, 2, G, self.igdn3, False), (D, self.deconv4, 2, _lib.MPC_ACT_NONE, None, False)]

    def forward_nhwc(self, y, score_against=None, events=None):
        """y: NHWC fp32 latent -> NCHW fp32 image (unclamped).  With precision "fp16" (default) the four layers
        run on fp16-operand MFMA with fp16 activations: x_hat stays within 1e-2 of the fp32 result.
        score_against (an NCHW fp32 image): -> (clamp(x_hat, 0, 1), float64 [B] squared errors against it), from the
        last layer's epilogue where that kernel exists, else from one more pass.
        events (a dict): receives "deconv2_done", recorded in front of the third transposed layer."""
        l1, l2, l3, l4 = self.layers()
        if self.precision == "fp16" and y.dtype != torch.float16:
            y = ops.f32_to_f16(y)
        t = l2(l1(y))
        if events is not None:
            events["deconv2_done"] = torch.cuda.Event()
            events["deconv2_done"].record()
        t = l3(t)
        if score_against is None:
            return l4(t, nchw_out=True)
        return _scored_last_layer(l4, t, score_against)

    def latent_from_symbols(self, sym, min_feat, q_intv):
        """int16 symbols NHWC -> the latent in the type this network's first layer takes."""
        return ops.symbols_to_latent_f16(sym, min_feat, q_intv) if self.precision == "fp16" else ops.symbols_to_latent(sym, min_feat, q_intv)

    def forward(self, x):
        return self.forward_nhwc(ops.nchw_to_nhwc(x.contiguous()))


class Analysis_prior_net(_FusedNet):
    """h_a: |y| -> conv3 s1 + ReLU -> conv5 s2 + ReLU -> conv5 s2 (Balle2018PT_compressor.py:105-127)."""

    def __init__(self, out_channel_N=192, out_channel_M=320):
        super().__init__()
        N, M = out_channel_N, out_channel_M
        self.conv1 = _xavier(nn.Conv2d(M, N, 3, stride=1, padding=1), math.sqrt(2 * (M + N) / (M + M)))
        self.conv2 = _xavier(nn.Conv2d(N, N, 5, stride=2, padding=2), math.sqrt(2))
        self.conv3 = _xavier(nn.Conv2d(N, N, 5, stride=2, padding=2), math.sqrt(2))

    def _spec(self):
        C, R = _lib.MPC_CONV, _lib.MPC_ACT_RELU
        return [(C, self.conv1, 1, R, None, True), (C, self.conv2, 2, R, None, False),
                (C, self.conv3, 2, _lib.MPC_ACT_NONE, None, False)]

    def forward_nhwc(self, y, round_out=False):
        l1, l2, l3 = self.layers()
        return l3(l2(l1(y)), round_out=round_out)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x.contiguous())))


class SynthesisTFCodeExp_prior_net(_FusedNet):
    """h_s: deconv5 s2 + ReLU -> deconv5 s2 + ReLU -> deconv3 s1 -> exp (Balle2018PT_compressor.py:223-251)."""

    def __init__(self, out_channel_N=192, out_channel_M=320):
        super().__init__()
        N, M = out_channel_N, out_channel_M
        self.deconv1 = _xavier(nn.ConvTranspose2d(N, N, 5, stride=2, padding=2, output_padding=1), math.sqrt(2))
        self.deconv2 = _xavier(nn.ConvTranspose2d(N, N, 5, stride=2, padding=2, output_padding=1), math.sqrt(2))
        self.deconv3 = _xavier(nn.ConvTranspose2d(N, M, 3, stride=1, padding=1), math.sqrt(2 * (M + N) / (N + N)))

    def _spec(self):
        D, R = _lib.MPC_DECONV, _lib.MPC_ACT_RELU
        return [(D, self.deconv1, 2, R, None, False), (D, self.deconv2, 2, R, None, False),
                (D, self.deconv3, 1, _lib.MPC_ACT_EXP, None, False)]

    def forward_nhwc(self, z):
        l1, l2, l3 = self.layers()
        return l3(l2(l1(z)))

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x.contiguous())))


from . import abstract_model  # noqa: E402  (abstract_model does not import this module)


class Balle2018PTTFExp_ImageCompressor(abstract_model.AbstractHyperpriorImageCompressor):
    """The hyperprior baseline the reference actually uses (Balle2018PT_compressor.py:10-19)."""

    def __init__(self, out_channel_N=192, out_channel_M=320, lossf="mse", device="cuda:0", entropy_coding=False, **kwargs):
        super().__init__(out_channel_N=out_channel_N, out_channel_M=out_channel_M, lossf=lossf, device=device,
                         entropy_coding=entropy_coding, **kwargs)
        # analysis_precision: "fp32" (exact f32 MFMA, default) | "f16x3" (split-precision fp16 MFMA, fp32-grade: the same
        # symbols up to rounding ties; h_a still reads the unrounded fp32 latent)
        self.Encoder = Analysis_net(out_channel_N=out_channel_N, out_channel_M=out_channel_M,
                                    precision=kwargs.get("analysis_precision") or "fp32")
        self.Decoder = Synthesis_net(out_channel_N=out_channel_N, out_channel_M=out_channel_M,
                                     precision=kwargs.get("synthesis_precision", "fp16"))
        self.priorDecoder = SynthesisTFCodeExp_prior_net(out_channel_N=out_channel_N, out_channel_M=out_channel_M)
        self.priorEncoder = Analysis_prior_net(out_channel_N=out_channel_N, out_channel_M=out_channel_M)
