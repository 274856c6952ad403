"""Base classes of the codec models (mirror of src/compression/models/abstract_model.py).

Inference only: `forward`, `encode`, `decode`, `compress`, `update_entropy_table` run on the MI355X
through libmpc.  Training-mode forward (noise quantisation, straight-through rounding) is outside this
package and raises.
"""
import math

import torch
from torch import nn

from . import _lib, ops
from .bitEstimator import BitEstimator


def _renorm_cast_cdf_(cdf, precision):
    """Float CDF [..., L] -> int16 CDF, "add arange, then round" order (abstract_model.py:59-73)."""
    return ops.renorm_cast_cdf(cdf.contiguous(), precision, _lib.MPC_ORDER_ADD_THEN_ROUND)


def _scalar(v, device):
    return torch.tensor(float(v), dtype=torch.float32, device=device)


class AbstractImageCompressor(nn.Module):
    def __init__(self, device):
        super().__init__()
        self.device = torch.device(device) if not isinstance(device, torch.device) else device

    def forward(self, input_image):
        """(b,c,h,w) image batch -> dict with "reconstructed_image", "visual_loss", "bpp", and optionally
        "bpp_feature", "bpp_sidestring" (abstract_model.py:39-48)."""
        raise NotImplementedError

    def cpu(self):
        raise _lib.MpcError("compression_amd models run on the MI355X only; there is no CPU path")

    def todev(self, device):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        return self.to(self.device)


class AbstractHyperpriorImageCompressor(AbstractImageCompressor):
    """Hyperprior codec (abstract_model.py:76-272): g_a/g_s, h_a/h_s, Laplace(0, sigma) conditional for y,
    learned factorised prior for z."""

    def __init__(self, out_channel_N=192, out_channel_M=320, lossf="mse", device="cuda:0", min_feat=-127,
                 max_feat=128, min_feat_gaussian=-127, max_feat_gaussian=128, q_intv=1, precision=16,
                 entropy_coding=False, conditional_distribution="Laplace", passthrough_ae=False, **kwargs):
        super().__init__(device=device)
        self.entropy_coding = bool(entropy_coding)
        self.out_channel_N, self.out_channel_M = out_channel_N, out_channel_M
        self.bitEstimator_z = BitEstimator(out_channel_N)
        self.lossf = lossf
        if lossf not in ("mse", None):
            raise NotImplementedError(f"lossf={lossf}: the MI355X path computes the MSE visual loss only")
        self.min_feat, self.max_feat, self.q_intv, self.precision = min_feat, max_feat, q_intv, precision
        self.min_feat_gaussian, self.max_feat_gaussian = min_feat_gaussian, max_feat_gaussian
        if q_intv != 1:
            raise NotImplementedError("q_intv != 1")
        if conditional_distribution != "Laplace":
            raise NotImplementedError(f"conditional_distribution={conditional_distribution}")
        ntargets = int((-min_feat + max_feat) / q_intv + 1)
        self.z_entropy_table = torch.empty(out_channel_M, ntargets, dtype=torch.short)
        self.num_distributions = 1
        self.passthrough_ae = passthrough_ae
        self._synthesis_precision = kwargs.get("synthesis_precision", None)
        self.table_authority = kwargs.get("table_authority", "device")     # see update_entropy_table

    # ---- helpers ---------------------------------------------------------------------------
    def _require_eval(self):
        if self.training:
            raise NotImplementedError("training-mode forward is not part of the MI355X inference path; call .eval()")

    def _visual_loss(self, recon, x):
        xhat, sse = ops.clamp_sse(recon, x if self.lossf == "mse" else None)
        loss = None if sse is None else (sse / recon.numel()).to(torch.float32).reshape(())
        return xhat, loss

    def _n_targets(self):
        return int((-self.min_feat + self.max_feat) / self.q_intv + 2)

    # ---- reference API ----------------------------------------------------------------------
    def forward(self, input_image, **kwargs):
        """Eval-mode rate/distortion forward (abstract_model.py:168-272)."""
        self._require_eval()
        x = input_image.contiguous()
        B, _, H, W = x.shape
        npix = B * H * W
        y, yhat = self.Encoder.forward_nhwc(ops.nchw_to_nhwc(x, 4), round_out=True)
        _, zhat = self.priorEncoder.forward_nhwc(y, round_out=True)
        sigma = self.priorDecoder.forward_nhwc(zhat)
        recon = self.Decoder.forward_nhwc(yhat)
        xhat, visual_loss = self._visual_loss(recon, x)
        bits_y = ops.laplace_bits_total(yhat, sigma)
        hi = self.bitEstimator_z.forward_nhwc(zhat, 0.5)
        lo = self.bitEstimator_z.forward_nhwc(zhat, -0.5)
        bits_z = ops.prob_bits_total(hi, lo)
        bpp_feature = (bits_y / npix).to(torch.float32).reshape(())
        bpp_z = (bits_z / npix).to(torch.float32).reshape(())
        return {
            "reconstructed_image": xhat,
            "visual_loss": visual_loss,
            "bpp_feature": bpp_feature,
            "bpp_sidestring": bpp_z,
            "bpp": bpp_feature + bpp_z,
        }

    def update_entropy_table(self, authority=None):
        """int16 CDF table of the factorised z prior, in-repo rounding order (abstract_model.py:274-291).
        authority "device" (default) | "host": who evaluates the float table, as in the Manypriors model's method."""
        authority = authority or self.table_authority
        if authority not in ("device", "host"):
            raise ValueError(f"table authority must be 'device' or 'host', got {authority!r}")
        L = self._n_targets()
        if authority == "host":
            dev = next(self.bitEstimator_z.parameters()).device
            tf = self.bitEstimator_z.table_host(self.min_feat - 0.5, self.max_feat + 0.5, L).to(dev)
        else:
            tf = self.bitEstimator_z.table(self.min_feat - 0.5, L)             # [N, L]
        self.main_targets_first, self.main_targets_n = self.min_feat_gaussian - 0.5, \
            int((-self.min_feat_gaussian + self.max_feat_gaussian) / self.q_intv + 2)
        self.z_entropy_table_float = self.entropy_table_float = tf
        self.z_entropy_table = _renorm_cast_cdf_(tf, self.precision)
        if tf.is_cuda:
            torch.cuda.current_stream(tf.device).synchronize()                  # complete before any other stream reads it
        object.__setattr__(self, "_z_table_made_from", self._z_table_inputs(authority))
        return True

    def _z_table_inputs(self, authority=None):
        return (authority or self.table_authority, self._n_targets(), self.min_feat, self.max_feat, self.precision,
                tuple((p.data_ptr(), p._version) for p in self.bitEstimator_z.parameters()))

    def _ensure_z_table(self):
        """The codec's own call before every batch: builds the z table again only when the prior's parameters (or what
        the table is made with) have changed since it was built -- batches that are in flight on different streams
        share the table.  (In-place changes are seen through the parameters' version counters; after writing through
        `.data`, call update_entropy_table() or eval().)"""
        if getattr(self, "_z_table_made_from", None) != self._z_table_inputs():
            self.update_entropy_table()

    def eval(self):
        if self.entropy_coding and next(self.parameters()).is_cuda:
            self.update_entropy_table()
        return self.train(False)

    def get_parameters(self, lr=None):
        assert lr is not None
        return [
            {"params": self.Encoder.parameters(), "name": "encoder"},
            {"params": self.Decoder.parameters(), "name": "decoder"},
            {"params": self.priorEncoder.parameters(), "name": "prior_encoder"},
            {"params": self.priorDecoder.parameters(), "name": "prior_decoder"},
            {"params": self.bitEstimator_z.parameters(), "lr": lr * 10, "name": "bit_estimator"},
        ]

    def _staging(self, nbytes, purpose="bytes"):
        """Page-locked host buffer of at least nbytes (uint8), reused between calls.  One buffer per purpose and per HIP
        stream: calls that are in flight on different streams (CodecPipeline) never share one."""
        bufs = self.__dict__.setdefault("_stage_bufs", {})
        key = (purpose, torch.cuda.current_stream().cuda_stream)
        buf = bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = bufs[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
        return buf

    def _meta_host(self, n, purpose="meta"):
        """Page-locked int32 host buffer for the one small readback of a call (per purpose and stream, as above)."""
        bufs = self.__dict__.setdefault("_meta_bufs", {})
        key = (purpose, torch.cuda.current_stream().cuda_stream)
        buf = bufs.get(key)
        if buf is None or buf.numel() < n:
            buf = bufs[key] = torch.empty(max(n, 256), dtype=torch.int32, pin_memory=True)
        return buf[:n]

    # ---- entropy-coded encode / decode (abstract_model.py:370-500) ----------------------------------
    def _latent_hw(self, H, W):
        """Spatial size of y and z for an H x W image (four, then two more, stride-2 k5 p2 convolutions)."""
        def down(n, times):
            for _ in range(times):
                n = (n + 1) // 2
            return n
        return (down(H, 4), down(W, 4)), (down(H, 6), down(W, 6))

    def _main_L(self):
        return int((-self.min_feat_gaussian + self.max_feat_gaussian) / self.q_intv + 2)

    def _main_rows(self, zhat):
        """Per-element int16 Laplace(0, sigma) CDF rows of the main stream, in NHWC element order, as a
        [n_loc, M, L] table (abstract_model.py:402-416): the tensor the reference hands to torchac, 514 bytes per
        symbol.  For inspection and tests only -- encode() and decode() never materialise it."""
        sigma = self.priorDecoder.forward_nhwc(zhat)
        L = self._main_L()
        rows = ops.laplace_cdf_rows(sigma.reshape(-1), self.min_feat_gaussian - 0.5, L, self.precision)
        return sigma, rows.view(-1, self.out_channel_M, L)

    def _z_table(self):
        return self.z_entropy_table.view(1, self.out_channel_N, -1).contiguous()

    def encode_batch(self, input_image, entropy_coding=True):
        """B equally sized images -> a list of B bitstreams [main string, z string, H, W] (abstract_model.py:370-454,
        batched); with entropy_coding=False the two int16 symbol tensors, one row per image, in the reference's
        orders (y: NCHW flattened, z: location-major).  The strings are MRS1 rANS containers, not torchac bytes.
        H and W must be multiples of 64.

        The main stream is coded under per-element Laplace(0, sigma) rows that are evaluated where they are needed
        (mpc_rans_encode_laplace: two row entries per symbol) -- the reference materialises all of them
        (abstract_model.py:409-416, 514 bytes per symbol); here the only per-symbol scratch is the coder's 8-byte
        (start, frequency, reciprocal) entry."""
        job = self.encode_batch_begin(input_image, entropy_coding)
        if not entropy_coding:
            self._check_clamped(job)
            B, h, w, M = job["B"], job["h"], job["w"], self.out_channel_M
            return job["sym_y"].view(B, h, w, M).permute(0, 3, 1, 2).reshape(B, -1), job["sym_z"].reshape(B, -1)
        return self.encode_batch_finish(job)[0]

    def encode_batch_begin(self, input_image, entropy_coding=True, input_nhwc=None):
        """Enqueues g_a, h_a, the symbols and -- for an entropy-coded call -- h_s and the bit estimate on the current
        stream, starts the copy of the three numbers the host needs (two clamped counts, the estimate) and returns at
        once.  -> a job for encode_batch_finish; job["transforms_done"] is an event recorded behind g_a.  input_nhwc:
        the image already in the first layer's layout (ops.nchw_to_nhwc(x, 4))."""
        self._require_eval()
        x = input_image.contiguous()
        B, _, H, W = x.shape
        if entropy_coding and (H % 64 or W % 64):
            raise ValueError("entropy-coded encode needs H and W to be multiples of 64")
        x4 = input_nhwc if input_nhwc is not None else ops.nchw_to_nhwc(x, 4)
        y, yhat = self.Encoder.forward_nhwc(x4, round_out=True)
        transforms_done = torch.cuda.Event()
        transforms_done.record()
        _, zhat = self.priorEncoder.forward_nhwc(y, round_out=True)
        sym_y, clamp_y = ops.latent_to_symbols(yhat, self.min_feat_gaussian, self.q_intv, self._main_L() - 1)
        sym_z, clamp_z = ops.latent_to_symbols(zhat, self.min_feat, self.q_intv, self._n_targets() - 1)
        _, h, w, _ = yhat.shape
        sigma = None
        bits = torch.zeros(1, dtype=torch.float64, device=x.device)
        if entropy_coding:
            self._ensure_z_table()                      # small (N x L); keeps the z table in step with the parameters
            sigma = self.priorDecoder.forward_nhwc(zhat)
            bits = ops.laplace_bits_total(yhat, sigma)
        # one readback for the two clamped counts and the bit estimate the stream count is chosen from
        numbers = torch.cat([clamp_y.view(-1)[:1].double(), clamp_z.view(-1)[:1].double(), bits.view(-1)])
        host = self._meta_host(6, "encode").view(torch.float64)
        host.copy_(numbers, non_blocking=True)
        ready = torch.cuda.Event()
        ready.record()
        return dict(B=B, H=H, W=W, h=h, w=w, sym_y=sym_y, sym_z=sym_z, sigma=sigma, host=host, ready=ready,
                    transforms_done=transforms_done, dev=x.device)

    def _check_clamped(self, job):
        job["ready"].synchronize()
        n_clamp_y, n_clamp_z, est_bits = job["host"].tolist()
        if n_clamp_y or n_clamp_z:
            raise ValueError("encode: a latent falls outside the CDF tables' range "
                             f"[{self.min_feat_gaussian},{self.max_feat_gaussian}] / [{self.min_feat},{self.max_feat}]")
        return est_bits

    def encode_batch_finish(self, job):
        """Waits for the job's three numbers, enqueues the two coders with the stream counts chosen from the estimate
        and brings the byte strings to the host.  -> (list of [main string, z string, H, W],): the arguments of
        decode_batch_begin.  Call it with the stream the job began on current."""
        B, H, W, h, w, dev = job["B"], job["H"], job["W"], job["h"], job["w"], job["dev"]
        N, M = self.out_channel_N, self.out_channel_M
        est = self._check_clamped(job) / 8 / B
        sym_z = job["sym_z"]
        main = ops.rans_encode_laplace(job["sigma"], job["sym_y"], B, M, self.min_feat_gaussian - 0.5, self._main_L(),
                                       self.precision, *ops.choose_streams(h * w, M, est))
        n_zloc = sym_z.numel() // (B * N)
        zidx = torch.zeros(B * n_zloc, dtype=torch.int32, device=dev)
        # the z string is a hundredth of the image: its streams are budgeted against the image's bytes, not its own
        # (a handful of streams would make its two coders the longest serial chains of the whole codec)
        side = ops.rans_encode(self._z_table(), zidx, sym_z.reshape(-1), B,
                               *ops.choose_streams(n_zloc, N, est + float(n_zloc * N), ops.MRS1_STREAM_BUDGET / 2))
        mains, zstrs = ops.containers_to_bytes([main, side])   # both coders enqueued, then sizes and payloads read back together
        return ([[mains[i], zstrs[i], H, W] for i in range(B)],)

    def encode(self, input_image, entropy_coding=True):
        """One image -> [main string, z string, H, W] (abstract_model.py:370-454); with entropy_coding=False the
        two int16 symbol vectors in the reference's orders.  See encode_batch."""
        if input_image.shape[0] != 1:
            raise ValueError("encode() handles one image per call like the reference; encode_batch() takes more")
        out = self.encode_batch(input_image, entropy_coding)
        return out[0] if entropy_coding else (out[0].reshape(-1), out[1].reshape(-1))

    def decode_batch(self, bitstreams):
        """A list of B bitstreams [main string, z string, H, W] of equally sized images -> reconstruction [B, 3, H, W]
        (NCHW, unclamped).  The z strings are decoded first; sigma = h_s(z_hat) then gives the decoder of the main
        strings its rows (mpc_rans_decode_laplace), exactly as the encoder saw them."""
        job = self.decode_batch_begin(bitstreams)
        self.decode_batch_synthesis(job)
        return self.decode_batch_finish(job)

    def decode_batch_begin(self, bitstreams):
        """Copies the strings to the device and enqueues both decoders (with h_s between them) on the current stream.
        -> a job for decode_batch_synthesis (enqueues g_s) and decode_batch_finish (checks the decoders' status)."""
        self._require_eval()
        B = len(bitstreams)
        if B == 0:
            raise ValueError("decode_batch: no bitstream given")
        H, W = bitstreams[0][2], bitstreams[0][3]
        if any((b[2], b[3]) != (H, W) for b in bitstreams):
            raise ValueError("decode_batch: all images of one call must have one size")
        if H % 64 or W % 64:
            raise ValueError("decode needs H and W to be multiples of 64")
        dev = next(self.parameters()).device
        self._ensure_z_table()
        (h, w), (zh, zw) = self._latent_hw(H, W)
        N, M = self.out_channel_N, self.out_channel_M
        zidx = torch.zeros(B * zh * zw, dtype=torch.int32, device=dev)
        mains, zstrs = [b[0] for b in bitstreams], [b[1] for b in bitstreams]
        z_geometry = ops.parse_mrs1_headers(zstrs, zh * zw, N)
        y_geometry = ops.parse_mrs1_headers(mains, h * w, M)
        (zdata, znb), (ydata, ynb) = ops._upload_containers(zstrs, dev), ops._upload_containers(mains, dev)
        sym_z, zstatus = ops.rans_decode(self._z_table(), zidx, zdata, znb, B, *z_geometry)
        zhat = ops.symbols_to_latent(sym_z.view(B, zh, zw, N), self.min_feat, self.q_intv)
        sigma = self.priorDecoder.forward_nhwc(zhat)
        if tuple(sigma.shape) != (B, h, w, M):
            raise ValueError("decode: the z string does not match the image size")
        # (a damaged z string gives some sigma or other; both decoders stay inside their buffers whatever they read,
        # so the two statuses are read back together, once, before the result is used)
        sym_y, ystatus = ops.rans_decode_laplace(sigma, ydata, ynb, B, M, self.min_feat_gaussian - 0.5, self._main_L(),
                                                 self.precision, *y_geometry)
        st_host = self._meta_host(2 * B, "decode")
        st_host.copy_(torch.cat([zstatus, ystatus]), non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return dict(B=B, h=h, w=w, sym=sym_y, st_host=st_host, done=done, recon=None, keep=(zdata, ydata, sigma))

    def decode_batch_synthesis(self, job, score_against=None, events=None):
        """Enqueues g_s for a job of decode_batch_begin (on the stream that job began on).  score_against (the
        batch's original, NCHW fp32): the job's reconstruction is then clamp(x_hat, 0, 1) and job["sse"] holds the
        per-image squared errors against it.  events: passed to the synthesis network (see Synthesis_net.forward_nhwc)."""
        B, h, w, M = job["B"], job["h"], job["w"], self.out_channel_M
        yhat = ops.symbols_to_latent(job["sym"].view(B, h, w, M), self.min_feat_gaussian, self.q_intv)
        kw = dict(events=events) if events is not None else {}
        if score_against is None:
            job["recon"] = self.Decoder.forward_nhwc(yhat, **kw)
        else:
            job["recon"], job["sse"] = self.Decoder.forward_nhwc(yhat, score_against=score_against, **kw)
        return job

    def decode_batch_finish(self, job):
        """Waits for the job's decoders (not for g_s), checks their status, returns the reconstruction."""
        job["done"].synchronize()
        B = job["B"]
        status = job["st_host"].tolist()
        if min(status) < 0:
            raise _lib.MpcError(f"corrupt MRS1 stream: decoder status z {status[:B]}, main {status[B:]}")
        return job["recon"]

    def decode(self, bitstream):
        """[main string, z string, H, W] -> reconstruction (NCHW, unclamped); a tensor is taken as the latent and
        only goes through g_s.  (The reference's decode, abstract_model.py:456-500, stops in a breakpoint and
        mis-sizes z; this one inverts encode().)"""
        if isinstance(bitstream, torch.Tensor):
            return self.Decoder(bitstream)
        return self.decode_batch([bitstream])

    def compress(self, input_image):
        """-> (round(y), round(z)) as NCHW tensors (abstract_model.py:733-738)."""
        y, yhat = self.Encoder.forward_nhwc(ops.nchw_to_nhwc(input_image.contiguous(), 4), round_out=True)
        _, zhat = self.priorEncoder.forward_nhwc(y, round_out=True)
        return ops.nhwc_to_nchw(yhat), ops.nhwc_to_nchw(zhat)
