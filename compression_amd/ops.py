"""Operator layer: one Python function per libmpc entry point (include/mpc.h), with torch tensors used
only as device-memory containers (`.data_ptr()`).  Every function enqueues HIP work on torch's current
stream of the tensor's device.  There is no fallback: a CPU tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes
import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, MpcError, context_for, ptr

c_void_p = ctypes.c_void_p


class Profiler:
    """Optional per-call HIP-event timing of libmpc launches (events are recorded on torch's current
    stream, which is the stream the kernels are enqueued on).  Used by bench.py for the roofline numbers."""

    def __init__(self, only=None):
        self.records = []          # (name, start_event, end_event, meta)
        self.only = only           # None: every launch; else the set of names to time (the others get no events)

    def summary(self):
        """-> {name: {"calls", "ms", "flops", "bytes"}} (synchronises)."""
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, meta in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += meta.get("flops", 0.0)
            d["bytes"] += meta.get("bytes", 0.0)
        return out


PROFILER: Optional[Profiler] = None


class _timed:
    def __init__(self, name, **meta):
        self.name, self.meta = name, meta

    def __enter__(self):
        self.on = PROFILER is not None and (PROFILER.only is None or self.name in PROFILER.only)
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on and PROFILER is not None and exc[0] is None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILER.records.append((self.name, self.e0, e1, self.meta))
        return False


def _dev(t: torch.Tensor, dtype=None, name="tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise MpcError(f"{name} must live on the GPU (got {t.device}); compression_amd has no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise MpcError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise MpcError(f"{name} must be contiguous")
    return t


# ---- layout --------------------------------------------------------------------------------------
def nchw_to_nhwc(x: torch.Tensor, cpad: Optional[int] = None) -> torch.Tensor:
    _dev(x, torch.float32, "x")
    B, C, H, W = x.shape
    cpad = C if cpad is None else cpad
    out = torch.empty((B, H, W, cpad), dtype=torch.float32, device=x.device)
    context_for(x.device).call("mpc_nchw_to_nhwc", ptr(x), B, C, H, W, cpad, ptr(out))
    return out


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    _dev(x, torch.float32, "x")
    B, H, W, C = x.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    context_for(x.device).call("mpc_nhwc_to_nchw", ptr(x), B, C, H, W, ptr(out))
    return out


def latent_to_symbols(latent: torch.Tensor, min_feat: int, q_intv: int, n_sym_values: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (int16 symbols of latent.shape, int32[1] count of clamped values)."""
    _dev(latent, torch.float32, "latent")
    sym = torch.empty(latent.shape, dtype=torch.int16, device=latent.device)
    nclamp = torch.empty(1, dtype=torch.int32, device=latent.device)
    context_for(latent.device).call("mpc_latent_to_symbols", ptr(latent), latent.numel(), int(min_feat), int(q_intv),
                                    int(n_sym_values), ptr(sym), ptr(nclamp))
    return sym, nclamp


def symbols_to_latent(sym: torch.Tensor, min_feat: int, q_intv: int) -> torch.Tensor:
    _dev(sym, torch.int16, "sym")
    out = torch.empty(sym.shape, dtype=torch.float32, device=sym.device)
    context_for(sym.device).call("mpc_symbols_to_latent", ptr(sym), sym.numel(), int(min_feat), int(q_intv), ptr(out))
    return out


def f32_to_f16(x: torch.Tensor) -> torch.Tensor:
    _dev(x, torch.float32, "x")
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    context_for(x.device).call("mpc_f32_to_f16", ptr(x), x.numel(), ptr(out))
    return out


def f16_to_f32(x: torch.Tensor) -> torch.Tensor:
    _dev(x, torch.float16, "x")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    context_for(x.device).call("mpc_f16_to_f32", ptr(x), x.numel(), ptr(out))
    return out


def sum_f32(x: torch.Tensor) -> torch.Tensor:
    """Double-precision sum -> float64[1] device tensor."""
    _dev(x, torch.float32, "x")
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    context_for(x.device).call("mpc_sum_f32", ptr(x), x.numel(), ptr(out))
    return out


def gdn_forward(x: torch.Tensor, beta: torch.Tensor, gamma: torch.Tensor, inverse: bool) -> torch.Tensor:
    """GDN / IGDN layer on an NCHW (or 5-D [B, C, D, W, H]) fp32 tensor with the raw parameters (GDN.py:65-95)."""
    x = _dev(x, torch.float32, "x")
    if x.dim() not in (4, 5):
        raise ValueError(f"gdn_forward: expected a 4-D or 5-D tensor, got {tuple(x.shape)}")
    B, C = x.shape[:2]
    if tuple(beta.shape) != (C,) or tuple(gamma.shape) != (C, C):
        raise ValueError(f"gdn_forward: beta {tuple(beta.shape)} / gamma {tuple(gamma.shape)} do not fit {C} channels")
    out = torch.empty_like(x)
    context_for(x.device).call("mpc_gdn_forward", ptr(x), ptr(_dev(beta, torch.float32, "beta")),
                               ptr(_dev(gamma, torch.float32, "gamma")), B, C, x[0, 0].numel(), int(bool(inverse)), ptr(out))
    return out


def clamp_sse(recon: torch.Tensor, target: Optional[torch.Tensor]):
    """-> (clamp(recon,0,1), float64[1] sum of squared errors vs target | None)."""
    _dev(recon, torch.float32, "recon")
    xhat = torch.empty_like(recon)
    sse = None
    if target is not None:
        _dev(target, torch.float32, "target")
        if target.shape != recon.shape:
            raise MpcError("clamp_sse: shape mismatch")
        sse = torch.empty(1, dtype=torch.float64, device=recon.device)
    context_for(recon.device).call("mpc_clamp_sse", ptr(recon), ptr(target), recon.numel(), ptr(xhat), ptr(sse))
    return xhat, sse


def clamp_sse_per_image(recon: torch.Tensor, target: torch.Tensor, score_clamped: bool = True):
    """-> (clamp(recon,0,1), float64[B] sum of squared errors of every image of the batch, taken on the clamped
    values (what the test harness' PSNR sees) or, with score_clamped=False, on the raw reconstruction)."""
    _dev(recon, torch.float32, "recon"); _dev(target, torch.float32, "target")
    if target.shape != recon.shape:
        raise MpcError("clamp_sse_per_image: shape mismatch")
    B = recon.shape[0]
    xhat = torch.empty_like(recon)
    sse = torch.empty(B, dtype=torch.float64, device=recon.device)
    with _timed("clamp_sse", bytes=3.0 * recon.numel() * 4):
        context_for(recon.device).call("mpc_clamp_sse_batched", ptr(recon), ptr(target), B, recon.numel() // B,
                                       int(score_clamped), ptr(xhat), ptr(sse))
    return xhat, sse


def stats_accumulate(vec: torch.Tensor, sse: torch.Tensor, values_per_image: int, bits: float, pixels: float, nbytes: float,
                     seconds: float = 0.0, side_bits: float = 0.0, msssim: Optional[torch.Tensor] = None,
                     reset: bool = False) -> None:
    """Adds one batch to the 16-double statistics vector `vec` on the device (layout: include/mpc.h,
    mpc_stats_accumulate), or with `reset` makes it this batch's sums.  sse / msssim: float64 [B] device tensors."""
    _dev(vec, torch.float64, "vec"); _dev(sse, torch.float64, "sse")
    if vec.numel() < 16:
        raise MpcError("stats_accumulate: vec must hold 16 doubles")
    if msssim is not None:
        _dev(msssim, torch.float64, "msssim")
        if msssim.numel() != sse.numel():
            raise MpcError("stats_accumulate: msssim and sse must have one entry per image")
    hs = (ctypes.c_double * 5)(float(bits), float(pixels), float(nbytes), float(seconds), float(side_bits))
    context_for(vec.device).call("mpc_stats_accumulate", ptr(sse), ptr(msssim), sse.numel(), int(values_per_image), hs,
                                 int(reset), ptr(vec))


# ---- learned CDF tables ------------------------------------------------------------------------
def _param_array(params: Sequence[torch.Tensor], n_rows: int):
    if len(params) != 11:
        raise MpcError("expected 11 Bitparm tensors (h1,b1,a1,h2,b2,a2,h3,b3,a3,h4,b4)")
    keep = []
    arr = (c_void_p * 11)()
    for i, t in enumerate(params):
        t = _dev(t.detach().reshape(-1).contiguous(), torch.float32, f"params[{i}]")
        if t.numel() != n_rows:
            raise MpcError(f"params[{i}] has {t.numel()} elements, expected {n_rows}")
        keep.append(t)
        arr[i] = t.data_ptr()
    return arr, keep


def bitparm_cdf_table(params: Sequence[torch.Tensor], n_rows: int, first_target: float, L: int) -> torch.Tensor:
    arr, keep = _param_array(params, n_rows)
    dev = keep[0].device
    out = torch.empty((n_rows, L), dtype=torch.float32, device=dev)
    context_for(dev).call("mpc_bitparm_cdf_table", arr, n_rows, ctypes.c_float(first_target), L, ptr(out))
    return out


def bitparm_cdf_eval(params: Sequence[torch.Tensor], x: torch.Tensor, offset: float, row_minor: bool = False) -> torch.Tensor:
    """CDF_row(x + offset), same shape as x.  x is [n_rows, n_per_row], or with row_minor any tensor whose
    LAST axis is the row (channel) axis, e.g. an NHWC activation."""
    _dev(x, torch.float32, "x")
    if row_minor:
        n_rows, n_per_row = x.shape[-1], x.numel() // x.shape[-1]
    else:
        n_rows, n_per_row = x.shape
    arr, keep = _param_array(params, n_rows)
    out = torch.empty_like(x)
    context_for(x.device).call("mpc_bitparm_cdf_eval", arr, n_rows, ptr(x), n_per_row, int(row_minor),
                               ctypes.c_float(offset), ptr(out))
    return out


def renorm_cast_cdf(cdf: torch.Tensor, precision: int = 16, order: int = _lib.MPC_ORDER_ROUND_THEN_ADD) -> torch.Tensor:
    _dev(cdf, torch.float32, "cdf")
    L = cdf.shape[-1]
    out = torch.empty(cdf.shape, dtype=torch.int16, device=cdf.device)
    context_for(cdf.device).call("mpc_renorm_cast_cdf", ptr(cdf), cdf.numel() // L, L, precision, order, ptr(out))
    return out


def bits_lut(table_f: torch.Tensor, form: int = _lib.MPC_BITS_LOG2) -> torch.Tensor:
    """table_f [K,M,L] -> code-length LUT [M, L-1, K]."""
    _dev(table_f, torch.float32, "table_f")
    K, M, L = table_f.shape
    out = torch.empty((M, L - 1, K), dtype=torch.float32, device=table_f.device)
    context_for(table_f.device).call("mpc_bits_lut", ptr(table_f), K, M, L, form, ptr(out))
    return out


# ---- competition / CDF rows ----------------------------------------------------------------------
def prior_argmin(sym: torch.Tensor, lut: torch.Tensor, want_total: bool = False):
    """sym int16 [n_loc, M]; lut [M,S,K] -> (idx int32 [n_loc], minbits fp32 [n_loc], total [K,n_loc] | None)."""
    _dev(sym, torch.int16, "sym")
    _dev(lut, torch.float32, "lut")
    n_loc, M = sym.shape
    M2, S, K = lut.shape
    if M2 != M:
        raise MpcError(f"LUT has {M2} channels, symbols have {M}")
    idx = torch.empty(n_loc, dtype=torch.int32, device=sym.device)
    minbits = torch.empty(n_loc, dtype=torch.float32, device=sym.device)
    total = torch.empty((K, n_loc), dtype=torch.float32, device=sym.device) if want_total else None
    with _timed("prior_argmin", bytes=float(n_loc) * M * 4 * K, symbols=n_loc * M):
        context_for(sym.device).call("mpc_prior_argmin", ptr(sym), n_loc, M, K, S, ptr(lut), ptr(idx), ptr(minbits), ptr(total))
    return idx, minbits, total


def prior_bits_network(params: Sequence[torch.Tensor], y: torch.Tensor, K: int, noise: Optional[torch.Tensor] = None,
                       want_entropy_latent: bool = False):
    """The competition in its network form (latents that need not be integers): y fp32 [n_loc, M] (NHWC), params the 11
    Bitparm tensors of the K heads ([K,1,M,1,1] each), noise (optional) of y's shape is added first.
    -> (idx int32 [n_loc], minbits fp32 [n_loc], total fp32 [K, n_loc], y + noise | None)."""
    _dev(y, torch.float32, "y")
    n_loc, M = y.shape
    if noise is not None and _dev(noise, torch.float32, "noise").shape != y.shape:
        raise MpcError("prior_bits_network: noise must have the latent's shape")
    arr, keep = _param_array(params, K * M)
    idx = torch.empty(n_loc, dtype=torch.int32, device=y.device)
    minbits = torch.empty(n_loc, dtype=torch.float32, device=y.device)
    total = torch.empty((K, n_loc), dtype=torch.float32, device=y.device)
    ye = torch.empty_like(y) if want_entropy_latent else None
    context_for(y.device).call("mpc_prior_bits_network", arr, K, M, ptr(y), ptr(noise), n_loc, ptr(ye), ptr(total), ptr(idx),
                               ptr(minbits))
    return idx, minbits, total, ye


def gather_cdf(table: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """table int16 [K,M,L], idx int32 [n_loc] -> int16 [n_loc, M, L] (written into `out` when given)."""
    _dev(table, torch.int16, "table")
    _dev(idx, torch.int32, "idx")
    K, M, L = table.shape
    n_loc = idx.numel()
    if out is None:
        out = torch.empty((n_loc, M, L), dtype=torch.int16, device=table.device)
    elif _dev(out, torch.int16, "out").numel() != n_loc * M * L:
        raise MpcError("gather_cdf: out has the wrong size")
    with _timed("gather_cdf", bytes=float(n_loc) * M * L * 2):
        context_for(table.device).call("mpc_gather_cdf", ptr(table), K, M, L, ptr(idx), n_loc, ptr(out))
    return out


# ---- entropy coder ----------------------------------------------------------------------------
MRS1_MAX_SYMBOLS_PER_STREAM = 65533
MRS1_STREAM_OVERHEAD = 6          # 2-byte length + 4-byte final state
MRS1_FIXED_HEADER = 16


def choose_streams(n_loc: int, M: int, est_bytes: float, budget: float = 0.004, max_streams: int = 4096) -> Tuple[int, int]:
    """(n_streams, locs_per_stream): as many streams as fit in `budget` of the estimated payload, at least
    what the 16-bit stream-length field requires, at most one location per stream."""
    min_streams = -(-n_loc // max(1, MRS1_MAX_SYMBOLS_PER_STREAM // M))
    w = est_bytes * budget - MRS1_FIXED_HEADER
    want = int(math.floor(w / MRS1_STREAM_OVERHEAD)) if w > 0.0 else 0       # (NaN: no extra streams) same rule on the device
    n = max(min_streams, min(want, max_streams, n_loc), 1)
    lps = -(-n_loc // n)
    n = -(-n_loc // lps)                 # drop streams that would be empty
    return n, lps


def rans_encode(table: torch.Tensor, idx: torch.Tensor, sym: torch.Tensor, n_images: int, n_streams: int,
                locs_per_stream: int, out_stride: Optional[int] = None):
    """table int16 [K,M,L]; idx int32 [n_images*n_loc]; sym int16 [n_images*n_loc*M].
    -> (out uint8 [n_images, out_stride], nbytes uint64-as-int64 [n_images], status int32 [n_images])."""
    _dev(table, torch.int16, "table"); _dev(idx, torch.int32, "idx"); _dev(sym, torch.int16, "sym")
    K, M, L = table.shape
    n_loc = idx.numel() // n_images
    if sym.numel() != idx.numel() * M:
        raise MpcError("rans_encode: sym must hold M symbols per location")
    if out_stride is None:
        out_stride = MRS1_FIXED_HEADER + n_streams * MRS1_STREAM_OVERHEAD + 2 * n_loc * M
    out_stride = (out_stride + 3) // 4 * 4
    dev = table.device
    ctx = context_for(dev)
    ws_bytes = ctx.lib.mpc_rans_workspace_bytes(n_images, n_loc, M, n_streams)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.empty((n_images, out_stride), dtype=torch.uint8, device=dev)
    nbytes = torch.empty(n_images, dtype=torch.int64, device=dev)
    status = torch.empty(n_images, dtype=torch.int32, device=dev)
    with _timed("rans_encode", symbols=n_images * n_loc * M):
        ctx.call("mpc_rans_encode", ptr(table), K, M, L, ptr(idx), ptr(sym), n_images, n_loc, n_streams, locs_per_stream,
                 ptr(ws), ws_bytes, ptr(out), out_stride, ptr(nbytes), ptr(status))
    return out, nbytes, status


MRS1_STREAM_BUDGET = 0.004       # share of the estimated payload that the per-stream overhead may take
MRS1_MAX_STREAMS = 4096


def rans_encode_auto(table: torch.Tensor, idx: torch.Tensor, sym: torch.Tensor, minbits: torch.Tensor, n_images: int,
                     nbytes: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None,
                     geometry: Optional[torch.Tensor] = None, budget: float = MRS1_STREAM_BUDGET,
                     max_streams: int = MRS1_MAX_STREAMS):
    """rans_encode with the stream geometry chosen on the device from `minbits` (the competition's minima), by the rule
    of choose_streams: nothing is read back before the coder is launched.  nbytes (int64 [n_images]), status (int32
    [n_images]) and geometry (int32 [2] = n_streams, locs_per_stream) may be caller-provided device tensors (slots of one
    readback buffer).  -> (out uint8 [n_images, out_stride], nbytes, status, geometry)."""
    _dev(table, torch.int16, "table"); _dev(idx, torch.int32, "idx"); _dev(sym, torch.int16, "sym")
    _dev(minbits, torch.float32, "minbits")
    K, M, L = table.shape
    n_loc = idx.numel() // n_images
    if sym.numel() != idx.numel() * M or minbits.numel() != idx.numel():
        raise MpcError("rans_encode_auto: sym must hold M symbols and minbits one value per location")
    dev = table.device
    max_streams = max(1, min(max_streams, n_loc))
    out_stride = (MRS1_FIXED_HEADER + max_streams * MRS1_STREAM_OVERHEAD + 2 * n_loc * M + 3) // 4 * 4
    ctx = context_for(dev)
    ws_bytes = ctx.lib.mpc_rans_auto_workspace_bytes(n_images, n_loc, M, max_streams)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.empty((n_images, out_stride), dtype=torch.uint8, device=dev)
    nbytes = torch.empty(n_images, dtype=torch.int64, device=dev) if nbytes is None else _dev(nbytes, torch.int64, "nbytes")
    status = torch.empty(n_images, dtype=torch.int32, device=dev) if status is None else _dev(status, torch.int32, "status")
    geometry = torch.empty(2, dtype=torch.int32, device=dev) if geometry is None else _dev(geometry, torch.int32, "geometry")
    with _timed("rans_encode", symbols=n_images * n_loc * M):
        ctx.call("mpc_rans_encode_auto", ptr(table), K, M, L, ptr(idx), ptr(sym), ptr(minbits), n_images, n_loc,
                 ctypes.c_double(budget), max_streams, ptr(ws), ws_bytes, ptr(out), out_stride, ptr(nbytes), ptr(status), ptr(geometry))
    return out, nbytes, status, geometry


def download_rows(src: torch.Tensor, width: int, dst_host: torch.Tensor) -> None:
    """Asynchronous copy of the first `width` bytes of every row of the 2-D uint8 device tensor `src` into the
    (page-locked) host tensor dst_host [rows, width], on the current stream."""
    _dev(src, torch.uint8, "src")
    rows = src.shape[0]
    if dst_host.is_cuda or dst_host.dtype != torch.uint8 or dst_host.numel() < rows * width or not dst_host.is_contiguous():
        raise MpcError("download_rows: dst_host must be a contiguous uint8 host tensor of rows*width bytes")
    context_for(src.device).call("mpc_download_rows", ptr(src), src.stride(0), width, rows, ptr(dst_host), width)


def symbols_to_latent_f16(sym: torch.Tensor, min_feat: int, q_intv: int) -> torch.Tensor:
    _dev(sym, torch.int16, "sym")
    out = torch.empty(sym.shape, dtype=torch.float16, device=sym.device)
    context_for(sym.device).call("mpc_symbols_to_latent_f16", ptr(sym), sym.numel(), int(min_feat), int(q_intv), ptr(out))
    return out


def rans_decode(table: torch.Tensor, idx: torch.Tensor, data: torch.Tensor, nbytes: torch.Tensor, n_images: int,
                n_streams: int, locs_per_stream: int):
    """data uint8 [n_images, stride] on the device -> (sym int16 [n_images, n_loc*M], status int32 [n_images])."""
    _dev(table, torch.int16, "table"); _dev(idx, torch.int32, "idx"); _dev(data, torch.uint8, "data")
    _dev(nbytes, torch.int64, "nbytes")
    K, M, L = table.shape
    n_loc = idx.numel() // n_images
    dev = table.device
    sym = torch.empty((n_images, n_loc * M), dtype=torch.int16, device=dev)
    status = torch.empty(n_images, dtype=torch.int32, device=dev)
    stride = data.shape[-1] if data.dim() == 2 else data.numel() // n_images
    with _timed("rans_decode", symbols=n_images * n_loc * M):
        context_for(dev).call("mpc_rans_decode", ptr(table), K, M, L, ptr(idx), ptr(data), stride, ptr(nbytes), n_images,
                              n_loc, n_streams, locs_per_stream, ptr(sym), ptr(status))
    return sym, status


def rans_decode_latent_f16(table: torch.Tensor, idx: torch.Tensor, data: torch.Tensor, nbytes: torch.Tensor, n_images: int,
                           n_streams: int, locs_per_stream: int, min_feat: int, q_intv: int):
    """rans_decode whose output is the fp16 latent (symbol * q_intv + min_feat) the fp16 synthesis transform reads:
    -> (latent fp16 [n_images, n_loc*M], status int32 [n_images])."""
    _dev(table, torch.int16, "table"); _dev(idx, torch.int32, "idx"); _dev(data, torch.uint8, "data")
    _dev(nbytes, torch.int64, "nbytes")
    K, M, L = table.shape
    n_loc = idx.numel() // n_images
    dev = table.device
    latent = torch.empty((n_images, n_loc * M), dtype=torch.float16, device=dev)
    status = torch.empty(n_images, dtype=torch.int32, device=dev)
    stride = data.shape[-1] if data.dim() == 2 else data.numel() // n_images
    with _timed("rans_decode", symbols=n_images * n_loc * M):
        context_for(dev).call("mpc_rans_decode_latent_f16", ptr(table), K, M, L, ptr(idx), ptr(data), stride, ptr(nbytes), n_images,
                              n_loc, n_streams, locs_per_stream, int(min_feat), int(q_intv), ptr(latent), ptr(status))
    return latent, status


def rans_decode_noalloc(table: torch.Tensor, idx: torch.Tensor, data: torch.Tensor, nbytes: torch.Tensor, n_images: int,
                        n_streams: int, locs_per_stream: int, sym_out: torch.Tensor, status_out: torch.Tensor) -> None:
    """mpc_rans_decode_noalloc: the decode of rans_decode into caller-provided sym_out (int16 [n_images, n_loc*M]) and
    status_out (int32 [n_images]); neither this wrapper nor the library allocates device memory.  Standard tables
    only (L = 257, K*M <= 16384); calls that overlap in time on one GPU race on the library's static scratch."""
    _dev(table, torch.int16, "table"); _dev(idx, torch.int32, "idx"); _dev(data, torch.uint8, "data")
    _dev(nbytes, torch.int64, "nbytes"); _dev(sym_out, torch.int16, "sym_out"); _dev(status_out, torch.int32, "status_out")
    K, M, L = table.shape
    n_loc = idx.numel() // n_images
    if sym_out.numel() != n_images * n_loc * M or status_out.numel() != n_images:
        raise MpcError("rans_decode_noalloc: sym_out / status_out have the wrong size")
    stride = data.shape[-1] if data.dim() == 2 else data.numel() // n_images
    context_for(table.device).call("mpc_rans_decode_noalloc", ptr(table), K, M, L, ptr(idx), ptr(data), stride, ptr(nbytes),
                                   n_images, n_loc, n_streams, locs_per_stream, ptr(sym_out), ptr(status_out))


# ---- image-quality statistics ------------------------------------------------------------------------
def ssim_stats(x: torch.Tensor, y: torch.Tensor, window: Sequence[float], C1: float, C2: float):
    """x, y NCHW fp32 -> (mean SSIM map, mean contrast-structure map) per (image, channel), float64 [B, C]."""
    _dev(x, torch.float32, "x"); _dev(y, torch.float32, "y")
    if x.shape != y.shape or x.dim() != 4:
        raise MpcError("ssim_stats: x and y must be NCHW tensors of one shape")
    B, C, H, W = x.shape
    n = len(window)
    ss = torch.empty(B * C, dtype=torch.float64, device=x.device)
    cs = torch.empty(B * C, dtype=torch.float64, device=x.device)
    win = (ctypes.c_float * n)(*[float(v) for v in window])
    with _timed("ssim_stats", bytes=2.0 * x.numel() * 4):
        context_for(x.device).call("mpc_ssim_stats", ptr(x), ptr(y), B * C, H, W, win, n, ctypes.c_float(C1), ctypes.c_float(C2),
                                   ptr(ss), ptr(cs))
    area = float((H - n + 1) * (W - n + 1))
    return (ss / area).view(B, C), (cs / area).view(B, C)


def avg_pool2(x: torch.Tensor) -> torch.Tensor:
    """F.avg_pool2d(x, kernel_size=2, padding=(H % 2, W % 2)) of an NCHW fp32 tensor."""
    _dev(x, torch.float32, "x")
    B, C, H, W = x.shape
    Ho, Wo = (H + 2 * (H % 2) - 2) // 2 + 1, (W + 2 * (W % 2) - 2) // 2 + 1
    out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
    context_for(x.device).call("mpc_avg_pool2", ptr(x), B * C, H, W, ptr(out))
    return out


def containers_to_bytes(encoded: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]) -> List[List[bytes]]:
    """[(out uint8 [n_images, stride], nbytes int64 [n_images], status int32 [n_images]), ...] of encoder calls that have
    been enqueued -> per call the list of its images' MRS1 containers.  Two readbacks whatever the number of calls:
    all sizes and statuses, then all payloads."""
    counts = [e[1].numel() for e in encoded]
    meta = torch.cat([t for out, nbytes, status in encoded for t in (nbytes, status.to(torch.int64))]).cpu().tolist()
    sizes, at = [], 0
    for n in counts:
        nb, st = meta[at:at + n], meta[at + n:at + 2 * n]
        at += 2 * n
        if min(st) < 0:
            raise MpcError(f"rANS encoder reported {st} (symbol outside the table or zero frequency)")
        sizes.append(nb)
    used = [max(nb) for nb in sizes]
    host = torch.cat([e[0][:, :u].reshape(-1) for e, u in zip(encoded, used)]).cpu().numpy()
    res, at = [], 0
    for n, nb, u in zip(counts, sizes, used):
        rows = host[at:at + n * u].reshape(n, u)
        at += n * u
        res.append([rows[i, :nb[i]].tobytes() for i in range(n)])
    return res


def rans_encode_to_bytes(table: torch.Tensor, idx: torch.Tensor, sym: torch.Tensor, n_images: int,
                         est_bytes_per_image: float, budget: float = 0.004) -> List[bytes]:
    """Codes n_images equally shaped images and returns their MRS1 containers as byte strings.  The stream count is
    choose_streams(n_loc, M, est_bytes_per_image, budget)."""
    K, M, L = table.shape
    n_loc = idx.numel() // n_images
    n_streams, lps = choose_streams(n_loc, M, est_bytes_per_image, budget)
    return containers_to_bytes([rans_encode(table, idx.reshape(-1), sym.reshape(-1), n_images, n_streams, lps)])[0]


def parse_mrs1_headers(streams: Sequence[bytes], n_loc: int, M: int) -> Tuple[int, int]:
    """Validates the 16-byte headers of MRS1 containers that are to be decoded in one call and returns the shared
    (n_streams, locs_per_stream).  The header of a file is untrusted input: everything the device kernels derive
    addresses from is checked here (magic, 0 < n_streams <= n_loc, the symbol count of the image, a whole number
    of locations per stream, the length table inside the container)."""
    import numpy as np
    geo = None
    for i, s in enumerate(streams):
        if len(s) < MRS1_FIXED_HEADER:
            raise ValueError(f"MRS1 container {i}: {len(s)} bytes is shorter than the fixed header")
        magic, n_streams, n_symbols, sps = (int(v) for v in np.frombuffer(s[:16], dtype="<u4"))
        if magic != 0x3153524D:
            raise ValueError(f"MRS1 container {i}: bad magic {magic:#x}")
        if not 0 < n_streams <= n_loc:
            raise ValueError(f"MRS1 container {i}: {n_streams} streams for {n_loc} locations")
        if n_symbols != n_loc * M:
            raise ValueError(f"MRS1 container {i}: holds {n_symbols} symbols, the image has {n_loc * M}")
        if sps == 0 or sps % M or sps > MRS1_MAX_SYMBOLS_PER_STREAM or n_streams * sps < n_symbols:
            raise ValueError(f"MRS1 container {i}: {sps} symbols per stream does not fit {n_streams} streams x {M} channels")
        if len(s) < MRS1_FIXED_HEADER + 2 * n_streams:
            raise ValueError(f"MRS1 container {i}: truncated inside the stream-length table")
        if geo is None:
            geo = (n_streams, sps // M)
        elif geo != (n_streams, sps // M):
            raise ValueError("all containers of one decode call must share one stream geometry")
    if geo is None:
        raise ValueError("no MRS1 container given")
    return geo


def rans_decode_from_bytes(table: torch.Tensor, idx: torch.Tensor, streams: List[bytes]) -> torch.Tensor:
    """Inverse of rans_encode_to_bytes: -> int16 symbols [n_images, n_loc*M].  The stream geometry is read from the
    containers' headers (all images of the call must share it)."""
    import numpy as np
    K, M, L = table.shape
    B = len(streams)
    n_streams, lps = parse_mrs1_headers(streams, idx.numel() // max(B, 1), M)
    stride = (max(len(s) for s in streams) + 3) // 4 * 4
    host = np.zeros((B, stride), np.uint8)
    for i, s in enumerate(streams):
        host[i, :len(s)] = np.frombuffer(s, np.uint8)
    dev = table.device
    data = torch.from_numpy(host).to(dev)
    nbytes = torch.tensor([len(s) for s in streams], dtype=torch.int64, device=dev)
    sym, status = rans_decode(table, idx.reshape(-1), data, nbytes, B, n_streams, lps)
    if int(status.min().item()) < 0:
        raise MpcError(f"corrupt MRS1 stream: decoder status {status.cpu().tolist()}")
    return sym


def profile_mark(device, mark_id: int) -> None:
    """One-thread marker kernel on torch's current stream of `device` (shows up in a kernel trace as
    mpc_profile_mark_kernel)."""
    context_for(device).call("mpc_profile_mark", int(mark_id))


# ---- coder over per-element Laplace rows (never materialised) ---------------------------------------
def _upload_containers(streams: Sequence[bytes], dev):
    """Byte strings -> (uint8 [B, stride] on the device, their lengths as int64 on the device)."""
    import numpy as np
    stride = (max(len(s) for s in streams) + 3) // 4 * 4
    host = np.zeros((len(streams), stride), np.uint8)
    for i, s in enumerate(streams):
        host[i, :len(s)] = np.frombuffer(s, np.uint8)
    return torch.from_numpy(host).to(dev), torch.tensor([len(s) for s in streams], dtype=torch.int64, device=dev)


def rans_encode_laplace(sigma: torch.Tensor, sym: torch.Tensor, n_images: int, M: int, first_target: float, L: int,
                        precision: int, n_streams: int, locs_per_stream: int):
    """sigma fp32 and sym int16, both [n_images, n_loc, M] (NHWC): codes symbol i under the int16 CDF row of
    Laplace(0, sigma[i]) -- the rows of laplace_cdf_rows, evaluated where they are needed instead of stored.
    -> (out uint8 [n_images, out_stride], nbytes int64 [n_images], status int32 [n_images])."""
    _dev(sigma, torch.float32, "sigma"); _dev(sym, torch.int16, "sym")
    if sigma.numel() != sym.numel() or sym.numel() % (n_images * M):
        raise MpcError("rans_encode_laplace: sigma and sym must both hold n_images * n_loc * M elements")
    n_loc = sym.numel() // (n_images * M)
    out_stride = (MRS1_FIXED_HEADER + n_streams * MRS1_STREAM_OVERHEAD + 2 * n_loc * M + 3) // 4 * 4
    dev = sym.device
    ctx = context_for(dev)
    ws_bytes = ctx.lib.mpc_rans_workspace_bytes(n_images, n_loc, M, n_streams)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.empty((n_images, out_stride), dtype=torch.uint8, device=dev)
    nbytes = torch.empty(n_images, dtype=torch.int64, device=dev)
    status = torch.empty(n_images, dtype=torch.int32, device=dev)
    with _timed("rans_encode_laplace", symbols=n_images * n_loc * M):
        ctx.call("mpc_rans_encode_laplace", ptr(sigma), ptr(sym), ctypes.c_float(first_target), L, precision, n_images, n_loc, M,
                 n_streams, locs_per_stream, ptr(ws), ws_bytes, ptr(out), out_stride, ptr(nbytes), ptr(status))
    return out, nbytes, status


def rans_decode_laplace(sigma: torch.Tensor, data: torch.Tensor, nbytes: torch.Tensor, n_images: int, M: int,
                        first_target: float, L: int, precision: int, n_streams: int, locs_per_stream: int):
    """Inverse of rans_encode_laplace with the same sigma: -> (sym int16 [n_images, n_loc*M], status int32 [n_images])."""
    _dev(sigma, torch.float32, "sigma"); _dev(data, torch.uint8, "data"); _dev(nbytes, torch.int64, "nbytes")
    n_loc = sigma.numel() // (n_images * M)
    dev = sigma.device
    sym = torch.empty((n_images, n_loc * M), dtype=torch.int16, device=dev)
    status = torch.empty(n_images, dtype=torch.int32, device=dev)
    stride = data.shape[-1] if data.dim() == 2 else data.numel() // n_images
    with _timed("rans_decode_laplace", symbols=n_images * n_loc * M):
        context_for(dev).call("mpc_rans_decode_laplace", ptr(sigma), ctypes.c_float(first_target), L, precision, ptr(data), stride,
                              ptr(nbytes), n_images, n_loc, M, n_streams, locs_per_stream, ptr(sym), ptr(status))
    return sym, status


def rans_encode_laplace_to_bytes(sigma: torch.Tensor, sym: torch.Tensor, n_images: int, M: int, first_target: float, L: int,
                                 precision: int, est_bytes_per_image: float) -> List[bytes]:
    n_loc = sym.numel() // (n_images * M)
    n_streams, lps = choose_streams(n_loc, M, est_bytes_per_image)
    return containers_to_bytes([rans_encode_laplace(sigma, sym, n_images, M, first_target, L, precision, n_streams, lps)])[0]


def rans_decode_laplace_from_bytes(sigma: torch.Tensor, streams: List[bytes], M: int, first_target: float, L: int,
                                   precision: int) -> torch.Tensor:
    """-> int16 symbols [n_images, n_loc*M]; the stream geometry is read from the (validated) container headers."""
    B = len(streams)
    n_loc = sigma.numel() // (max(B, 1) * M)
    n_streams, lps = parse_mrs1_headers(streams, n_loc, M)
    data, nbytes = _upload_containers(streams, sigma.device)
    sym, status = rans_decode_laplace(sigma, data, nbytes, B, M, first_target, L, precision, n_streams, lps)
    if int(status.min().item()) < 0:
        raise MpcError(f"corrupt MRS1 stream: decoder status {status.cpu().tolist()}")
    return sym


# ---- hyperprior likelihoods -----------------------------------------------------------------------
def laplace_bits_total(yhat: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    _dev(yhat, torch.float32, "yhat"); _dev(sigma, torch.float32, "sigma")
    total = torch.zeros(1, dtype=torch.float64, device=yhat.device)
    context_for(yhat.device).call("mpc_laplace_bits", ptr(yhat), ptr(sigma), yhat.numel(), None, ptr(total))
    return total


def prob_bits_total(hi: torch.Tensor, lo: torch.Tensor) -> torch.Tensor:
    _dev(hi, torch.float32, "hi"); _dev(lo, torch.float32, "lo")
    total = torch.zeros(1, dtype=torch.float64, device=hi.device)
    context_for(hi.device).call("mpc_prob_bits", ptr(hi), ptr(lo), hi.numel(), None, ptr(total))
    return total


def laplace_cdf_rows(sigma: torch.Tensor, first_target: float, L: int, precision: int = 16) -> torch.Tensor:
    _dev(sigma, torch.float32, "sigma")
    out = torch.empty((sigma.numel(), L), dtype=torch.int16, device=sigma.device)
    context_for(sigma.device).call("mpc_laplace_cdf_rows", ptr(sigma), sigma.numel(), ctypes.c_float(first_target), L,
                                   precision, ptr(out))
    return out


# ---- transform layers ----------------------------------------------------------------------------
class ConvLayer:
    """Owns one mpc_conv (packed weights on the device).  Built from PyTorch-layout parameter tensors."""

    def __init__(self, kind: int, weight: torch.Tensor, bias: torch.Tensor, ksize: int, stride: int, act: int = 0,
                 beta: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None, in_abs: bool = False,
                 f16: bool = False, x3: bool = False, in_bound: float = 0.0):
        w = _dev(weight.detach().contiguous(), torch.float32, "weight")
        b = _dev(bias.detach().contiguous(), torch.float32, "bias")
        if kind == _lib.MPC_CONV:
            cout, cin = w.shape[0], w.shape[1]
        else:
            cin, cout = w.shape[0], w.shape[1]
        if w.shape[2] != ksize or w.shape[3] != ksize or b.numel() != cout:
            raise MpcError(f"weight {tuple(w.shape)} / bias {tuple(b.shape)} do not match ksize {ksize}")
        be = ga = None
        if act in (_lib.MPC_ACT_GDN, _lib.MPC_ACT_IGDN):
            be = _dev(beta.detach().contiguous(), torch.float32, "beta")
            ga = _dev(gamma.detach().contiguous(), torch.float32, "gamma")
            if be.numel() != cout or tuple(ga.shape) != (cout, cout):
                raise MpcError("GDN parameter shapes do not match cout")
        self.desc = ConvDesc(kind, cin, cout, ksize, stride, (ksize - 1) // 2,
                             stride - 1 if kind == _lib.MPC_DECONV else 0, act,
                             (_lib.MPC_IN_ABS if in_abs else 0) | (_lib.MPC_F16 if f16 else 0) |
                             (_lib.MPC_F16X3 if x3 else 0), float(in_bound))
        self.f16, self.x3 = f16, x3
        self.device = w.device
        self.cin, self.cout, self.kind, self.stride, self.ksize = cin, cout, kind, stride, ksize
        self.small_cout = kind == _lib.MPC_DECONV and stride == 2 and cout * 4 <= 32
        self.act = act
        gdn = act in (_lib.MPC_ACT_GDN, _lib.MPC_ACT_IGDN)
        # name of the device kernel template this layer launches (matches the rocprof kernel trace)
        tiles = 1 if self.small_cout else cout // 32
        nf = tiles // (2 if tiles > 6 else 1)
        epi = 2 if self.small_cout else (1 if gdn else 0)
        self._name_fmt = ("conv_igemm_x3_kernel<%d, %d, %s, %%d>" % (nf, epi, "true" if cin <= 4 else "false")) if x3 else \
            ("conv_igemm_f16_kernel<%d, %d, %%d>" % (nf, epi)) if f16 else \
            ("conv_igemm_kernel<%d, %s, %d>" % (nf, "true" if cin <= 4 else "false", epi))
        # f16x3 5x5 stride-2 layers with cin > 4 run on the LDS-patch kernel (same conditions as the library's dispatch)
        self._patch = x3 and cin > 4 and kind == _lib.MPC_CONV and ksize == 5 and stride == 2 and cin % 32 == 0 and nf in (4, 6) \
            and (not gdn or in_bound > 0.0)
        if self._patch:
            self._name_fmt = "conv5s2_patch_x3_kernel<%d, %d, 8>" % (nf, epi)
        # the 3-channel first layer with a fused GDN has a patch kernel of its own
        if x3 and cin <= 4 and kind == _lib.MPC_CONV and ksize == 5 and stride == 2 and act == _lib.MPC_ACT_GDN and in_bound > 0.0 \
                and nf in (4, 6) and tiles == nf:
            self._patch = True
            self._name_fmt = "conv1_rgb_patch_x3_kernel<%d>" % nf
        # fp16 5x5 stride-2 transposed convolutions run on their own LDS-patch kernel (the library's dispatch conditions)
        self._dpatch = f16 and kind == _lib.MPC_DECONV and ksize == 5 and stride == 2 and cin % 64 == 0 and nf in (4, 6) \
            and not self.small_cout and cout <= 192
        if self._dpatch:
            self._patch = True
            self._name_fmt = "deconv5s2_patch_f16_kernel<%d, %d>" % (nf, epi)
        self._launches = (tiles // nf) * (4 if (kind == _lib.MPC_DECONV and stride == 2 and not self.small_cout) else 1)
        self.kernel_name = self._name_fmt if (self._patch or not (x3 or f16)) else self._name_fmt % 4
        # exact f32 layers without GDN run one 32-channel tile per workgroup group when the launch has fewer workgroups
        # than the GPU has CUs (the library's rule, mirrored for the kernel name in profiles)
        self._narrow_ok = not (x3 or f16 or gdn or self.small_cout) and cin > 4 and tiles > 1
        # the fp16 last layer (cout <= 4) has a kernel of its own
        self._rgb16 = f16 and self.small_cout and cout <= 4 and cin % 64 == 0 and cin <= 256
        ctx = context_for(w.device)
        h = c_void_p()
        ctx.check(ctx.lib.mpc_conv_create(ctx.handle, ctypes.byref(self.desc), ptr(w), ptr(b), ptr(be), ptr(ga), ctypes.byref(h)))
        self.handle = h
        self._lib = ctx.lib
        _lib.register_owner(self)

    def out_hw(self, H: int, W: int) -> Tuple[int, int]:
        ho, wo = ctypes.c_int(), ctypes.c_int()
        self._lib.mpc_conv_out_hw(self.handle, H, W, ctypes.byref(ho), ctypes.byref(wo))
        return ho.value, wo.value

    def __call__(self, x: torch.Tensor, nchw_out: bool = False, clamp01: bool = False, round_out: bool = False,
                 keep_unrounded: bool = True, out_split: bool = False, wide_block: bool = False):
        """x: NHWC [B,H,W,cin (3 channels padded to 4)] fp32; fp16 for an f16 layer; the split-fp16 "S16" tensor
        [B,H,W,cin/8,16] fp16 for an f16x3 layer.  Returns NHWC (or NCHW) output of the layer's activation type
        (the small-cout RGB layer always returns fp32 NCHW; out_split returns an S16 tensor [B,Ho,Wo,cout/8,16]);
        with round_out returns (unrounded | None, rounded).  wide_block forces the 8-wave workgroups of the
        fp16 / f16x3 kernels (normally chosen by problem size; same results)."""
        adt = torch.float16 if (self.f16 or self.x3) else torch.float32
        if self.x3 and self.cin <= 4:
            adt = torch.float32
        _dev(x, adt, "x")
        if self.x3 and self.cin > 4:
            if x.dim() != 5 or x.shape[3] * 8 != self.cin or x.shape[4] != 16:
                raise MpcError(f"f16x3 layer expects a split tensor [B,H,W,{self.cin // 8},16], got {tuple(x.shape)}")
            B, H, W = x.shape[:3]
        else:
            B, H, W, C = x.shape
            if C != (4 if self.cin <= 4 else self.cin):
                raise MpcError(f"input has {C} channels, layer expects {self.cin}")
        Ho, Wo = self.out_hw(H, W)
        nchw_out = nchw_out or self.small_cout
        flags = (_lib.MPC_OUT_NCHW if nchw_out else 0) | (_lib.MPC_OUT_CLAMP01 if clamp01 else 0) | \
                (_lib.MPC_OUT_ROUND if round_out else 0) | (_lib.MPC_OUT_SPLIT if out_split else 0) | \
                (_lib.MPC_WIDE_BLOCK if wide_block else 0)
        if self.f16 or self.x3:
            # mirrors the library's choice of waves per workgroup (for the kernel name in profiles)
            base = B * (H * W if self.kind == _lib.MPC_DECONV else Ho * Wo)
            wide = (wide_block or -(-base // 256) * self._launches >= 512) and not self.small_cout
            self.kernel_name = "deconv_rgb16_kernel" if self._rgb16 else self._name_fmt if self._patch else self._name_fmt % (8 if wide else 4)
        elif self._narrow_ok:
            base = B * (H * W if self.kind == _lib.MPC_DECONV else Ho * Wo)
            self.kernel_name = "conv_igemm_kernel<1, false, 0>" if -(-base // 128) * self._launches < 256 else self._name_fmt
        if out_split:
            shape, odt = (B, Ho, Wo, self.cout // 8, 16), torch.float16
        else:
            shape = (B, self.cout, Ho, Wo) if nchw_out else (B, Ho, Wo, self.cout)
            odt = torch.float32 if (self.small_cout or self.x3) else adt
        out = torch.empty(shape, dtype=odt, device=x.device) if (keep_unrounded or not round_out) else None
        outr = torch.empty(shape, dtype=torch.float32, device=x.device) if round_out else None
        with _timed(self.kernel_name, flops=self.flops(B, H, W)):
            context_for(x.device).call("mpc_conv_forward", self.handle, ptr(x), B, H, W, flags, ptr(out), ptr(outr))
        return (out, outr) if round_out else out

    def forward_scored(self, x: torch.Tensor, target: torch.Tensor):
        """The last synthesis layer with the evaluation's clamp and squared error in its epilogue: -> (clamp(x_hat, 0, 1)
        fp32 NCHW, float64 [B] sums of squared errors against `target`), or None when this layer / size has no such
        kernel (then: the plain call followed by clamp_sse_per_image)."""
        if not (self.f16 and self.small_cout):
            return None
        _dev(x, torch.float16, "x"); _dev(target, torch.float32, "target")
        B, H, W, C = x.shape
        if C != self.cin:
            raise MpcError(f"input has {C} channels, layer expects {self.cin}")
        if not self._lib.mpc_conv_scored_supported(self.handle, H, W):
            return None
        Ho, Wo = self.out_hw(H, W)
        if tuple(target.shape) != (B, self.cout, Ho, Wo):
            raise MpcError(f"forward_scored: target must be {(B, self.cout, Ho, Wo)}, got {tuple(target.shape)}")
        xhat = torch.empty((B, self.cout, Ho, Wo), dtype=torch.float32, device=x.device)
        sse = torch.empty(B, dtype=torch.float64, device=x.device)
        self.kernel_name = "deconv_rgb16_kernel"
        with _timed(self.kernel_name, flops=self.flops(B, H, W)):
            context_for(x.device).call("mpc_conv_forward_scored", self.handle, ptr(x), B, H, W, ptr(xhat), ptr(target), ptr(sse))
        return xhat, sse

    def forward_symbols(self, x: torch.Tensor, min_feat: int, n_sym_values: int, n_clamped: Optional[torch.Tensor] = None,
                        keep_rounded: bool = False):
        """The last analysis layer with quantisation and symbolisation fused: -> (round(y) fp32 NHWC | None, int16 symbols
        NHWC, int32[1] count of clamped values).  `n_clamped` may be a caller-provided int32 device tensor (a slot of a
        larger readback buffer)."""
        if self.f16 or self.small_cout or self.act != _lib.MPC_ACT_NONE:
            raise MpcError("forward_symbols needs a layer without activation on the f32 or f16x3 path")
        if self.x3 and self.cin > 4:
            _dev(x, torch.float16, "x")
            if x.dim() != 5 or x.shape[3] * 8 != self.cin or x.shape[4] != 16:
                raise MpcError(f"f16x3 layer expects a split tensor [B,H,W,{self.cin // 8},16], got {tuple(x.shape)}")
        else:
            _dev(x, torch.float32, "x")
            if x.dim() != 4 or x.shape[3] != (4 if self.cin <= 4 else self.cin):
                raise MpcError(f"input {tuple(x.shape)} does not match the layer's {self.cin} input channels")
        B, H, W = x.shape[:3]
        Ho, Wo = self.out_hw(H, W)
        sym = torch.empty((B, Ho, Wo, self.cout), dtype=torch.int16, device=x.device)
        outr = torch.empty((B, Ho, Wo, self.cout), dtype=torch.float32, device=x.device) if keep_rounded else None
        if n_clamped is None:
            n_clamped = torch.empty(1, dtype=torch.int32, device=x.device)
        _dev(n_clamped, torch.int32, "n_clamped")
        if self.x3 and not self._patch:
            base = B * Ho * Wo
            wide = -(-base // 256) * self._launches >= 512
            self.kernel_name = self._name_fmt % (8 if wide else 4)
        with _timed(self.kernel_name, flops=self.flops(B, H, W)):
            context_for(x.device).call("mpc_conv_forward_symbols", self.handle, ptr(x), B, H, W, int(min_feat), int(n_sym_values),
                                       ptr(outr), ptr(sym), ptr(n_clamped))
        return outr, sym, n_clamped

    def flops(self, B: int, H: int, W: int) -> float:
        """Algorithmic FLOPs of one forward: 2 per multiply-accumulate of the convolution (each real tap
        counted once: k*k*cin*cout per output pixel of a Conv2d, per INPUT pixel of a ConvTranspose2d) plus
        2*cout^2 per output pixel for a fused GDN/IGDN."""
        Ho, Wo = self.out_hw(H, W)
        base = B * (Ho * Wo if self.kind == _lib.MPC_CONV else H * W)
        fl = 2.0 * base * self.ksize * self.ksize * self.cin * self.cout
        if self.act in (_lib.MPC_ACT_GDN, _lib.MPC_ACT_IGDN):
            fl += 2.0 * B * Ho * Wo * self.cout * self.cout
        return fl

    def close(self):
        if getattr(self, "handle", None):
            handle, self.handle = self.handle, None
            self._lib.mpc_conv_destroy(handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
