"""ctypes binding of libmpc.so (the C ABI declared in include/mpc.h).

The HIP library is the only compute path of this package: if it is missing, or no GPU is visible, the
operations raise -- there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import atexit
import ctypes
import os
import subprocess
import weakref
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPC_LIB selects another build of the same library (A/B measurements of experimental builds)
LIB_PATH = os.environ.get("MPC_LIB") or os.path.join(_HERE, "libmpc.so")

MPC_ORDER_ROUND_THEN_ADD = 0
MPC_ORDER_ADD_THEN_ROUND = 1
MPC_BITS_LOG2 = 0
MPC_BITS_LN = 1
MPC_CONV, MPC_DECONV = 0, 1
MPC_ACT_NONE, MPC_ACT_GDN, MPC_ACT_IGDN, MPC_ACT_RELU, MPC_ACT_EXP = 0, 1, 2, 3, 4
MPC_IN_ABS, MPC_OUT_ROUND, MPC_OUT_CLAMP01, MPC_OUT_NCHW, MPC_F16, MPC_F16X3, MPC_WIDE_BLOCK, MPC_OUT_SPLIT = 1, 2, 4, 8, 16, 32, 64, 128


class MpcError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("kind", "cin", "cout", "ksize", "stride", "pad", "out_pad", "act", "flags")] + \
               [("in_bound", c_float)]


# name -> (restype, argtypes); every symbol declared in include/mpc.h
SIGNATURES = {
    "mpc_create": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "mpc_destroy": (c_int, [c_void_p]),
    "mpc_set_stream": (c_int, [c_void_p, c_void_p]),
    "mpc_synchronize": (c_int, [c_void_p]),
    "mpc_last_error": (c_char_p, [c_void_p]),
    "mpc_version": (c_char_p, []),
    "mpc_profile_mark": (c_int, [c_void_p, c_int]),
    "mpc_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mpc_nhwc_to_nchw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mpc_latent_to_symbols": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mpc_symbols_to_latent": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "mpc_symbols_to_latent_f16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "mpc_download_rows": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t]),
    "mpc_sum_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mpc_gdn_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p]),
    "mpc_clamp_sse": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mpc_clamp_sse_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p]),
    "mpc_stats_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, POINTER(c_double), c_int, c_void_p]),
    "mpc_ssim_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_float), c_int, c_float, c_float,
                               c_void_p, c_void_p]),
    "mpc_avg_pool2": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "mpc_bitparm_cdf_table": (c_int, [c_void_p, POINTER(c_void_p), c_int64, c_float, c_int, c_void_p]),
    "mpc_bitparm_cdf_eval": (c_int, [c_void_p, POINTER(c_void_p), c_int64, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "mpc_renorm_cast_cdf": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "mpc_bits_lut": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mpc_prior_argmin": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mpc_prior_bits_network": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "mpc_gather_cdf": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "mpc_rans_workspace_bytes": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "mpc_rans_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_int,
                                c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mpc_rans_auto_workspace_bytes": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "mpc_rans_encode_auto": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_double,
                                     c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "mpc_rans_decode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_int,
                                c_int64, c_int, c_int, c_void_p, c_void_p]),
    "mpc_rans_decode_latent_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_int,
                                           c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mpc_rans_decode_noalloc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_int,
                                        c_int64, c_int, c_int, c_void_p, c_void_p]),
    "mpc_rans_encode_laplace": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int64, c_int, c_int, c_int,
                                        c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mpc_rans_decode_laplace": (c_int, [c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int, c_int64,
                                        c_int, c_int, c_int, c_void_p, c_void_p]),
    "mpc_conv_create": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "mpc_conv_destroy": (c_int, [c_void_p]),
    "mpc_conv_out_hw": (c_int, [c_void_p, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "mpc_conv_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mpc_conv_forward_symbols": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mpc_conv_scored_supported": (c_int, [c_void_p, c_int, c_int]),
    "mpc_conv_forward_scored": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mpc_f32_to_f16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mpc_f16_to_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mpc_laplace_bits": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mpc_prob_bits": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mpc_laplace_cdf_rows": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_int, c_int, c_void_p]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile libmpc.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force:
        import shutil
        shutil.rmtree(os.path.join(_HERE, "csrc", "build"), ignore_errors=True)
    subprocess.run(["bash", script], check=True)
    return LIB_PATH


def load():
    """dlopen libmpc.so and declare every prototype.  Raises MpcError if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MpcError(f"{LIB_PATH} is missing: build it with compression_amd/csrc/build.sh "
                           "(this package has no CPU fallback)")
        # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so) and the process must end up with
        # ONE runtime shared by torch's allocator/streams and by libmpc's kernels
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)           # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        atexit.register(shutdown)             # registered after torch's own hooks, so it runs before them
    return _lib


# Objects that own device memory through a library handle (Context, ops' layer objects).  They are closed from an
# atexit hook, i.e. while the interpreter, torch and the HIP runtime are all still up: a __del__ that runs during
# interpreter finalisation would call hipFree after other libraries' exit handlers (a profiler's tool library, the
# runtime's own teardown) may already have run, which ended a profiled run with a segmentation fault inside exit().
_owners = weakref.WeakSet()
_shut_down = False


def register_owner(obj):
    """obj.close() is called at interpreter exit if obj is still alive then."""
    _owners.add(obj)


def shutdown():
    """Frees every live handle now; later close() / __del__ calls find nothing left to free.  Idempotent."""
    global _shut_down
    if _shut_down:
        return
    _shut_down = True
    for obj in list(_owners):
        try:
            obj.close()
        except Exception:
            pass
    _contexts.clear()


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else c_void_p(t.data_ptr())


class Context:
    """One mpc_ctx per (device, HIP stream)."""

    def __init__(self, device_index: int = 0, stream_handle: int = 0):
        self.lib = load()
        h = c_void_p()
        rc = self.lib.mpc_create(device_index, c_void_p(stream_handle), ctypes.byref(h))
        if rc:
            raise MpcError(f"mpc_create(device {device_index}) failed with {rc}: no usable MI355X / HIP runtime")
        self.handle = h
        self.device_index = device_index
        register_owner(self)

    def set_stream(self, stream_handle: int):
        self.check(self.lib.mpc_set_stream(self.handle, c_void_p(stream_handle)))

    def check(self, rc: int):
        if rc:
            msg = self.lib.mpc_last_error(self.handle)
            raise MpcError(f"libmpc error {rc}: {msg.decode() if msg else ''}")

    def call(self, name: str, *args):
        self.check(getattr(self.lib, name)(self.handle, *args))

    def close(self):
        if getattr(self, "handle", None):
            handle, self.handle = self.handle, None
            self.lib.mpc_destroy(handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def context_for(device) -> Context:
    """Context bound to `device` (a torch.device) and to torch's current stream on it."""
    import torch

    if device.type != "cuda":
        raise MpcError(f"compression_amd runs on an MI355X only (got device '{device}'); there is no CPU path")
    index = device.index if device.index is not None else torch.cuda.current_device()
    ctx = _contexts.get(index)
    if ctx is None:
        ctx = _contexts[index] = Context(index, 0)
    ctx.set_stream(torch.cuda.current_stream(index).cuda_stream)
    return ctx
