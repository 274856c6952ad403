"""ctypes front-end of the C oracle coders (range_coder.c, rans_ref.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_coders.so")
_lib = None


def build() -> str:
    """Compile liboracle_coders.so with gcc (needs only the two C files next to this one)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        u16p, i16p, i32p, u8p, u32p = (ctypes.c_void_p,) * 5
        L.oracle_range_encode_rows.restype = ctypes.c_long
        L.oracle_range_encode_rows.argtypes = [u16p, ctypes.c_int, i16p, ctypes.c_size_t, u8p, ctypes.c_size_t]
        L.oracle_range_decode_rows.restype = ctypes.c_int
        L.oracle_range_decode_rows.argtypes = [u16p, ctypes.c_int, u8p, ctypes.c_size_t, i16p, ctypes.c_size_t]
        L.oracle_range_encode_table.restype = ctypes.c_long
        L.oracle_range_encode_table.argtypes = [u16p, ctypes.c_int, ctypes.c_int, i32p, i16p, ctypes.c_size_t,
                                                u8p, ctypes.c_size_t]
        L.oracle_range_decode_table.restype = ctypes.c_int
        L.oracle_range_decode_table.argtypes = [u16p, ctypes.c_int, ctypes.c_int, i32p, u8p, ctypes.c_size_t,
                                                i16p, ctypes.c_size_t]
        L.oracle_mrs1_header_bytes.restype = ctypes.c_size_t
        L.oracle_mrs1_header_bytes.argtypes = [ctypes.c_uint32]
        L.oracle_mrs1_encode.restype = ctypes.c_long
        L.oracle_mrs1_encode.argtypes = [u16p, ctypes.c_int, ctypes.c_int, i32p, i16p, ctypes.c_uint32,
                                         ctypes.c_uint32, ctypes.c_uint32, u8p, ctypes.c_size_t]
        L.oracle_mrs1_decode.restype = ctypes.c_int
        L.oracle_mrs1_decode.argtypes = [u16p, ctypes.c_int, ctypes.c_int, i32p, u8p, ctypes.c_size_t, i16p,
                                         ctypes.c_uint32]
        L.oracle_symbol_freqs.restype = ctypes.c_int
        L.oracle_symbol_freqs.argtypes = [u16p, ctypes.c_int, ctypes.c_int, i32p, i16p, ctypes.c_uint32, u32p]
        _lib = L
    return _lib


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data


def range_encode_rows(rows: np.ndarray, sym: np.ndarray) -> bytes:
    """rows: int16/uint16 [n_sym, Lp] (what the reference hands to torchac.encode_cdf); sym int16 [n_sym]."""
    rows, rp = _c(rows.view(np.uint16) if rows.dtype == np.int16 else rows, np.uint16)
    sym, sp = _c(sym, np.int16)
    cap = sym.size * 4 + 64
    out = np.empty(cap, np.uint8)
    n = lib().oracle_range_encode_rows(rp, rows.shape[-1], sp, sym.size, out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError(f"range encode failed: {n}")
    return out[:n].tobytes()


def range_decode_rows(rows: np.ndarray, data: bytes) -> np.ndarray:
    rows, rp = _c(rows.view(np.uint16) if rows.dtype == np.int16 else rows, np.uint16)
    n_sym = rows.size // rows.shape[-1]
    buf = np.frombuffer(data, np.uint8)
    out = np.empty(n_sym, np.int16)
    rc = lib().oracle_range_decode_rows(rp, rows.shape[-1], buf.ctypes.data, buf.size, out.ctypes.data, n_sym)
    if rc:
        raise RuntimeError(f"range decode failed: {rc}")
    return out


def _table_args(table, idx):
    table = table.view(np.uint16) if table.dtype == np.int16 else table
    table, tp = _c(table, np.uint16)
    idx, ip = _c(np.asarray(idx).reshape(-1), np.int32)
    K, M, Lp = table.shape
    return table, tp, idx, ip, M, Lp


def range_encode_table(table: np.ndarray, idx: np.ndarray, sym: np.ndarray) -> bytes:
    """table int16 [K,M,Lp], idx [n_loc], sym int16 [n_loc*M] -> bytes, without materialising rows."""
    table, tp, idx, ip, M, Lp = _table_args(table, idx)
    sym, sp = _c(sym, np.int16)
    cap = sym.size * 4 + 64
    out = np.empty(cap, np.uint8)
    n = lib().oracle_range_encode_table(tp, M, Lp, ip, sp, sym.size, out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError(f"range encode failed: {n}")
    return out[:n].tobytes()


def range_decode_table(table: np.ndarray, idx: np.ndarray, data: bytes) -> np.ndarray:
    table, tp, idx, ip, M, Lp = _table_args(table, idx)
    n_sym = idx.size * M
    buf = np.frombuffer(data, np.uint8)
    out = np.empty(n_sym, np.int16)
    rc = lib().oracle_range_decode_table(tp, M, Lp, ip, buf.ctypes.data, buf.size, out.ctypes.data, n_sym)
    if rc:
        raise RuntimeError(f"range decode failed: {rc}")
    return out


def mrs1_encode(table: np.ndarray, idx: np.ndarray, sym: np.ndarray, n_streams: int, sps: int) -> bytes:
    table, tp, idx, ip, M, Lp = _table_args(table, idx)
    sym, sp = _c(sym, np.int16)
    cap = int(lib().oracle_mrs1_header_bytes(n_streams)) + 2 * sym.size + 4 * n_streams + 16
    out = np.empty(cap, np.uint8)
    n = lib().oracle_mrs1_encode(tp, M, Lp, ip, sp, sym.size, n_streams, sps, out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError(f"mrs1 encode failed: {n}")
    return out[:n].tobytes()


def mrs1_decode(table: np.ndarray, idx: np.ndarray, data: bytes) -> np.ndarray:
    table, tp, idx, ip, M, Lp = _table_args(table, idx)
    n_sym = idx.size * M
    buf = np.frombuffer(data, np.uint8)
    out = np.empty(n_sym, np.int16)
    rc = lib().oracle_mrs1_decode(tp, M, Lp, ip, buf.ctypes.data, buf.size, out.ctypes.data, n_sym)
    if rc:
        raise RuntimeError(f"mrs1 decode failed: {rc}")
    return out


def ideal_bits(table: np.ndarray, idx: np.ndarray, sym: np.ndarray) -> float:
    """sum_i -log2(freq_i / 2^16): the information-theoretic size under the int16 tables."""
    table, tp, idx, ip, M, Lp = _table_args(table, idx)
    sym, sp = _c(sym, np.int16)
    f = np.empty(sym.size, np.uint32)
    rc = lib().oracle_symbol_freqs(tp, M, Lp, ip, sp, sym.size, f.ctypes.data)
    if rc:
        raise RuntimeError(f"symbol_freqs failed: {rc}")
    return float(np.sum(16.0 - np.log2(f.astype(np.float64))))
