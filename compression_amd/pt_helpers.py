"""Host-side helpers for the prior-index side string (mirror of the parts of
src/common/libs/pt_helpers.py:32-107 and src/common/libs/utilities.py:357-390 that the codec uses).

The side string is a few hundred bytes per image (one index per 16x16 pixels); its candidates (raw,
LZMA, PNG, fixed-length) are host compressors in the reference and stay on the host here.
"""
import lzma
import math
import struct
import zlib

import numpy as np
import torch


def to_smallest_type(tensor, integers=False):
    """uint8 if the values fit, else int16 (pt_helpers.py:32-50)."""
    if not integers:
        raise NotImplementedError("to_smallest_type with integers=False")
    mn, mx = int(tensor.min()), int(tensor.max())
    if mn >= 0 and mx <= 255:
        return tensor.to(torch.uint8)
    if mn >= 0 and mx <= 32767:
        return tensor.to(torch.int16)
    raise NotImplementedError(f"to_smallest_type: min={mn}, max={mx}")


def bits_per_value(tensor):
    """Fixed-length code size of the largest value (pt_helpers.py:52-84)."""
    mn, mx = int(tensor.min()), int(tensor.max())
    if mn >= 0 and mx <= 0:
        return 0
    if mn >= 0:
        return math.floor(math.log2(mx) + 1)
    raise NotImplementedError(f"bits_per_value w/ min={mn}, max={mx}")


def png_bytes(arr: np.ndarray) -> bytes:
    """Greyscale PNG of a [1,H,W] (or [H,W]) integer array: bit depth ceil(log2(max+1)) rounded up to a legal
    PNG depth, filter type 0 on every row, zlib level 9 -- the settings utilities.compress_png passes to
    pypng (utilities.py:376-390).  pypng itself is not available here, so byte-for-byte equality with it
    is not verified."""
    a = np.asarray(arr)
    if a.ndim == 3:
        if a.shape[0] != 1:
            raise ValueError("greyscale only")
        a = a[0]
    need = max(1, int(np.ceil(np.log2(int(a.max()) + 1)))) if a.max() > 0 else 1
    depth = next(d for d in (1, 2, 4, 8, 16) if d >= need)
    H, W = a.shape
    if depth == 16:
        rows = a.astype(">u2").tobytes()
        stride = 2 * W
    elif depth == 8:
        rows = a.astype(np.uint8).tobytes()
        stride = W
    else:
        per = 8 // depth
        pad = (-W) % per
        ap = np.pad(a.astype(np.uint8), ((0, 0), (0, pad)))
        ap = ap.reshape(H, -1, per)
        shifts = np.arange(per - 1, -1, -1, dtype=np.uint8) * depth
        rows = (ap << shifts).sum(axis=2).astype(np.uint8).tobytes()
        stride = (W + pad) // per
    raw = b"".join(b"\x00" + rows[i * stride:(i + 1) * stride] for i in range(H))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, 0, 0, 0, 0))
    if depth != need:
        out += chunk(b"sBIT", bytes([need]))
    return out + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")


def png_read_grey(data: bytes) -> np.ndarray:
    """Inverse of png_bytes for the files this package writes (greyscale, filter 0, non-interlaced)."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    pos, idat, W = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            W, H, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if ctype != 0 or interlace != 0:
                raise NotImplementedError("only non-interlaced greyscale PNGs")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    stride = (W * depth + 7) // 8
    out = np.zeros((H, W), np.int64)
    for y in range(H):
        line = raw[y * (stride + 1):(y + 1) * (stride + 1)]
        if line[0] != 0:
            raise NotImplementedError("PNG row filter != 0")
        b = np.frombuffer(line[1:], np.uint8)
        if depth == 16:
            out[y] = b.view(">u2")[:W]
        elif depth == 8:
            out[y] = b[:W]
        else:
            per = 8 // depth
            shifts = np.arange(per - 1, -1, -1, dtype=np.uint8) * depth
            out[y] = ((b[:, None] >> shifts) & ((1 << depth) - 1)).reshape(-1)[:W]
    return out


def compress_png(tensor, outfpath):
    """Greyscale [1,H,W] index map -> PNG file; False if it has more than one channel (utilities.py:376-390)."""
    if tensor.shape[0] > 1:
        return False
    with open(outfpath, "wb") as fp:
        fp.write(png_bytes(np.asarray(tensor)))
    return True


def get_num_bits(tensor, integers=False, compression="lzma", try_png=True):
    """Size in bits of the cheapest coding of an integer map: raw, LZMA, PNG (single-image maps only) or
    fixed-length; 0 for an all-zero map (pt_helpers.py:86-107)."""
    if compression != "lzma":
        raise NotImplementedError(compression)
    tensor = tensor.detach().cpu()
    if tensor.min() == 0 and tensor.max() == 0:
        return 0
    arr = to_smallest_type(tensor, integers=integers).numpy()
    raw = arr.tobytes()
    num_bits = min(len(raw) * 8, len(lzma.compress(raw)) * 8)
    if try_png and arr.ndim == 3 and arr.shape[0] == 1:
        num_bits = min(num_bits, len(png_bytes(arr)) * 8)
    if integers:
        num_bits = min(num_bits, arr.size * bits_per_value(arr))
    return num_bits


# ---- one self-describing file per image ----------------------------------------------------------------
# The reference writes a pair of files (<name>.mbs = torchac bytes, <name>.png = prior-index map,
# manynets_compressor.py:279-311) and leaves "one archive" as a TODO (:278).  MPA1 is that archive:
#   "MPA1" | u16 version | u16 side codec | u32 H | u32 W | u16 h | u16 w | u8 bits per index | u32 side bytes |
#   u32 main bytes | side string | main string (MRS1)
# side codec: 0 all-zero map (no bytes), 1 raw, 2 LZMA, 3 PNG, 4 fixed-length packed bits -- the candidates of
# get_num_bits; the smallest is stored, so the archive's side string costs exactly bpp_sidestring.
ARCHIVE_MAGIC = b"MPA1"
SIDE_ZERO, SIDE_RAW, SIDE_LZMA, SIDE_PNG, SIDE_FIXED = 0, 1, 2, 3, 4
_ARCHIVE_HEAD = struct.Struct("<4sHHIIHHBII")


def _pack_fixed(arr: np.ndarray, nbits: int) -> bytes:
    bits = ((arr.reshape(-1, 1).astype(np.uint32) >> np.arange(nbits - 1, -1, -1, dtype=np.uint32)) & 1).astype(np.uint8)
    return np.packbits(bits.reshape(-1)).tobytes()


def _unpack_fixed(data: bytes, n: int, nbits: int) -> np.ndarray:
    bits = np.unpackbits(np.frombuffer(data, np.uint8))[:n * nbits].reshape(n, nbits).astype(np.uint32)
    return (bits << np.arange(nbits - 1, -1, -1, dtype=np.uint32)).sum(1)


def encode_side_string(indices, try_png=True):
    """[h, w] integer map -> (codec, bits per index, bytes): the cheapest of get_num_bits' candidates."""
    idx = torch.as_tensor(indices).detach().cpu()
    if idx.dim() != 2:
        raise ValueError("the prior-index map of one image is [h, w]")
    if int(idx.min()) == 0 and int(idx.max()) == 0:
        return SIDE_ZERO, 0, b""
    arr = to_smallest_type(idx, integers=True).numpy()
    nbits = bits_per_value(arr)
    cands = [(SIDE_RAW, arr.tobytes()), (SIDE_LZMA, lzma.compress(arr.tobytes())), (SIDE_FIXED, _pack_fixed(arr, nbits))]
    if try_png:
        cands.append((SIDE_PNG, png_bytes(arr[None])))
    # the packed candidate is charged its exact bit count (like get_num_bits), the others their byte size
    cost = lambda c: arr.size * nbits if c[0] == SIDE_FIXED else 8 * len(c[1])
    codec, data = min(cands, key=cost)
    return codec, nbits, data


def decode_side_string(codec, nbits, data, h, w):
    n = h * w
    if codec == SIDE_ZERO:
        return np.zeros((h, w), np.int64)
    if codec == SIDE_PNG:
        out = png_read_grey(data)
    elif codec == SIDE_FIXED:
        out = _unpack_fixed(data, n, nbits)
    else:
        raw = lzma.decompress(data) if codec == SIDE_LZMA else data
        if codec not in (SIDE_RAW, SIDE_LZMA):
            raise ValueError(f"unknown side-string codec {codec}")
        out = np.frombuffer(raw, np.uint8 if len(raw) == n else np.int16)
    out = np.asarray(out).reshape(h, w).astype(np.int64)
    return out


def write_archive(path, bitstream: bytes, indices, H: int, W: int, try_png=True) -> int:
    """Writes one MPA1 file; returns its size in bytes."""
    idx = torch.as_tensor(indices).detach().cpu()
    h, w = idx.shape
    codec, nbits, side = encode_side_string(idx, try_png)
    blob = _ARCHIVE_HEAD.pack(ARCHIVE_MAGIC, 1, codec, H, W, h, w, nbits, len(side), len(bitstream)) + side + bytes(bitstream)
    with open(path, "wb") as fp:
        fp.write(blob)
    return len(blob)


def read_archive(path):
    """-> (main string, indices int64 [h, w], H, W)."""
    with open(path, "rb") as fp:
        blob = fp.read()
    if len(blob) < _ARCHIVE_HEAD.size:
        raise ValueError("not an MPA1 archive: too short")
    magic, version, codec, H, W, h, w, nbits, nside, nmain = _ARCHIVE_HEAD.unpack_from(blob)
    if magic != ARCHIVE_MAGIC or version != 1:
        raise ValueError("not an MPA1 archive (bad magic or version)")
    o = _ARCHIVE_HEAD.size
    if len(blob) != o + nside + nmain:
        raise ValueError("MPA1 archive is truncated")
    side, main = blob[o:o + nside], blob[o + nside:]
    return main, torch.from_numpy(decode_side_string(codec, nbits, side, h, w)), H, W
