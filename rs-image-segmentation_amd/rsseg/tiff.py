"""
Minimal baseline-TIFF reader / writer (host I/O plumbing; no compute).

The reference reads and writes rasters through rasterio/GDAL
(scripts/2_feature_extraction.py:154-168, 239-258; modules/features/extract.py:810-833), neither of
which is available here.  This module covers what the hot path's callers need: uncompressed,
strip-organised TIFFs, chunky or planar (PlanarConfiguration 1 or 2), uint8/uint16/int16/float32/
float64 samples — the layout of the bundled scene data/raw/AA.tif (600x600x7 uint8, planar, one
row per strip) — and writing a (bands, H, W) array back in the same band-sequential layout.
GeoTIFF tags, tiling and LZW are out of scope for this round.
"""
from __future__ import annotations

import struct
from typing import Dict, Tuple

import numpy as np

_TYPE_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8}
_TYPE_FMT = {1: "B", 3: "H", 4: "I", 6: "b", 8: "h", 9: "i", 11: "f", 12: "d", 16: "Q"}


def _read_ifd(buf: bytes, bo: str, off: int) -> Dict[int, Tuple]:
    (n,) = struct.unpack_from(bo + "H", buf, off)
    tags = {}
    for i in range(n):
        tag, typ, cnt, raw = struct.unpack_from(bo + "HHI4s", buf, off + 2 + 12 * i)
        size = _TYPE_SIZES.get(typ, 1) * cnt
        if size <= 4:
            data = raw[:size]
        else:
            (p,) = struct.unpack(bo + "I", raw)
            data = buf[p:p + size]
        if typ in _TYPE_FMT:
            vals = struct.unpack(bo + _TYPE_FMT[typ] * cnt, data)
        elif typ == 5:
            v = struct.unpack(bo + "I" * (2 * cnt), data)
            vals = tuple(v[2 * j] / max(v[2 * j + 1], 1) for j in range(cnt))
        else:
            vals = (data,)
        tags[tag] = vals
    return tags


def read_tiff(path: str) -> np.ndarray:
    """Returns (bands, H, W) in the file's sample dtype."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:2] == b"II":
        bo = "<"
    elif buf[:2] == b"MM":
        bo = ">"
    else:
        raise ValueError(f"{path}: not a TIFF file")
    magic, off = struct.unpack_from(bo + "HI", buf, 2)
    if magic != 42:
        raise ValueError(f"{path}: BigTIFF / unknown magic {magic} not supported")
    t = _read_ifd(buf, bo, off)
    W, H = t[256][0], t[257][0]
    spp = t.get(277, (1,))[0]
    bits = t.get(258, (1,))[0]
    comp = t.get(259, (1,))[0]
    planar = t.get(284, (1,))[0]
    fmt = t.get(339, (1,))[0]
    if comp != 1:
        raise ValueError(f"{path}: compression {comp} not supported (uncompressed only)")
    if 322 in t:
        raise ValueError(f"{path}: tiled TIFF not supported")
    key = (fmt, bits)
    dt = {(1, 8): "u1", (1, 16): "u2", (2, 16): "i2", (1, 32): "u4", (2, 32): "i4",
          (3, 32): "f4", (3, 64): "f8"}.get(key)
    if dt is None:
        raise ValueError(f"{path}: sample format {key} not supported")
    dt = np.dtype(bo + dt)
    offs, cnts = t[273], t[279]
    rps = min(t.get(278, (H,))[0], H)
    data = b"".join(buf[o:o + c] for o, c in zip(offs, cnts))
    arr = np.frombuffer(data, dtype=dt)
    if planar == 2:
        spb = (H + rps - 1) // rps
        if len(offs) != spb * spp:
            raise ValueError(f"{path}: unexpected strip count {len(offs)}")
        out = arr[:spp * H * W].reshape(spp, H, W)
    else:
        out = arr[:H * W * spp].reshape(H, W, spp).transpose(2, 0, 1)
    return np.ascontiguousarray(out).astype(dt.newbyteorder("="))


def write_tiff(path: str, arr: np.ndarray) -> None:
    """Writes (bands, H, W) or (H, W) as an uncompressed little-endian planar TIFF, one strip per band."""
    a = np.asarray(arr)
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3:
        raise ValueError("write_tiff expects (bands, H, W) or (H, W)")
    kinds = {"u1": (1, 8), "u2": (1, 16), "i2": (2, 16), "i4": (2, 32), "f4": (3, 32), "f8": (3, 64)}
    k = a.dtype.str[1:]
    if k not in kinds:
        raise ValueError(f"dtype {a.dtype} not supported")
    fmt, bits = kinds[k]
    a = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<")))
    B, H, W = a.shape
    band_bytes = H * W * a.dtype.itemsize
    if 8 + B * band_bytes >= 2 ** 32 - 4096:
        raise ValueError("raster too large for classic TIFF")
    entries = []
    extra = b""
    n_tags = 12
    ifd_off = 8 + B * band_bytes
    extra_off = ifd_off + 2 + 12 * n_tags + 4

    def add(tag, typ, vals):
        nonlocal extra
        cnt = len(vals)
        data = struct.pack("<" + _TYPE_FMT[typ] * cnt, *vals)
        if len(data) <= 4:
            entries.append(struct.pack("<HHI4s", tag, typ, cnt, data.ljust(4, b"\0")))
        else:
            entries.append(struct.pack("<HHII", tag, typ, cnt, extra_off + len(extra)))
            extra += data

    add(256, 4, [W]); add(257, 4, [H]); add(258, 3, [bits] * B); add(259, 3, [1])
    add(262, 3, [1]); add(273, 4, [8 + i * band_bytes for i in range(B)]); add(277, 3, [B])
    add(278, 4, [H]); add(279, 4, [band_bytes] * B); add(284, 3, [2]); add(339, 3, [fmt] * B)
    add(338, 3, [0] * max(B - 1, 1))
    entries.sort(key=lambda e: struct.unpack("<H", e[:2])[0])
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, ifd_off))
        f.write(a.tobytes())
        f.write(struct.pack("<H", len(entries)) + b"".join(entries) + struct.pack("<I", 0) + extra)
