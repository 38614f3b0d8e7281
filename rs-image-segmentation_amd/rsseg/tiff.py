"""
Minimal baseline-TIFF reader / writer (host I/O plumbing; no compute).

The reference reads and writes rasters through rasterio/GDAL
(scripts/2_feature_extraction.py:154-168, 239-258; modules/features/extract.py:810-833), neither of
which is available here.  This module covers what the hot path's callers need: uncompressed,
strip-organised TIFFs, chunky or planar (PlanarConfiguration 1 or 2), uint8/uint16/int16/float32/
float64 samples — the layout of the bundled scene data/raw/AA.tif (600x600x7 uint8, planar, one
row per strip) — and writing a (bands, H, W) array back in the same band-sequential layout, with the
georeferencing the reference hands to rasterio (`transform`, `crs`, `nodata`: scripts/2:244-256,
extract.py:818-830) stored as GeoTIFF tags: ModelPixelScale + ModelTiepoint (or ModelTransformation for a
rotated grid), a GeoKey directory naming the EPSG code, and GDAL's NODATA tag.  Tiling and LZW are not written
(files stay uncompressed strips, which GDAL / rasterio read back as the same raster).
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


def _open(path: str):
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
    return buf, bo, _read_ifd(buf, bo, off)


def read_tiff_georef(path: str) -> Dict[str, object]:
    """{'transform': (a, b, c, d, e, f) or None, 'epsg': int or None, 'nodata': float or None} — the affine transform
    in rasterio's order (x = a*col + b*row + c, y = d*col + e*row + f)."""
    _, _, t = _open(path)
    transform = None
    if 34264 in t:
        m = t[34264]
        transform = (m[0], m[1], m[3], m[4], m[5], m[7])
    elif 33550 in t and 33922 in t:
        sx, sy = t[33550][0], t[33550][1]
        i, j, _, x, y, _ = t[33922][:6]
        transform = (sx, 0.0, x - i * sx, 0.0, -sy, y + j * sy)
    epsg = None
    if 34735 in t:
        g = t[34735]
        for k in range(g[3]):
            key, loc, _, val = g[4 + 4 * k:8 + 4 * k]
            if key in (2048, 3072) and loc == 0 and val != 32767:
                epsg = int(val)
    nodata = None
    if 42113 in t:
        txt = t[42113][0].split(b"\0")[0].decode("ascii", "replace").strip()
        try:
            nodata = float(txt)
        except ValueError:
            nodata = None
    return {"transform": transform, "epsg": epsg, "nodata": nodata}


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


def write_tiff(path: str, arr: np.ndarray, transform=None, epsg=None, nodata=None) -> None:
    """Writes (bands, H, W) or (H, W) as an uncompressed little-endian planar TIFF, one strip per band.
    transform: affine (a, b, c, d, e, f) in rasterio's order (or an object with those attributes); epsg: integer
    code of the coordinate reference system (4000-4999: geographic, otherwise projected); nodata: number."""
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
    if 8 + B * band_bytes >= 2 ** 32 - 65536:
        raise ValueError("raster too large for classic TIFF")
    tags = [(256, 4, [W]), (257, 4, [H]), (258, 3, [bits] * B), (259, 3, [1]), (262, 3, [1]),
            (273, 4, [8 + i * band_bytes for i in range(B)]), (277, 3, [B]), (278, 4, [H]), (279, 4, [band_bytes] * B),
            (284, 3, [2]), (339, 3, [fmt] * B), (338, 3, [0] * max(B - 1, 1))]
    if transform is not None:
        if hasattr(transform, "a"):
            transform = (transform.a, transform.b, transform.c, transform.d, transform.e, transform.f)
        ta, tb, tc, td, te, tf = [float(v) for v in transform]
        if tb == 0.0 and td == 0.0:
            tags.append((33550, 12, [ta, -te, 0.0]))
            tags.append((33922, 12, [0.0, 0.0, 0.0, tc, tf, 0.0]))
        else:
            tags.append((34264, 12, [ta, tb, 0.0, tc, td, te, 0.0, tf, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]))
    if epsg is not None:
        epsg = int(epsg)
        geographic = 4000 <= epsg < 5000
        tags.append((34735, 3, [1, 1, 0, 3, 1024, 0, 1, 2 if geographic else 1, 1025, 0, 1, 1,
                               2048 if geographic else 3072, 0, 1, epsg]))
    if nodata is not None:
        txt = (repr(int(nodata)) if float(nodata).is_integer() else repr(float(nodata))).encode("ascii") + b"\0"
        tags.append((42113, 2, txt))
    tags.sort(key=lambda t: t[0])
    ifd_off = 8 + B * band_bytes
    extra_off = ifd_off + 2 + 12 * len(tags) + 4
    entries, extra = [], b""
    for tag, typ, vals in tags:
        cnt = len(vals)
        data = bytes(vals) if typ == 2 else struct.pack("<" + _TYPE_FMT[typ] * cnt, *vals)
        if len(data) <= 4:
            entries.append(struct.pack("<HHI4s", tag, typ, cnt, data.ljust(4, b"\0")))
        else:
            if (extra_off + len(extra)) & 1:
                extra += b"\0"
            entries.append(struct.pack("<HHII", tag, typ, cnt, extra_off + len(extra)))
            extra += data
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, ifd_off))
        f.write(a.tobytes())
        f.write(struct.pack("<H", len(entries)) + b"".join(entries) + struct.pack("<I", 0) + extra)
