"""
Minimal TIFF / BigTIFF / GeoTIFF reader and writer (host I/O plumbing; no compute).

The reference reads and writes rasters through rasterio/GDAL (scripts/2_feature_extraction.py:154-168, 239-258;
scripts/3_classification.py:509-538; modules/features/extract.py:810-833), neither of which is available here.
This module covers what the hot path's callers need:

  reading   strips or tiles, chunky or planar (PlanarConfiguration 1 or 2), uncompressed or LZW, classic or BigTIFF,
            uint8 / uint16 / int16 / int32 / float32 / float64 — e.g. the bundled scene data/raw/AA.tif (600x600x7
            uint8, planar, one row per strip);
  writing   a (bands, H, W) array band-sequentially, as uncompressed strips or — the layout the reference asks rasterio
            for — 256 x 256 tiles with LZW (`compress='lzw'`, `tiled=True`), classic TIFF or BigTIFF (chosen
            automatically above 4 GB: the (19, 16384, 16384) float64 stack is 41 GB), one band at a time so that a
            memory-mapped stack never has to be resident;
  georef    `transform`, `crs`, `nodata` as GeoTIFF tags: ModelPixelScale + ModelTiepoint (or ModelTransformation for a
            rotated grid), a GeoKey directory naming the EPSG code, GDAL's NODATA tag.

LZW is done by the library's host helpers (rsseg_host_lzw_encode / _decode, include/rsseg.h).
"""
from __future__ import annotations

import ctypes as C
import os
import struct
from typing import Dict, Optional, Tuple

import numpy as np

_TYPE_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8, 17: 8, 18: 8}
_TYPE_FMT = {1: "B", 3: "H", 4: "I", 6: "b", 8: "h", 9: "i", 11: "f", 12: "d", 16: "Q", 17: "q", 18: "Q"}
TILE = 256


# ---- LZW through the C ABI ------------------------------------------------------------------------------------------
def _lzw(fn_name: str, data: bytes, cap: int) -> bytes:
    from . import _lib as L
    fn = getattr(L.load(), fn_name)
    src = np.frombuffer(data, np.uint8)
    dst = np.empty(max(cap, 16), np.uint8)
    n = fn(src.ctypes.data_as(C.c_void_p), src.size, dst.ctypes.data_as(C.c_void_p), dst.size)
    if n < 0:
        raise ValueError(f"{fn_name} failed ({n})")
    return dst[:n].tobytes()


def lzw_encode(data: bytes) -> bytes:
    return _lzw("rsseg_host_lzw_encode", data, len(data) * 2 + 1024)


def lzw_decode(data: bytes, expected: int) -> bytes:
    return _lzw("rsseg_host_lzw_decode", data, expected)


# ---- reading ---------------------------------------------------------------------------------------------------------
def _read_ifd(buf, bo: str, off: int, big: bool) -> Dict[int, Tuple]:
    if big:
        (n,) = struct.unpack_from(bo + "Q", buf, off)
        base, esz, inl = off + 8, 20, 8
    else:
        (n,) = struct.unpack_from(bo + "H", buf, off)
        base, esz, inl = off + 2, 12, 4
    tags = {}
    for i in range(n):
        if big:
            tag, typ, cnt = struct.unpack_from(bo + "HHQ", buf, base + esz * i)
            raw = bytes(buf[base + esz * i + 12:base + esz * i + 20])
        else:
            tag, typ, cnt = struct.unpack_from(bo + "HHI", buf, base + esz * i)
            raw = bytes(buf[base + esz * i + 8:base + esz * i + 12])
        size = _TYPE_SIZES.get(typ, 1) * cnt
        if size <= inl:
            data = raw[:size]
        else:
            (p,) = struct.unpack(bo + ("Q" if big else "I"), raw)
            data = bytes(buf[p:p + size])
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
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    head = bytes(buf[:16])
    if head[:2] == b"II":
        bo = "<"
    elif head[:2] == b"MM":
        bo = ">"
    else:
        raise ValueError(f"{path}: not a TIFF file")
    (magic,) = struct.unpack_from(bo + "H", head, 2)
    if magic == 42:
        (off,) = struct.unpack_from(bo + "I", head, 4)
        big = False
    elif magic == 43:
        osz, _, off = struct.unpack_from(bo + "HHQ", head, 4)
        if osz != 8:
            raise ValueError(f"{path}: BigTIFF with {osz}-byte offsets")
        big = True
    else:
        raise ValueError(f"{path}: unknown TIFF magic {magic}")
    return buf, bo, _read_ifd(buf, bo, off, big)


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
    buf, bo, t = _open(path)
    W, H = t[256][0], t[257][0]
    spp = t.get(277, (1,))[0]
    bits = t.get(258, (1,))[0]
    comp = t.get(259, (1,))[0]
    planar = t.get(284, (1,))[0]
    fmt = t.get(339, (1,))[0]
    pred = t.get(317, (1,))[0]
    if comp not in (1, 5):
        raise ValueError(f"{path}: compression {comp} not supported (uncompressed and LZW only)")
    if pred != 1:
        raise ValueError(f"{path}: predictor {pred} not supported")
    key = (fmt, bits)
    dt = {(1, 8): "u1", (1, 16): "u2", (2, 16): "i2", (1, 32): "u4", (2, 32): "i4", (3, 32): "f4", (3, 64): "f8"}.get(key)
    if dt is None:
        raise ValueError(f"{path}: sample format {key} not supported")
    dt = np.dtype(bo + dt)
    ch = 1 if planar == 2 else spp            # samples interleaved in one strip / tile
    planes = spp if planar == 2 else 1
    out = np.empty((planes, H, W, ch), dt)
    tiled = 322 in t
    if tiled:
        tw, th = t[322][0], t[323][0]
        offs, cnts = t[324], t[325]
        tx, ty = (W + tw - 1) // tw, (H + th - 1) // th
        if len(offs) != tx * ty * planes:
            raise ValueError(f"{path}: unexpected tile count {len(offs)}")
        for p in range(planes):
            for j in range(ty):
                for i in range(tx):
                    k = (p * ty + j) * tx + i
                    raw = bytes(buf[offs[k]:offs[k] + cnts[k]])
                    if comp == 5:
                        raw = lzw_decode(raw, tw * th * ch * dt.itemsize)
                    tile = np.frombuffer(raw, dt, tw * th * ch).reshape(th, tw, ch)
                    y0, x0 = j * th, i * tw
                    out[p, y0:y0 + th, x0:x0 + tw] = tile[:min(th, H - y0), :min(tw, W - x0)]
    else:
        offs, cnts = t[273], t[279]
        rps = min(t.get(278, (H,))[0], H)
        spb = (H + rps - 1) // rps
        if len(offs) != spb * planes:
            raise ValueError(f"{path}: unexpected strip count {len(offs)}")
        for p in range(planes):
            for s in range(spb):
                k = p * spb + s
                rows = min(rps, H - s * rps)
                raw = bytes(buf[offs[k]:offs[k] + cnts[k]])
                if comp == 5:
                    raw = lzw_decode(raw, rows * W * ch * dt.itemsize)
                out[p, s * rps:s * rps + rows] = np.frombuffer(raw, dt, rows * W * ch).reshape(rows, W, ch)
    res = out[:, :, :, 0] if planar == 2 else np.moveaxis(out[0], -1, 0)
    return np.ascontiguousarray(res).astype(dt.newbyteorder("="))


# ---- writing ---------------------------------------------------------------------------------------------------------
def write_tiff(path: str, arr, transform=None, epsg=None, nodata=None, compress: Optional[str] = None, tiled: Optional[bool] = None,
               bigtiff: Optional[bool] = None, geographic: Optional[bool] = None) -> None:
    """Writes (bands, H, W) or (H, W) — an ndarray or anything sliceable per band (np.memmap) — as a little-endian,
    band-sequential (PlanarConfiguration 2) TIFF.
      compress   None: one uncompressed strip per band; 'lzw': LZW (the reference's rasterio call)
      tiled      256 x 256 tiles (default: when compressed, as the reference writes); otherwise strips
      bigtiff    None: BigTIFF only when the file would not fit the 4 GB offsets of classic TIFF
      transform  affine (a, b, c, d, e, f) in rasterio's order (or an object with those attributes)
      epsg       EPSG code of the coordinate reference system; `geographic` says whether it names a geographic (True) or a
                 projected (False) system — when None, codes 4000-4999 are taken as geographic (EPSG's own block for
                 geographic 2-D systems; pass the flag for the projected codes that live in that block, e.g. 4087)
      nodata     number (GDAL_NODATA tag)"""
    a = arr if hasattr(arr, "shape") else np.asarray(arr)
    if len(a.shape) == 2:
        a = a[None]
    if len(a.shape) != 3:
        raise ValueError("write_tiff expects (bands, H, W) or (H, W)")
    kinds = {"u1": (1, 8), "u2": (1, 16), "i2": (2, 16), "i4": (2, 32), "f4": (3, 32), "f8": (3, 64)}
    k = np.dtype(a.dtype).str[1:]
    if k not in kinds:
        raise ValueError(f"dtype {a.dtype} not supported")
    if compress not in (None, "none", "lzw"):
        raise ValueError(f"compress={compress!r}: only None and 'lzw'")
    lzw = compress == "lzw"
    if tiled is None:
        tiled = lzw
    fmt, bits = kinds[k]
    ledt = np.dtype("<" + k)
    B, H, W = a.shape
    isz = ledt.itemsize
    raw_total = B * H * W * isz
    if bigtiff is None:
        # LZW can GROW incompressible data (float64 noise: 9..12-bit codes for 8-bit symbols, up to ~1.4x), so with
        # compression the choice is made on a worst-case bound; a classic file that would pass 4 GB is never started
        worst = raw_total * 3 // 2 if lzw else raw_total
        bigtiff = worst + (1 << 20) >= 2 ** 32 - (1 << 16)
    tx, ty = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    rows_per_strip = H if not lzw else max(1, min(H, (1 << 20) // max(W * isz, 1)))   # ~1 MB strips when compressed
    nstrips = (H + rows_per_strip - 1) // rows_per_strip
    offs, cnts = [], []
    try:
        _write_body(path, a, bigtiff, tiled, lzw, ledt, tx, ty, rows_per_strip, nstrips, offs, cnts, bits, fmt, transform, epsg, geographic, nodata)
    except BaseException:
        if os.path.exists(path):    # never leave a multi-GB partial file behind
            os.remove(path)
        raise


def _write_body(path, a, bigtiff, tiled, lzw, ledt, tx, ty, rows_per_strip, nstrips, offs, cnts, bits, fmt, transform, epsg, geographic, nodata):
    B, H, W = a.shape
    with open(path, "wb") as f:
        f.write(b"\0" * (16 if bigtiff else 8))           # header, patched at the end
        for b in range(B):
            band = np.ascontiguousarray(np.asarray(a[b]), dtype=ledt)
            if tiled:
                for j in range(ty):
                    for i in range(tx):
                        tile = np.zeros((TILE, TILE), ledt)
                        blk = band[j * TILE:(j + 1) * TILE, i * TILE:(i + 1) * TILE]
                        tile[:blk.shape[0], :blk.shape[1]] = blk
                        data = tile.tobytes()
                        if lzw:
                            data = lzw_encode(data)
                        offs.append(f.tell())
                        cnts.append(len(data))
                        if not bigtiff and offs[-1] + len(data) >= 2 ** 32 - (1 << 16):
                            raise ValueError("raster too large for classic TIFF (pass bigtiff=True)")
                        f.write(data)
                        if len(data) & 1:
                            f.write(b"\0")
            else:
                for s in range(nstrips):
                    data = band[s * rows_per_strip:(s + 1) * rows_per_strip].tobytes()
                    if lzw:
                        data = lzw_encode(data)
                    offs.append(f.tell())
                    cnts.append(len(data))
                    if not bigtiff and offs[-1] + len(data) >= 2 ** 32 - (1 << 16):
                        raise ValueError("raster too large for classic TIFF (pass bigtiff=True)")
                    f.write(data)
                    if len(data) & 1:
                        f.write(b"\0")
            del band
        LONG = 16 if bigtiff else 4
        tags = [(256, 4, [W]), (257, 4, [H]), (258, 3, [bits] * B), (259, 3, [5 if lzw else 1]), (262, 3, [1]), (277, 3, [B]),
                (284, 3, [2]), (339, 3, [fmt] * B)]
        if B > 1:
            tags.append((338, 3, [0] * (B - 1)))   # ExtraSamples: every band after the first is unspecified data (none for one band)
        if tiled:
            tags += [(322, 4, [TILE]), (323, 4, [TILE]), (324, LONG, offs), (325, LONG, cnts)]
        else:
            tags += [(273, LONG, offs), (278, 4, [rows_per_strip]), (279, LONG, cnts)]
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
            geo = (4000 <= epsg < 5000) if geographic is None else bool(geographic)
            tags.append((34735, 3, [1, 1, 0, 3, 1024, 0, 1, 2 if geo else 1, 1025, 0, 1, 1, 2048 if geo else 3072, 0, 1, epsg]))
        if nodata is not None:
            txt = (repr(int(nodata)) if float(nodata).is_integer() else repr(float(nodata))).encode("ascii") + b"\0"
            tags.append((42113, 2, txt))
        tags.sort(key=lambda t: t[0])
        if f.tell() & 1:
            f.write(b"\0")
        ifd_off = f.tell()
        if not bigtiff and ifd_off >= 2 ** 32 - (1 << 16):
            raise ValueError("raster too large for classic TIFF (pass bigtiff=True)")
        esz, inl = (20, 8) if bigtiff else (12, 4)
        extra_off = ifd_off + (8 if bigtiff else 2) + esz * len(tags) + (8 if bigtiff else 4)
        entries, extra = [], b""
        for tag, typ, vals in tags:
            cnt = len(vals)
            data = bytes(vals) if typ == 2 else struct.pack("<" + _TYPE_FMT[typ] * cnt, *vals)
            head = struct.pack("<HHQ" if bigtiff else "<HHI", tag, typ, cnt)
            if len(data) <= inl:
                entries.append(head + data.ljust(inl, b"\0"))
            else:
                if (extra_off + len(extra)) & 1:
                    extra += b"\0"
                entries.append(head + struct.pack("<Q" if bigtiff else "<I", extra_off + len(extra)))
                extra += data
        if bigtiff:
            f.write(struct.pack("<Q", len(entries)) + b"".join(entries) + struct.pack("<Q", 0) + extra)
            f.seek(0)
            f.write(struct.pack("<2sHHHQ", b"II", 43, 8, 0, ifd_off))
        else:
            f.write(struct.pack("<H", len(entries)) + b"".join(entries) + struct.pack("<I", 0) + extra)
            f.seek(0)
            f.write(struct.pack("<2sHI", b"II", 42, ifd_off))
