"""
Host arithmetic that turns exact order statistics (K1, on the GPU) into NumPy's percentile / median
values.  Only scalars are touched here: the virtual index, its floor, and the linear interpolation
between two neighbouring order statistics — restated from numpy/lib/_function_base_impl.py
(_quantile :4740-4860, _get_indexes, _get_gamma :4630-4637, _lerp :4639-4661) so that dtypes and
rounding follow NumPy 2.x exactly.
"""
from __future__ import annotations

import functools
from typing import List, Sequence, Tuple

import numpy as np


def _virtual_indexes(n: int, q, a_dtype, scalar_q: bool):
    """np.percentile(a, q): q is divided by a.dtype.type(100) (a float array), so a Python scalar q
    becomes an a.dtype 0-d array (virtual index computed in float32 for float32 data) while a
    tuple/array q promotes to float64 (np.nanpercentile(col, (25.0, 75.0)) in RobustScaler)."""
    a_dtype = np.dtype(a_dtype)
    if scalar_q:
        qq = np.asanyarray(np.true_divide(q, a_dtype.type(100)))
    else:
        qq = np.asanyarray(np.true_divide(np.asarray(q), a_dtype.type(100)))
    virtual = np.asanyarray((n - 1) * qq)
    prev = np.asanyarray(np.floor(virtual))
    nxt = np.asanyarray(prev + 1)
    above = virtual >= n - 1
    prev = np.where(above, -1, prev)
    nxt = np.where(above, -1, nxt)
    below = virtual < 0
    prev = np.where(below, 0, prev)
    nxt = np.where(below, 0, nxt)
    prev_i = prev.astype(np.intp)
    next_i = nxt.astype(np.intp)
    gamma = np.asanyarray(virtual - prev_i)
    gamma = np.asanyarray(gamma, dtype=virtual.dtype)
    return prev_i, next_i, gamma


def _lerp(a, b, t):
    diff = np.subtract(b, a)
    out = np.asanyarray(np.add(a, diff * t))
    np.subtract(b, diff * (1 - t), out=out, where=t >= 0.5, casting="unsafe", dtype=type(out.dtype))
    return out


@functools.lru_cache(maxsize=256)   # a plan depends on (n, q) only; a step asks for the same few again and again
def percentile_plan(n: int, q, a_dtype=np.float32, scalar_q: bool = True):
    """Returns (ranks, finish): `ranks` are the 0-based order statistics to fetch; finish(values) maps
    the fetched values (same order, a_dtype) to the percentile value(s)."""
    prev_i, next_i, gamma = _virtual_indexes(n, q, a_dtype, scalar_q)
    pi = np.atleast_1d(prev_i) % n
    ni = np.atleast_1d(next_i) % n
    ranks = [int(x) for x in np.concatenate([pi, ni])]
    m = pi.size

    def finish(values: np.ndarray):
        v = np.asarray(values, dtype=a_dtype)
        if v.ndim == 2:
            # several planes at once, one row of fetched values each: the same elementwise operations on (P,) / (P, m)
            # arrays as on the scalars of one plane (rows holding a NaN come back as NaN)
            P = v.shape[0]
            prev_v = v[:, :m].reshape((P,) + np.shape(gamma))
            next_v = v[:, m:].reshape((P,) + np.shape(gamma))
            with np.errstate(invalid="ignore"):
                res = np.asarray(_lerp(prev_v, next_v, gamma))
            bad = np.isnan(v).any(axis=1)
            if bad.any():
                res = res.copy()
                res[bad] = np.nan
            return res
        if np.isnan(v).any():
            res = np.full(np.shape(gamma), np.nan, dtype=np.result_type(a_dtype, gamma.dtype))
            return res[()] if res.ndim == 0 else res
        prev_v = v[:m].reshape(np.shape(gamma))
        next_v = v[m:].reshape(np.shape(gamma))
        res = _lerp(prev_v, next_v, gamma)
        return res[()] if res.ndim == 0 else res

    # for callers that finish many planes at once with few NumPy calls (lerp_rows): the fetched values are [prev | next]
    finish.m, finish.gamma = m, gamma
    return ranks, finish


def lerp_rows(plan_finish, values: np.ndarray) -> np.ndarray:
    """plan_finish(values) for a (P, 2m) float32 array WITHOUT NaNs (one row per plane), in five NumPy calls: which of _lerp's
    two branches a column takes depends on gamma alone, which the plan knows.  Same operations on the same dtypes as
    _lerp, hence the same values (tests/test_host.py::test_lerp_rows_equals_plan_finish)."""
    m, g = plan_finish.m, plan_finish.gamma
    a, b = values[:, :m], values[:, m:]
    if g.ndim == 0:
        a, b = a[:, 0], b[:, 0]
    diff = b - a
    high = g >= 0.5
    if not high.any():
        return a + diff * g
    if high.all():
        return b - diff * (1 - g)
    return np.where(high, b - diff * (1 - g), a + diff * g)


@functools.lru_cache(maxsize=256)
def median_plan(n: int, a_dtype=np.float32):
    """np.median / np.nanmedian of n valid values: mean of the middle one or two order statistics in
    the array dtype (numpy _median: mean(part[indexer]))."""
    if n % 2 == 1:
        ranks = [(n - 1) // 2]
    else:
        ranks = [n // 2 - 1, n // 2]

    def finish(values: np.ndarray):
        v = np.asarray(values, dtype=a_dtype)
        return np.mean(v, axis=1) if v.ndim == 2 else np.mean(v)   # rows = planes

    return ranks, finish


def band_percentiles(ctx, plane, qs: Sequence[float], n_global: int = None) -> List[np.floating]:
    """np.percentile(band, q) for each scalar q in qs, with ONE order-statistics call (3 passes over
    the plane).  `n_global` is the pixel count over all ranks when the raster is sharded."""
    n = int(plane.numel()) if n_global is None else int(n_global)
    plans = [percentile_plan(n, q, np.float32, True) for q in qs]
    ranks: List[int] = []
    for r, _ in plans:
        ranks.extend(r)
    vals, n_nan = ctx.order_stats(plane, ranks)
    out = []
    o = 0
    for r, fin in plans:
        v = vals[o:o + len(r)]
        o += len(r)
        out.append(np.float32(np.nan) if n_nan > 0 else fin(v))
    return out


def robust_scaler_stats(ctx, plane, n_global: int = None) -> Tuple[np.float32, np.float64]:
    """RobustScaler.fit for one column (sklearn/preprocessing/_data.py:1656-1677): center_ =
    np.nanmedian (float32), scale_ = nanpercentile 75 - nanpercentile 25 (float64), zero -> 1."""
    n = int(plane.numel()) if n_global is None else int(n_global)
    mr, mfin = median_plan(n, np.float32)
    pr, pfin = percentile_plan(n, (25.0, 75.0), np.float32, False)
    vals, n_nan = ctx.order_stats(plane, mr + pr)
    if n_nan > 0:
        nv = n - n_nan  # nan-aware variants ignore NaNs: redo the plan on the valid count
        if nv <= 0:
            return np.float32(np.nan), np.float64(np.nan)
        mr, mfin = median_plan(nv, np.float32)
        pr, pfin = percentile_plan(nv, (25.0, 75.0), np.float32, False)
        vals, _ = ctx.order_stats(plane, mr + pr)
    center = np.float32(mfin(vals[:len(mr)]))
    qv = pfin(vals[len(mr):])
    scale = np.float64(qv[1] - qv[0])
    if scale < 10 * np.finfo(np.float64).eps:
        scale = np.float64(1.0)
    return center, scale
