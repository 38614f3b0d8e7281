"""
Device-resident stage functions: the call order of run_feature_extraction_stage
(reference scripts/2_feature_extraction.py:27-133) and of the KMeans / random-forest branches of
run_classification_stage (scripts/3_classification.py:381-394, 459-480), expressed over planar device
buffers.  The NumPy-in / NumPy-out mirrors in modules/ call these; bench.py calls them directly with
the bands already resident in HBM.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from .quantiles import band_percentiles, robust_scaler_stats
from .runtime import Context

INDEX_NAMES = ["ndvi", "evi", "msavi", "ndwi", "mndwi", "ndbi", "bsi"]
GLCM_NAMES = ["contrast", "dissimilarity", "homogeneity", "energy", "correlation"]


def band_lohi(ctx: Context, bands: Sequence, n_global: Optional[int] = None) -> np.ndarray:
    """(lo, hi) = np.percentile(band, 2 / 98) per band — robust_normalize, indices.py:38-39."""
    out = np.zeros((len(bands), 2), np.float32)
    for i, b in enumerate(bands):
        lo, hi = band_percentiles(ctx, b, (2, 98), n_global)
        out[i] = (lo, hi)
    return out


def band_quantile_bundles(ctx: Context, bands: Sequence, n_global: Optional[int] = None) -> List[dict]:
    """Everything the stage needs from the order statistics of each band, with ONE grouped select (3 passes over each
    plane, 3 host synchronisations / all-reduces for the whole group):
      lo, hi          np.percentile(band, 2 / 98)                      (robust_normalize, indices.py:38-39)
      center, scale   RobustScaler statistics of the NORMALISED band  (indices.py:230-231)
      lo2, hi2        np.percentile(normalised band, 2 / 98)         (the texture functions re-normalise, :265)
    The normalisation f(v) = (clip(v, lo, hi) - lo) / (hi - lo + 1e-10) is monotone non-decreasing in float32
    arithmetic, so the k-th smallest normalised value is f(k-th smallest raw value): the quantiles of the
    normalised band follow from the raw band's order statistics pushed through f on the host (same float32
    operations as the K2 kernel).  A band that holds NaNs falls back to separate selects."""
    from .quantiles import lerp_rows, median_plan, percentile_plan
    n = int(bands[0].numel()) if n_global is None else int(n_global)
    p2r, p2f = percentile_plan(n, 2, np.float32, True)
    p98r, p98f = percentile_plan(n, 98, np.float32, True)
    mr, mf = median_plan(n, np.float32)
    qr, qf = percentile_plan(n, (25.0, 75.0), np.float32, False)
    ranks = p2r + p98r + mr + qr
    out: List[dict] = []
    cuts = np.cumsum([0, len(p2r), len(p98r), len(mr), len(qr)])
    for g0 in range(0, len(bands), 8):
        group = list(bands[g0:g0 + 8])
        vals_all, nan_all = ctx.order_stats_multi(group, [ranks] * len(group))
        # the host arithmetic of all planes of the group at once (elementwise NumPy operations on (P,) arrays give, value
        # for value, what the same operations give on one plane's scalars): the GPU idles while this runs
        # Planes that hold NaNs take the separate path below; their rows here are computed and discarded.
        with np.errstate(invalid="ignore"):
            lo, hi = lerp_rows(p2f, vals_all[:, cuts[0]:cuts[1]]), lerp_rows(p98f, vals_all[:, cuts[1]:cuts[2]])   # (P,) float32
            den = hi - lo + 1e-10                            # float32, as in robust_normalize
            lo_ = lo[:, None]
            fa = (np.minimum(np.maximum(vals_all, lo_), hi[:, None]) - lo_) / den[:, None]    # f of every fetched value (np.clip's two steps)
            center = mf(fa[:, cuts[2]:cuts[3]]).astype(np.float32)
            q = lerp_rows(qf, fa[:, cuts[3]:cuts[4]])        # (P, 2) float64
            scale = q[:, 1] - q[:, 0]
            scale = np.where(scale < 10 * np.finfo(np.float64).eps, np.float64(1.0), scale)
            lo2, hi2 = lerp_rows(p2f, fa[:, cuts[0]:cuts[1]]), lerp_rows(p98f, fa[:, cuts[1]:cuts[2]])
        for i, (band, n_nan) in enumerate(zip(group, nan_all)):
            if n_nan > 0:
                blo, bhi = band_percentiles(ctx, band, (2, 98), n_global)
                out.append(dict(lo=blo, hi=bhi, center=None, scale=None, lo2=None, hi2=None))
            else:
                out.append(dict(lo=lo[i], hi=hi[i], center=center[i], scale=scale[i], lo2=lo2[i], hi2=hi2[i]))
    return out


def band_quantile_bundle(ctx: Context, band, n_global: Optional[int] = None):
    """band_quantile_bundles for one band."""
    return band_quantile_bundles(ctx, [band], n_global)[0]


def normalize_bands(ctx: Context, bands: Sequence, lohi: np.ndarray) -> List:
    return [ctx.normalize(b, float(lohi[i, 0]), float(lohi[i, 1])) for i, b in enumerate(bands)]


def spectral_indices(ctx: Context, bands: Sequence, lohi: Optional[np.ndarray], want_norm=(False,) * 5):
    """7 index planes from bands[0:5] (blue, green, red, nir, swir1); fused with the normalisation
    when lohi (5x2) is given."""
    outs, norms = ctx.spectral_indices(list(bands[:5]), None if lohi is None else np.asarray(lohi)[:5], want_norm)
    return dict(zip(INDEX_NAMES, outs)), norms


def pca(ctx: Context, norm_bands: Sequence, n_components: Optional[int] = None, use_robust_scaling: bool = True,
        n_global: Optional[int] = None, stats=None, lohi=None):
    """perform_pca (indices.py:205-246) on normalised band planes.  `stats` (optional): precomputed
    [(center, scale)] per band from band_quantile_bundle.  `lohi` (optional, with `stats`): the planes are the RAW bands
    and are robust-normalised with these percentiles inside the PCA kernels (no normalised planes in memory)."""
    nb = len(norm_bands)
    nc = nb if n_components is None else n_components
    if lohi is not None and use_robust_scaling and stats is None:
        raise ValueError("pca: raw bands (lohi) need the RobustScaler statistics of the normalised bands (stats)")
    if use_robust_scaling:
        if stats is None:
            stats = [robust_scaler_stats(ctx, b, n_global) for b in norm_bands]
        center = np.array([s[0] for s in stats], np.float32)
        scale = np.array([s[1] for s in stats], np.float64)
    else:
        center = scale = None
    outs, comp, ratio, mean, ev = ctx.pca_fit_transform(list(norm_bands), center, scale, nc, lohi)
    return outs, ratio, dict(components=comp, mean=mean, explained_variance=ev, center=center, scale=scale)


def indices_and_pca(ctx: Context, bands: Sequence, qb: Sequence[dict], lohi: np.ndarray, n_components: Optional[int], want_norm=(False,) * 5,
                    fit: Optional[Tuple[int, int]] = None, quantize=None):
    """The seven indices and perform_pca of the same raw bands through the fused entry point (one Gram pass + ONE pass that
    writes indices, wanted normalised bands and components): (idx dict, norms, pcs, ratio, model)."""
    nc = len(bands) if n_components is None else n_components
    center = np.array([q["center"] for q in qb], np.float32)
    scale = np.array([q["scale"] for q in qb], np.float64)
    idx, norms, pcs, comp, ratio, mean, ev = ctx.indices_pca(list(bands), lohi, center, scale, nc, want_norm=want_norm, fit=fit, quantize=quantize)
    return dict(zip(INDEX_NAMES, idx)), norms, pcs, ratio, dict(components=comp, mean=mean, explained_variance=ev, center=center, scale=scale)


def renormalize(ctx: Context, plane, n_global: Optional[int] = None):
    """The texture functions re-apply robust_normalize to the band they receive (indices.py:265, 333,
    412, 455, 531)."""
    lo, hi = band_percentiles(ctx, plane, (2, 98), n_global)
    return ctx.normalize(plane, float(lo), float(hi))


def glcm_features(ctx: Context, nir_norm, H: int, W: int, levels=32, window_size=21, step_size=21, upsample=True, renorm=None, q=None):
    """calculate_glcm_features (indices.py:248-318) on an already re-normalised band, or — renorm=(lo, hi) — on the band
    as the function receives it, re-normalised with its percentiles and quantised in one pass, or — q — on the already
    quantised uint8 plane (the fused index / PCA pass can write it)."""
    if q is not None:
        pass
    elif renorm is None:
        q = ctx.quantize_u8(nir_norm, float(levels - 1))
    else:
        q = ctx.normalize_quantize_u8(nir_norm, float(renorm[0]), float(renorm[1]), float(levels - 1))
    small, (oh, ow) = ctx.glcm(q, H, W, levels, window_size, step_size)
    if not upsample:
        return dict(zip(GLCM_NAMES, small)), (oh, ow)
    return dict(zip(GLCM_NAMES, ctx.resize_bilinear_multi(small, oh, ow, 0, oh, H, W, 0, H))), (oh, ow)   # five maps, one launch


def _async_pipeline(fn):
    """Entry points whose results stay on the device do not wait for the stream inside the pipeline (rsseg_ctx_set_async);
    the calls that hand values to the host wait by themselves."""
    import functools

    @functools.wraps(fn)
    def wrapped(ctx, *args, **kwargs):
        was_async = getattr(ctx, "_async", False)
        if not was_async:
            ctx.set_async(True)
        try:
            return fn(ctx, *args, **kwargs)
        finally:
            if not was_async:
                ctx.end_async()
    return wrapped


@_async_pipeline
def feature_stack19(ctx: Context, bands: Sequence, H: int, W: int, glcm_window=21, glcm_step=21, glcm_levels=32,
                    n_global: Optional[int] = None, preprocessing: bool = True):
    """The 19 planes of hierarchical_features['all'] (scripts/2:112-127), in stack order.
    Returns (planes, dtypes_note): all planes float32 except index 16 (gradient_5) which is uint8 on
    the device and becomes uint8/255.0 (float64) on the host, as in indices.py:440.
    preprocessing=False (scripts/2:39-47): the bands are used as given — indices and PCA on the caller's values; the
    texture functions still robust-normalise the band they receive (indices.py:265, 412, 455, 531)."""
    if not preprocessing:
        idx, _ = spectral_indices(ctx, bands, None)
        norm_all = list(bands)
        pcs, ratio, model = pca(ctx, norm_all, None, True, n_global, None)
        nir2 = renormalize(ctx, bands[3], n_global)
        lohi, fused, qb = None, False, None
    else:
        qb = band_quantile_bundles(ctx, bands, n_global)
        lohi = np.array([[q["lo"], q["hi"]] for q in qb], np.float32)
        fused = all(q["center"] is not None for q in qb)
        if fused:
            # one grouped select serves every percentile; the PCA normalises the RAW bands inside its kernels (no
            # normalised planes are written, no range pass), only the normalised NIR band is kept for the texture chain
            idx, norms, pcs, ratio, model = indices_and_pca(ctx, bands, qb, lohi, None, want_norm=tuple(i == 3 for i in range(5)))
            nir2 = ctx.normalize(norms[3], float(qb[3]["lo2"]), float(qb[3]["hi2"]), out=norms[3])
            norm_all = None
        else:  # a band with NaNs: separate selects on the normalised planes
            idx, norms = spectral_indices(ctx, bands, lohi, want_norm=(True,) * 5)
            norm_all = list(norms) + [ctx.normalize(bands[i], float(lohi[i, 0]), float(lohi[i, 1])) for i in range(5, len(bands))]
            pcs, ratio, model = pca(ctx, norm_all, None, True, n_global, None)
            nir2 = renormalize(ctx, norm_all[3], n_global)
    level1 = [idx["ndwi"], idx["mndwi"], idx["ndvi"], idx["evi"], idx["ndbi"], idx["bsi"], pcs[0]]
    ctx_planes = ctx.box_mean_multi(level1, H, W, 7, L.BORDER_REFLECT)   # the 7 channels of add_spatial_context in one launch
    glcm, _ = glcm_features(ctx, nir2, H, W, glcm_levels, glcm_window, glcm_step)
    q255 = ctx.quantize_u8(nir2, 255.0)
    grad = ctx.morph_gradient(q255, H, W, 5)
    std5 = ctx.local_std(nir2, H, W, 5)
    sob = ctx.sobel_mag(q255, H, W)
    planes = level1 + ctx_planes + [glcm["contrast"], glcm["homogeneity"], grad, std5, sob]
    extras = dict(indices=idx, pca=pcs, pca_ratio=ratio, pca_model=model, glcm=glcm, lohi=lohi, nir2=nir2, q255=q255)
    return planes, extras


def stack19_forest_planes(ctx: Context, planes: Sequence) -> List:
    """The float32 matrix RandomForestClassifier.predict sees for the 19-feature stack: float32 planes as
    they are, the uint8 morphological gradient as float32(uint8 / 255.0)."""
    import torch
    return [ctx.u8_to_unit(p) if p.dtype == torch.uint8 else p for p in planes]


def stack19_to_host(planes: Sequence, H: int, W: int) -> np.ndarray:
    """(H, W, 19) float64, C order — the layout of all_hierarchical_features.npy (scripts/2:211)."""
    out = np.empty((H, W, len(planes)), np.float64)
    for i, p in enumerate(planes):
        a = p.cpu().numpy().reshape(H, W)
        out[:, :, i] = a / 255.0 if a.dtype == np.uint8 else a
    return out


def _with_minmax(fn):
    """The config pipelines let the producers of the feature planes (index / upsample / projection kernels) reduce
    each plane's minimum and maximum while they write it, so that KMeans' MinMaxScaler needs no pass of its own."""
    import functools

    @functools.wraps(fn)
    def wrapped(ctx, *args, **kwargs):
        ctx.collect_minmax(True)
        # entry points whose results stay on the device do not wait for the stream (rsseg_ctx_set_async): the calls that
        # hand values to the host (order statistics, extrema, PCA sums) wait by themselves, and KMeans ends with a wait
        was_async = getattr(ctx, "_async", False)
        if not was_async:
            ctx.set_async(True)
        try:
            return fn(ctx, *args, **kwargs)
        finally:
            ctx.collect_minmax(False)
            if not was_async:
                ctx.end_async()
    return wrapped


@_with_minmax
def config2(ctx: Context, bands: Sequence, k: int = 6):
    """BASELINE config 2: percentile normalisation + 7 spectral indices + KMeans(k)."""
    lohi = band_lohi(ctx, bands[:5])
    idx, _ = spectral_indices(ctx, bands, lohi)
    planes = [idx[n] for n in INDEX_NAMES]
    labels, meta = ctx.kmeans_fit_predict(planes, k)
    return labels, meta, planes


@_with_minmax
def config3(ctx: Context, bands: Sequence, H: int, W: int, k: int = 8, glcm_window=7, glcm_step=1, n_pca=3,
            n_global: Optional[int] = None, overlap: bool = False, qb: Optional[List[dict]] = None, glcm: Optional[dict] = None):
    """BASELINE config 3: 7 indices + 5 GLCM properties (window 7, 4 angles) + PCA(3) -> 15 float32
    features -> KMeans(k).  One select per band serves all percentile requests (band_quantile_bundle).
    overlap=True enqueues the GLCM chain (quantise -> windows -> 5 bilinear upsamples) on a second HIP
    stream beside the selects / indices / PCA of the other bands and joins before KMeans.  Measured on MI355X
    (profiles/r01_overlap_note.md): the GLCM kernel fills every CU and slows down by what the other stream
    executes (30.8 -> 44.6 ms), so the critical path does not shorten; it is off by default.
    qb: the bands' quantile bundles when the caller has them already (band_quantile_bundle per band, e.g. computed while the
    next band was still crossing PCIe); default: one grouped select here.
    glcm: the five upsampled texture planes when the caller has them already (texture_planes: they depend on the NIR band
    alone, so a caller that uploads NIR first can compute them while the other bands are still in flight)."""
    NIR = 3
    if not overlap:
        # one grouped select, then: indices (+ the normalised NIR band only), texture chain on the normalised NIR band
        # re-normalised and quantised in one pass, PCA straight from the RAW bands (normalisation inside its kernels)
        if qb is None:
            qb = band_quantile_bundles(ctx, bands, n_global)
        lohi = np.array([[q["lo"], q["hi"]] for q in qb], np.float32)
        fused = all(q["center"] is not None for q in qb)
        want = tuple(i == NIR for i in range(5)) if fused else (True,) * 5
        if fused:   # indices + PCA straight from the RAW bands: one Gram pass, one pass that writes the indices, the components and the
            # quantised texture band (the normalised NIR band itself is not needed by anything else in this configuration)
            lo2, hi2 = qb[NIR]["lo2"], qb[NIR]["hi2"]
            if glcm is None:
                idx, norms, pcs, ratio, model = indices_and_pca(ctx, bands, qb, lohi, n_pca, quantize=(lo2, hi2, 31.0))
                glcm, _ = glcm_features(ctx, None, H, W, 32, glcm_window, glcm_step, q=ctx.last_quantized)
            else:
                idx, norms, pcs, ratio, model = indices_and_pca(ctx, bands, qb, lohi, n_pca)
        else:
            idx, norms = spectral_indices(ctx, bands, lohi, want_norm=want)
            lo2, hi2 = band_percentiles(ctx, norms[NIR], (2, 98), n_global)
            glcm, _ = glcm_features(ctx, norms[NIR], H, W, 32, glcm_window, glcm_step, renorm=(lo2, hi2))
        if fused:
            pass
        else:  # a band with NaNs: separate selects on the normalised planes
            norm_all = list(norms) + [ctx.normalize(bands[i], float(lohi[i, 0]), float(lohi[i, 1])) for i in range(5, len(bands))]
            pcs, ratio, model = pca(ctx, norm_all, n_pca, True, n_global, None)
            del norm_all
        del norms
        planes = [idx[n] for n in INDEX_NAMES] + [glcm[n] for n in GLCM_NAMES] + list(pcs)
        labels, meta = ctx.kmeans_fit_predict(planes, k)
        return labels, meta, planes
    # overlap: the NIR band's statistics first, so that its texture chain can start beside the rest
    qn = band_quantile_bundle(ctx, bands[NIR], n_global)
    fused = qn["center"] is not None
    nir_norm = ctx.normalize(bands[NIR], float(qn["lo"]), float(qn["hi"]))
    if fused:
        lo2, hi2 = qn["lo2"], qn["hi2"]
    else:
        lo2, hi2 = band_percentiles(ctx, nir_norm, (2, 98), n_global)
    nir2 = ctx.normalize(nir_norm, float(lo2), float(hi2))
    g = ctx.aux()
    g.torch_stream.wait_stream(ctx.torch_stream)
    glcm, _ = glcm_features(g, nir2, H, W, 32, glcm_window, glcm_step)   # asynchronous on the aux stream
    rest = band_quantile_bundles(ctx, [b for i, b in enumerate(bands) if i != NIR], n_global)
    qb = rest[:NIR] + [qn] + rest[NIR:]
    lohi = np.array([[q["lo"], q["hi"]] for q in qb], np.float32)
    idx, norms = spectral_indices(ctx, bands, lohi, want_norm=(True,) * 5)
    norm_all = list(norms) + [ctx.normalize(bands[i], float(lohi[i, 0]), float(lohi[i, 1])) for i in range(5, len(bands))]
    stats = [(q["center"], q["scale"]) for q in qb] if all(q["center"] is not None for q in qb) else None
    pcs, ratio, model = pca(ctx, norm_all, n_pca, True, n_global, stats)
    del norm_all, norms
    g.sync()
    planes = [idx[n] for n in INDEX_NAMES] + [glcm[n] for n in GLCM_NAMES] + list(pcs)
    labels, meta = ctx.kmeans_fit_predict(planes, k)
    return labels, meta, planes


@_with_minmax
def texture_planes(ctx: Context, nir, qn: dict, H: int, W: int, glcm_window=7, glcm_step=1, levels: int = 32) -> dict:
    """The texture chain of config 3 from the NIR band ALONE (its quantile bundle qn = band_quantile_bundle(ctx, nir)):
    robust_normalize, the texture function's re-normalisation + quantisation (indices.py:265-268), the GLCM windows and the
    five bilinear upsamples — the same arithmetic as the fused path of config3 (which lets the index / PCA pass write the
    quantised band), hence the same planes bit for bit; tagged with their extrema for KMeans like config3's own."""
    if qn["center"] is None:
        raise ValueError("texture_planes: the band holds NaNs (use config3 without precomputed planes)")
    x = ctx.widen_u8(nir) if Context._is_u8(nir) else nir
    q = ctx.normalize_quantize_u8(ctx.normalize(x, float(qn["lo"]), float(qn["hi"])), float(qn["lo2"]), float(qn["hi2"]), float(levels - 1))
    planes, _ = glcm_features(ctx, None, H, W, levels, glcm_window, glcm_step, q=q)
    return planes


# ------------------------------------------------------------------------------------------------
# one raster, row stripes over the ranks (BASELINE config 4)
# ------------------------------------------------------------------------------------------------
def stripe_rows(H: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [r0, r1) owned by `rank`: contiguous stripes in rank order.  Every rank owns at least one row (the same check
    fails on every rank, so nobody is left waiting in a collective)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"stripe_rows: rank {rank} outside a world of {world}")
    if H < world:
        raise ValueError(f"stripe_rows: {H} rows cannot be split over {world} ranks (a rank would own no row)")
    return (H * rank) // world, (H * (rank + 1)) // world


def glcm_halo_rows(H: int, r0: int, r1: int, win: int, step: int) -> Tuple[int, int, int, int]:
    """For output rows [r0, r1) of the bilinear upsample of the GLCM maps of an H-row raster: the small-map
    rows [j0, j1] the stripe taps (cv2.resize pixel-centre mapping) and the image rows [i0, i1) those windows
    read.  Returns (j0, j1, i0, i1)."""
    sh = (H - win) // step + 1
    scale = 1.0 / (H / sh)

    def tap(y):
        return int(np.floor(np.float32((y + 0.5) * scale - 0.5)))

    j0 = min(max(tap(r0), 0), sh - 1)
    j1 = min(max(tap(r1 - 1) + 1, 0), sh - 1)
    return j0, j1, j0 * step, j1 * step + win


@_with_minmax
def config3_striped(ctx: Context, bands: Sequence, nir_ext, H: int, W: int, r0: int, r1: int, i0: int, k: int = 8,
                    glcm_window=7, glcm_step=1, n_pca=3):
    """BASELINE config 3/4 on ONE H x W raster sharded by rows: this rank owns rows [r0, r1) (`bands`: its
    stripe of each band) and additionally holds `nir_ext`, the NIR rows [i0, i0 + nir_ext.numel()/W) that its
    texture windows read (glcm_halo_rows).  Percentiles, PCA and KMeans reduce over all ranks through the
    context's all-reduce hook; the label stripe equals rows [r0, r1) of the single-GPU result bit for bit."""
    n_global = H * W
    qb = band_quantile_bundles(ctx, bands, n_global)
    lohi = np.array([[q["lo"], q["hi"]] for q in qb], np.float32)
    fused = all(q["center"] is not None for q in qb)
    if fused:   # indices + PCA straight from the raw stripes; no normalised planes are written
        idx, _, pcs, ratio, model = indices_and_pca(ctx, bands, qb, lohi, n_pca)
        lo2, hi2 = qb[3]["lo2"], qb[3]["hi2"]
    else:
        idx, norms = spectral_indices(ctx, bands, lohi, want_norm=(True,) * 5)
        norm_all = list(norms) + [ctx.normalize(bands[i], float(lohi[i, 0]), float(lohi[i, 1])) for i in range(5, len(bands))]
        pcs, ratio, model = pca(ctx, norm_all, n_pca, True, n_global, None)
        lo2, hi2 = band_percentiles(ctx, norm_all[3], (2, 98), n_global)
        del norm_all, norms
    ext_rows = nir_ext.numel() // W
    nir_n = ctx.normalize(nir_ext, float(lohi[3, 0]), float(lohi[3, 1]))
    q = ctx.normalize_quantize_u8(nir_n, float(lo2), float(hi2), 31.0)
    del nir_n
    j0, j1, need_i0, need_i1 = glcm_halo_rows(H, r0, r1, glcm_window, glcm_step)
    if need_i0 < i0 or need_i1 > i0 + ext_rows or (need_i0 - i0) % glcm_step:
        raise ValueError(f"nir_ext rows [{i0},{i0 + ext_rows}) do not cover the texture rows [{need_i0},{need_i1})")
    skip = need_i0 - i0
    qv = q[skip * W:(skip + need_i1 - need_i0) * W]
    small, (oh, ow) = ctx.glcm(qv, need_i1 - need_i0, W, 32, glcm_window, glcm_step)
    sh = (H - glcm_window) // glcm_step + 1
    glcm = ctx.resize_bilinear_multi(small, oh, ow, j0, sh, r1 - r0, W, r0, H)
    planes = [idx[n] for n in INDEX_NAMES] + glcm + list(pcs)
    labels, meta = ctx.kmeans_fit_predict(planes, k)
    return labels, meta, planes


# ------------------------------------------------------------------------------------------------
# the 19-feature stack of ONE raster sharded by rows (BASELINE config 5)
# ------------------------------------------------------------------------------------------------
CONTEXT_HALO = 3   # 7x7 context mean of the level-1 planes (indices.py:770)
TEXTURE_HALO = 2   # 5x5 blur / morphology (indices.py:433, 537-541); the 3x3 Sobel needs 1


def stack19_halo_rows(H: int, r0: int, r1: int, glcm_window: int = 21, glcm_step: int = 21) -> Tuple[int, int]:
    """Rows [e0, e1) of every band that the rank owning rows [r0, r1) of an H-row raster must hold for stack19_striped:
    its stripe, 3 rows either side for the 7x7 context mean (hence 3 halo rows of the index planes and of PC0's inputs),
    2 for the 5x5 blur / morphology, 1 for Sobel, and the rows of the texture windows its bilinear taps reach
    (glcm_halo_rows) — clipped to the raster, where the operators' own border rules apply."""
    _, _, i0, i1 = glcm_halo_rows(H, r0, r1, glcm_window, glcm_step)
    return max(0, min(r0 - CONTEXT_HALO, i0)), min(H, max(r1 + CONTEXT_HALO, i1))


def _rows_view(t, W: int, a: int, b: int):
    """Rows [a, b) of a planar tensor as a 16-byte aligned device buffer: a view when the offset allows, else a copy."""
    v = t[a * W:b * W]
    return v if v.data_ptr() % 16 == 0 else v.clone()


@_async_pipeline
def stack19_striped(ctx: Context, bands_ext: Sequence, H: int, W: int, r0: int, r1: int, e0: int, glcm_window=21, glcm_step=21,
                    glcm_levels=32):
    """feature_stack19 for the rank that owns rows [r0, r1) of ONE H x W raster.  `bands_ext`: rows [e0, e1) of every
    band, covering stack19_halo_rows(H, r0, r1, ...).  Order statistics, the PCA fit and the Sobel maximum reduce over
    the OWNED rows of all ranks through the context's all-reduce hook; window operators run on stripe + halo with true
    image borders only at the raster's edges.  Returns the 19 planes restricted to rows [r0, r1): bit for bit the rows
    of the single-GPU feature_stack19 (tests/test_gpu_dist.py)."""
    n_global = H * W
    e1 = e0 + bands_ext[0].numel() // W
    need0, need1 = stack19_halo_rows(H, r0, r1, glcm_window, glcm_step)
    if need0 < e0 or need1 > e1 or not (0 <= r0 <= r1 <= H):
        raise ValueError(f"bands_ext rows [{e0},{e1}) do not cover the rows [{need0},{need1}) that stripe [{r0},{r1}) needs")
    own = [_rows_view(b, W, r0 - e0, r1 - e0) for b in bands_ext]
    qb = band_quantile_bundles(ctx, own, n_global)
    lohi = np.array([[q["lo"], q["hi"]] for q in qb], np.float32)
    fused = all(q["center"] is not None for q in qb)
    # level 1 on rows [c0, c1): the stripe and the rows its 7x7 context mean reads
    c0, c1 = max(r0 - CONTEXT_HALO, 0), min(r1 + CONTEXT_HALO, H)
    bc = [_rows_view(b, W, c0 - e0, c1 - e0) for b in bands_ext]
    fit = ((r0 - c0) * W, (r1 - r0) * W)
    if fused:
        idx, _, pcs, ratio, model_ = indices_and_pca(ctx, bc, qb, lohi, len(bc), fit=fit)
        comp, mean, ev, center, scale = model_["components"], model_["mean"], model_["explained_variance"], model_["center"], model_["scale"]
        lo2, hi2 = qb[3]["lo2"], qb[3]["hi2"]
    else:  # a band with NaNs: RobustScaler statistics by separate selects on the normalised stripes
        idx, norms = spectral_indices(ctx, bc, lohi, want_norm=(True,) * 5)
        norm_c = list(norms) + [ctx.normalize(bc[i], float(lohi[i, 0]), float(lohi[i, 1])) for i in range(5, len(bc))]
        own_n = [_rows_view(p, W, r0 - c0, r1 - c0) for p in norm_c]
        stats = [robust_scaler_stats(ctx, p, n_global) for p in own_n]
        center = np.array([s[0] for s in stats], np.float32)
        scale = np.array([s[1] for s in stats], np.float64)
        pcs, comp, ratio, mean, ev = ctx.pca_fit_transform(norm_c, center, scale, len(bc), None, fit=fit)
        lo2, hi2 = band_percentiles(ctx, own_n[3], (2, 98), n_global)
        del norm_c, norms, own_n
    level1_c = [idx["ndwi"], idx["mndwi"], idx["ndvi"], idx["evi"], idx["ndbi"], idx["bsi"], pcs[0]]
    edges_c = (1 if c0 == 0 else 0) | (2 if c1 == H else 0)
    ctx_planes = ctx.box_mean_multi(level1_c, c1 - c0, W, 7, L.BORDER_REFLECT, rows=(r0 - c0, r1 - c0), edges=edges_c)
    level1 = [_rows_view(p, W, r0 - c0, r1 - c0) for p in level1_c]
    # texture chain of the NIR band on rows [t0, t1): GLCM windows + the 5x5 / 3x3 operators' halos
    j0, j1, i0, i1 = glcm_halo_rows(H, r0, r1, glcm_window, glcm_step)
    m0, m1 = max(r0 - TEXTURE_HALO, 0), min(r1 + TEXTURE_HALO, H)
    t0, t1 = min(i0, m0), max(i1, m1)
    nir2 = ctx.normalize(_rows_view(bands_ext[3], W, t0 - e0, t1 - e0), float(lohi[3, 0]), float(lohi[3, 1]))
    nir2 = ctx.normalize(nir2, float(lo2), float(hi2), out=nir2)
    q = ctx.quantize_u8(_rows_view(nir2, W, i0 - t0, i1 - t0), float(glcm_levels - 1))
    small, (oh, ow) = ctx.glcm(q, i1 - i0, W, glcm_levels, glcm_window, glcm_step)
    sh = (H - glcm_window) // glcm_step + 1
    glcm = dict(zip(GLCM_NAMES, ctx.resize_bilinear_multi(small, oh, ow, j0, sh, r1 - r0, W, r0, H)))
    nir_m = _rows_view(nir2, W, m0 - t0, m1 - t0)
    edges_m = (1 if m0 == 0 else 0) | (2 if m1 == H else 0)
    rows_m = (r0 - m0, r1 - m0)
    q255 = ctx.quantize_u8(nir_m, 255.0)
    grad = ctx.morph_gradient(q255, m1 - m0, W, 5, rows=rows_m, edges=edges_m)
    std5 = ctx.local_std(nir_m, m1 - m0, W, 5, rows=rows_m, edges=edges_m)
    sob = ctx.sobel_mag(q255, m1 - m0, W, rows=rows_m, edges=edges_m)
    planes = level1 + ctx_planes + [glcm["contrast"], glcm["homogeneity"], grad, std5, sob]
    extras = dict(indices=idx, pca=pcs, pca_ratio=ratio, pca_model=dict(components=comp, mean=mean, explained_variance=ev, center=center, scale=scale),
                  glcm=glcm, lohi=lohi, rows_level1=(c0, c1))
    return planes, extras
