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


def normalize_bands(ctx: Context, bands: Sequence, lohi: np.ndarray) -> List:
    return [ctx.normalize(b, float(lohi[i, 0]), float(lohi[i, 1])) for i, b in enumerate(bands)]


def spectral_indices(ctx: Context, bands: Sequence, lohi: Optional[np.ndarray], want_norm=(False,) * 5):
    """7 index planes from bands[0:5] (blue, green, red, nir, swir1); fused with the normalisation
    when lohi (5x2) is given."""
    outs, norms = ctx.spectral_indices(list(bands[:5]), None if lohi is None else np.asarray(lohi)[:5], want_norm)
    return dict(zip(INDEX_NAMES, outs)), norms


def pca(ctx: Context, norm_bands: Sequence, n_components: Optional[int] = None, use_robust_scaling: bool = True,
        n_global: Optional[int] = None):
    """perform_pca (indices.py:205-246) on normalised band planes."""
    nb = len(norm_bands)
    nc = nb if n_components is None else n_components
    if use_robust_scaling:
        stats = [robust_scaler_stats(ctx, b, n_global) for b in norm_bands]
        center = np.array([s[0] for s in stats], np.float32)
        scale = np.array([s[1] for s in stats], np.float64)
    else:
        center = scale = None
    outs, comp, ratio, mean, ev = ctx.pca_fit_transform(list(norm_bands), center, scale, nc)
    return outs, ratio, dict(components=comp, mean=mean, explained_variance=ev, center=center, scale=scale)


def renormalize(ctx: Context, plane, n_global: Optional[int] = None):
    """The texture functions re-apply robust_normalize to the band they receive (indices.py:265, 333,
    412, 455, 531)."""
    lo, hi = band_percentiles(ctx, plane, (2, 98), n_global)
    return ctx.normalize(plane, float(lo), float(hi))


def glcm_features(ctx: Context, nir_norm, H: int, W: int, levels=32, window_size=21, step_size=21, upsample=True):
    """calculate_glcm_features (indices.py:248-318) on an already re-normalised band."""
    q = ctx.quantize_u8(nir_norm, float(levels - 1))
    small, (oh, ow) = ctx.glcm(q, H, W, levels, window_size, step_size)
    if not upsample:
        return dict(zip(GLCM_NAMES, small)), (oh, ow)
    return {k: ctx.resize_bilinear(v, oh, ow, H, W) for k, v in zip(GLCM_NAMES, small)}, (oh, ow)


def feature_stack19(ctx: Context, bands: Sequence, H: int, W: int, glcm_window=21, glcm_step=21, glcm_levels=32):
    """The 19 planes of hierarchical_features['all'] (scripts/2:112-127), in stack order.
    Returns (planes, dtypes_note): all planes float32 except index 16 (gradient_5) which is uint8 on
    the device and becomes uint8/255.0 (float64) on the host, as in indices.py:440."""
    lohi = band_lohi(ctx, bands)
    idx, norms = spectral_indices(ctx, bands, lohi, want_norm=(True,) * 5)
    norm_all = list(norms) + [ctx.normalize(bands[i], float(lohi[i, 0]), float(lohi[i, 1])) for i in range(5, len(bands))]
    pcs, ratio, model = pca(ctx, norm_all, None, True)
    level1 = [idx["ndwi"], idx["mndwi"], idx["ndvi"], idx["evi"], idx["ndbi"], idx["bsi"], pcs[0]]
    ctx_planes = [ctx.box_mean(p, H, W, 7, L.BORDER_REFLECT) for p in level1]
    nir2 = renormalize(ctx, norm_all[3])
    glcm, _ = glcm_features(ctx, nir2, H, W, glcm_levels, glcm_window, glcm_step)
    q255 = ctx.quantize_u8(nir2, 255.0)
    grad = ctx.morph_gradient(q255, H, W, 5)
    std5 = ctx.local_std(nir2, H, W, 5)
    sob = ctx.sobel_mag(q255, H, W)
    planes = level1 + ctx_planes + [glcm["contrast"], glcm["homogeneity"], grad, std5, sob]
    extras = dict(indices=idx, norm=norm_all, pca=pcs, pca_ratio=ratio, pca_model=model, glcm=glcm, lohi=lohi)
    return planes, extras


def stack19_to_host(planes: Sequence, H: int, W: int) -> np.ndarray:
    """(H, W, 19) float64, C order — the layout of all_hierarchical_features.npy (scripts/2:211)."""
    out = np.empty((H, W, len(planes)), np.float64)
    for i, p in enumerate(planes):
        a = p.cpu().numpy().reshape(H, W)
        out[:, :, i] = a / 255.0 if a.dtype == np.uint8 else a
    return out


def config2(ctx: Context, bands: Sequence, k: int = 6):
    """BASELINE config 2: percentile normalisation + 7 spectral indices + KMeans(k)."""
    lohi = band_lohi(ctx, bands[:5])
    idx, _ = spectral_indices(ctx, bands, lohi)
    planes = [idx[n] for n in INDEX_NAMES]
    labels, meta = ctx.kmeans_fit_predict(planes, k)
    return labels, meta, planes


def config3(ctx: Context, bands: Sequence, H: int, W: int, k: int = 8, glcm_window=7, glcm_step=1, n_pca=3):
    """BASELINE config 3: 7 indices + 5 GLCM properties (window 7, 4 angles) + PCA(3) -> 15 float32
    features -> KMeans(k)."""
    lohi = band_lohi(ctx, bands)
    idx, norms = spectral_indices(ctx, bands, lohi, want_norm=(True,) * 5)
    norm_all = list(norms) + [ctx.normalize(bands[i], float(lohi[i, 0]), float(lohi[i, 1])) for i in range(5, len(bands))]
    pcs, ratio, model = pca(ctx, norm_all, n_pca, True)
    nir2 = renormalize(ctx, norm_all[3])
    del norm_all
    glcm, _ = glcm_features(ctx, nir2, H, W, 32, glcm_window, glcm_step)
    planes = [idx[n] for n in INDEX_NAMES] + [glcm[n] for n in GLCM_NAMES] + list(pcs)
    labels, meta = ctx.kmeans_fit_predict(planes, k)
    return labels, meta, planes
