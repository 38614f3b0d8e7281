"""
Drop-in counterpart of the reference's modules/features/indices.py for the hot path: same function
names, positional order, defaults, NumPy-in / NumPy-out contract and dtypes, with
the arithmetic done by the gfx950 kernels of librsseg_hip.so.  Inputs are never modified; outputs are
fresh host arrays.  There is no CPU fallback: without the HIP library / an MI355X these raise.

Functions of the reference module that are plotting (visualize_*) or never called by the stages (Gabor, HOG, ...)
are out of scope; every member of the stage-2 feature dictionary is produced.
"""
from __future__ import annotations

import os

import numpy as np

from rsseg import _lib as _L
from rsseg import pipeline as _P
from rsseg.quantiles import band_percentiles as _band_percentiles
from rsseg.runtime import default_context as _ctx

__all__ = [
    "robust_normalize", "calculate_ndvi", "calculate_evi", "calculate_msavi", "calculate_ndwi", "calculate_mndwi",
    "calculate_ndbi", "calculate_bsi", "perform_pca", "calculate_glcm_features", "calculate_lbp_features", "calculate_morphological_features",
    "calculate_multi_scale_features", "calculate_filter_responses", "add_spatial_context", "prepare_level_1_features",
    "prepare_level_2_features", "visualize_hierarchical_features", "visualize_selected_features", "np", "os",
]


def _dev(a):
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("expected a 2-D band array")
    if a.dtype.kind == "f" and a.dtype.itemsize > 4:
        # the reference would carry a float64 band through in float64 (NumPy promotion); the kernels of these functions compute
        # in float32 — the dtype scripts/2_feature_extraction.py:156 gives every band.  Narrowing silently would return other
        # values under the reference's name, so it is refused by name instead.
        from rsseg.runtime import RssegUnsupported
        raise RssegUnsupported(f"{a.dtype} band: these functions compute in float32 (what scripts/2 passes: src.read(i).astype(np.float32)); "
                               "cast the band with .astype(np.float32) to get the float32 result")
    return _ctx().to_device(np.ascontiguousarray(a, dtype=np.float32).reshape(-1)), a.shape


def _host(t, shape):
    return t.cpu().numpy().reshape(shape)


def robust_normalize(band, lower_percentile=2, upper_percentile=98):
    """reference indices.py:25-48"""
    d, shape = _dev(band)
    lo, hi = _band_percentiles(_ctx(), d, (lower_percentile, upper_percentile))
    return _host(_ctx().normalize(d, float(lo), float(hi)), shape)


def _indices(nir=None, red=None, blue=None, green=None, swir=None, which=0, evi_coef=None):
    ref = next(x for x in (nir, red, blue, green, swir) if x is not None)
    shape = np.asarray(ref).shape
    zeros = None
    planes = []
    for b in (blue, green, red, nir, swir):
        if b is None:
            if zeros is None:
                zeros = _ctx().to_device(np.zeros(int(np.prod(shape)), np.float32))
            planes.append(zeros)
        else:
            planes.append(_dev(b)[0])
    want = [i == which for i in range(7)]
    outs, _ = _ctx().spectral_indices(planes, None, want=want, evi_coef=evi_coef)
    return _host(outs[which], shape)


def calculate_ndvi(nir_band, red_band):  # indices.py:50-71
    return _indices(nir=nir_band, red=red_band, which=0)


def calculate_evi(nir_band, red_band, blue_band, L=1, C1=6, C2=7.5, G=2.5):  # indices.py:73-95
    return _indices(nir=nir_band, red=red_band, blue=blue_band, which=1, evi_coef=(L, C1, C2, G))


def calculate_msavi(nir_band, red_band):  # indices.py:97-114
    return _indices(nir=nir_band, red=red_band, which=2)


def calculate_ndwi(green_band, nir_band):  # indices.py:116-137
    return _indices(green=green_band, nir=nir_band, which=3)


def calculate_mndwi(green_band, swir_band):  # indices.py:139-158
    return _indices(green=green_band, swir=swir_band, which=4)


def calculate_ndbi(swir_band, nir_band):  # indices.py:160-179
    return _indices(swir=swir_band, nir=nir_band, which=5)


def calculate_bsi(blue_band, red_band, nir_band, swir_band):  # indices.py:181-203
    return _indices(blue=blue_band, red=red_band, nir=nir_band, swir=swir_band, which=6)


class _PCAModel:
    """Fitted-PCA stand-in when scikit-learn is not importable (attributes as sklearn names them)."""

    def transform(self, X):
        X = np.asarray(X, dtype=np.float32)
        return X @ self.components_.T - (self.mean_.reshape(1, -1) @ self.components_.T)


def perform_pca(bands_data, n_components=None, use_robust_scaling=True):
    """reference indices.py:205-246.  Returns (list of (H,W) float32 components, explained_variance_ratio_,
    fitted model exposing components_/mean_/explained_variance_/explained_variance_ratio_/transform)."""
    if len(bands_data) < 1:
        raise ValueError("perform_pca: bands_data is empty")
    shape = np.asarray(bands_data[0]).shape
    planes = [_dev(b)[0] for b in bands_data]
    ctx = _ctx()
    n = int(np.prod(shape))
    nans = 0
    if use_robust_scaling:
        from rsseg.quantiles import robust_scaler_stats
        stats = []
        for p in planes:   # RobustScaler ignores NaNs (nanmedian / nanpercentile); PCA then refuses them, as sklearn does
            stats.append(robust_scaler_stats(ctx, p))
        outs, ratio, m = _P.pca(ctx, planes, n_components, True, stats=stats)
    else:
        # (X - min) / (max - min + 1e-10) in float32 (indices.py:232-234): the arithmetic of robust_normalize with the
        # extrema in place of the percentiles, applied inside the PCA kernels
        lohi = np.zeros((len(planes), 2), np.float32)
        for i, p in enumerate(planes):
            vals, nn = ctx.order_stats(p, [0, max(n - 1, 0)])
            nans += nn
            lohi[i] = vals
        if nans:
            raise ValueError("Input X contains NaN.")
        outs, comp, ratio, mean, ev = ctx.pca_fit_transform(planes, None, None, len(planes) if n_components is None else n_components, lohi)
        m = dict(components=comp, mean=mean, explained_variance=ev, center=None, scale=None)
    result = [_host(o, shape) for o in outs]
    try:
        from sklearn.decomposition import PCA
        model = PCA(n_components=n_components)
    except Exception:  # noqa: BLE001
        model = _PCAModel()
    model.components_ = m["components"]
    model.mean_ = m["mean"]
    model.explained_variance_ = m["explained_variance"]
    model.explained_variance_ratio_ = ratio
    model.n_components_ = len(result)
    model.n_features_in_ = len(bands_data)
    model.n_samples_ = int(np.prod(shape))
    model.whiten = False
    model.robust_center_ = m["center"]
    model.robust_scale_ = m["scale"]
    return result, ratio, model


def calculate_glcm_features(band, distances=[1], angles=[0, np.pi / 4, np.pi / 2, 3 * np.pi / 4], levels=32, window_size=21,
                            step_size=21):
    """reference indices.py:248-318"""
    if list(distances) != [1] or not np.allclose(list(angles), [0, np.pi / 4, np.pi / 2, 3 * np.pi / 4]):
        raise ValueError("calculate_glcm_features: only distances=[1] and the four default angles are implemented")
    d, (h, w) = _dev(band)
    ctx = _ctx()
    nir2 = _P.renormalize(ctx, d)
    feats, _ = _P.glcm_features(ctx, nir2, h, w, levels, window_size, step_size)
    return {k: _host(v, (h, w)) for k, v in feats.items()}


def calculate_lbp_features(band, radius=3, n_points=24):
    """reference indices.py:320-344: uniform LBP codes of the re-normalised uint8 band, divided by their maximum (float64)."""
    d, (h, w) = _dev(band)
    ctx = _ctx()
    q = ctx.quantize_u8(_P.renormalize(ctx, d), 255.0)
    lbp = _host(ctx.lbp_uniform(q, h, w, n_points, radius), (h, w)).astype(np.float64)
    return lbp / lbp.max()


def calculate_morphological_features(band):
    """reference indices.py:401-442 — erosion / dilation / opening / closing / gradient at 3, 5, 7 (float64, u8 / 255.0;
    the stack consumes 'gradient_5')."""
    d, (h, w) = _dev(band)
    ctx = _ctx()
    q = ctx.quantize_u8(_P.renormalize(ctx, d), 255.0)
    ops = (("erosion", _L.MORPH_ERODE), ("dilation", _L.MORPH_DILATE), ("opening", _L.MORPH_OPEN), ("closing", _L.MORPH_CLOSE),
           ("gradient", _L.MORPH_GRADIENT))
    return {f"{name}_{k}": _host(ctx.morph(q, h, w, k, op), (h, w)) / 255.0 for k in (3, 5, 7) for name, op in ops}


def calculate_multi_scale_features(band, scales=[1, 3, 5, 7]):
    """reference indices.py:519-562 — mean_scale_k, variance_scale_k, std_dev_scale_k (the stack consumes
    'std_dev_scale_5') and, for scales <= 5, entropy_scale_k (rank entropy over disk(k), divided by its maximum)."""
    d, (h, w) = _dev(band)
    ctx = _ctx()
    n = _P.renormalize(ctx, d)
    q = ctx.quantize_u8(n, 255.0)
    out = {}
    for k in scales:
        if k == 1:  # a 1x1 blur is the identity: variance = x*x - x*x = 0
            out["mean_scale_1"] = _host(n, (h, w))
            out["variance_scale_1"] = np.zeros((h, w), np.float32)
            out["std_dev_scale_1"] = np.zeros((h, w), np.float32)
        else:
            out[f"mean_scale_{k}"] = _host(ctx.box_mean(n, h, w, k, _L.BORDER_REFLECT101), (h, w))
            out[f"variance_scale_{k}"] = _host(ctx.local_var(n, h, w, k), (h, w))
            out[f"std_dev_scale_{k}"] = _host(ctx.local_std(n, h, w, k), (h, w))
        if k <= 5:
            e = _host(ctx.rank_entropy(q, h, w, k), (h, w))
            out[f"entropy_scale_{k}"] = e / np.max(e)
    return out


def calculate_filter_responses(band):
    """reference indices.py:444-482 — gaussian_5, gaussian_15 (uint8 blur / 255.0, float64), dog (their difference,
    min-max normalised), laplacian, sobel_mag (the member the stack consumes)."""
    d, (h, w) = _dev(band)
    ctx = _ctx()
    q = ctx.quantize_u8(_P.renormalize(ctx, d), 255.0)
    return filter_members(ctx, q, h, w)


def filter_members(ctx, q, h, w):
    g5 = _host(ctx.gaussian_blur_u8(q, h, w, 5), (h, w)) / 255.0
    g15 = _host(ctx.gaussian_blur_u8(q, h, w, 15), (h, w)) / 255.0
    dog = g5 - g15
    return {"gaussian_5": g5, "gaussian_15": g15, "dog": (dog - dog.min()) / (dog.max() - dog.min() + 1e-10),
            "laplacian": _host(ctx.laplacian_norm(q, h, w), (h, w)), "sobel_mag": _host(ctx.sobel_mag(q, h, w), (h, w))}


def add_spatial_context(features_array, window_size=7):
    """reference indices.py:760-776: (H,W,C) -> (H,W,2C) float64."""
    features_array = np.asarray(features_array)
    if features_array.ndim != 3:
        raise ValueError("add_spatial_context expects an (H, W, C) array")
    h, w, c = features_array.shape
    if features_array.dtype.kind == "f" and features_array.dtype.itemsize > 4:
        # cv2.boxFilter(feature, -1, ...) answers in the depth of its input: a float64 stack would be averaged and returned in float64
        from rsseg.runtime import RssegUnsupported
        raise RssegUnsupported(f"{features_array.dtype} stack: the context mean is computed on float32 planes (prepare_level_1_features stacks "
                               "float32 planes); cast with .astype(np.float32)")
    ctx = _ctx()
    context = np.zeros((h, w, c))
    for i in range(c):
        d = ctx.to_device(np.ascontiguousarray(features_array[:, :, i], dtype=np.float32).reshape(-1))
        context[:, :, i] = _host(ctx.box_mean(d, h, w, window_size, _L.BORDER_REFLECT), (h, w))
    return np.concatenate([features_array, context], axis=-1)


def prepare_level_1_features(features_dict):
    """reference indices.py:808-835 (pure stacking)."""
    level = [features_dict["ndwi"], features_dict["mndwi"], features_dict["ndvi"], features_dict["evi"],
             features_dict["ndbi"], features_dict["bsi"]]
    if "pca_result" in features_dict and len(features_dict["pca_result"]) > 0:
        level.append(features_dict["pca_result"][0])
    return np.stack(level, axis=-1)


def prepare_level_2_features(features_dict):
    """reference indices.py:837-865 (pure stacking)."""
    level = []
    if "glcm_features" in features_dict:
        level.append(features_dict["glcm_features"]["contrast"])
        level.append(features_dict["glcm_features"]["homogeneity"])
    if "morphological_features" in features_dict and "gradient_5" in features_dict["morphological_features"]:
        level.append(features_dict["morphological_features"]["gradient_5"])
    if "multi_scale_features" in features_dict and "std_dev_scale_5" in features_dict["multi_scale_features"]:
        level.append(features_dict["multi_scale_features"]["std_dev_scale_5"])
    if "filter_features" in features_dict and "sobel_mag" in features_dict["filter_features"]:
        level.append(features_dict["filter_features"]["sobel_mag"])
    return np.stack(level, axis=-1) if level else np.zeros((1, 1, 1))


def visualize_hierarchical_features(hierarchical_features, features_dict):
    """indices.py:867- (matplotlib figures): plotting is outside this path.  Kept as a name so that
    scripts/2_feature_extraction.py:131 resolves; draws nothing."""
    print("[rsseg] visualize_hierarchical_features: plotting is out of scope, nothing drawn")


def visualize_selected_features(features_dict, max_features=12, save_path="selected_features_visualization.png"):
    """indices.py:564-628 (matplotlib figure): plotting is outside this path; kept as a name, writes nothing."""
    print(f"[rsseg] visualize_selected_features: plotting is out of scope, '{save_path}' not written")


PLOTTING_NAMES = ("visualize_hierarchical_features", "visualize_selected_features")   # resolve, draw nothing
