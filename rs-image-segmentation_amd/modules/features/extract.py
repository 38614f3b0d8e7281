"""
Drop-in counterpart of the classification entry points of the reference's modules/features/extract.py
that are on the hot path: unsupervised_kmeans_classification (extract.py:508-581) and
supervised_classification_predict (extract.py:690-719).  Same signatures, same error behaviour
(ValueError for empty / malformed inputs), NumPy in -> NumPy out.
"""
from __future__ import annotations

import numpy as np

from rsseg.runtime import default_context as _ctx

import os
import pickle

__all__ = ["load_features", "normalize_features_structure", "unsupervised_kmeans_classification",
           "supervised_classification_predict", "np", "os", "pickle"]

_META_MAP = {"geo_transform": "transform", "crs": "crs", "dimensions": "dimensions", "width": "width", "height": "height",
             "transform": "transform"}


def load_features(file_path):
    """extract.py:32-121: .npy (dict or (bands, H, W) array), .pkl, or .tif -> raw feature dict.
    Raises FileNotFoundError / ValueError like the reference.  TIFFs go through rsseg.tiff (plain strip TIFFs)."""
    if not os.path.exists(file_path):
        raise FileNotFoundError(f"特征文件未找到: {file_path}")
    ext = os.path.splitext(file_path)[1].lower()
    out = {}
    if ext == ".npy":
        data = np.load(file_path, allow_pickle=True)
        if data.ndim == 0 and isinstance(data.item(), dict) and data.item():
            out = data.item()
        elif data.ndim == 3:
            out["all_features"] = {f"feature_{i + 1}": data[i] for i in range(data.shape[0])}
            if data.shape[1] > 0 and data.shape[2] > 0:
                out["dimensions"] = (data.shape[1], data.shape[2])
        else:
            raise ValueError(f".npy 文件内容格式未知或不符合预期 (shape: {data.shape})。期望字典或 (bands, H, W) 数组。")
    elif ext == ".pkl":
        with open(file_path, "rb") as f:
            out = pickle.load(f)
    elif ext in (".tif", ".tiff"):
        from rsseg.tiff import read_tiff, read_tiff_georef
        arr = read_tiff(file_path)
        geo = read_tiff_georef(file_path)
        out["all_features"] = {f"band_{i + 1}": arr[i] for i in range(arr.shape[0])}
        out["transform"] = geo["transform"]
        out["crs"] = None if geo["epsg"] is None else f"EPSG:{geo['epsg']}"
        out["width"], out["height"] = int(arr.shape[2]), int(arr.shape[1])
        out["dimensions"] = (int(arr.shape[1]), int(arr.shape[2]))
    else:
        raise ValueError(f"Unsupported feature file format: {ext} (from file: {file_path})")
    return out


def normalize_features_structure(loaded_features):
    """extract.py:124-295: flatten nested dicts / lists into top-level lower-case keys '<outer>_<inner>'
    (list members by index), keep only arrays with ndim >= 2, map the metadata keys
    (geo_transform -> transform, ...), derive integer height / width from 'dimensions' or from the first
    array, drop 'dimensions'.  First occurrence of a key wins."""
    normalized = {}

    def walk(data, prefix):
        if isinstance(data, np.ndarray) and data.ndim >= 2:
            key = prefix.lower()
            if key and key not in normalized:
                normalized[key] = data
        elif isinstance(data, dict):
            for k, v in data.items():
                walk(v, f"{prefix}_{k}" if prefix else k)
        elif isinstance(data, list):
            for i, v in enumerate(data):
                walk(v, f"{prefix}_{i}" if prefix else str(i))

    for src, dst in _META_MAP.items():
        if src in loaded_features and dst not in normalized:
            normalized[dst] = loaded_features[src]
    meta_targets = set(_META_MAP.values())
    for k, v in loaded_features.items():
        if k.lower() in meta_targets:
            continue
        walk(v, k)

    def have_hw():
        return isinstance(normalized.get("height"), int) and isinstance(normalized.get("width"), int)

    if not have_hw() and isinstance(normalized.get("dimensions"), tuple):
        d = normalized["dimensions"]
        if len(d) == 2:
            normalized["height"], normalized["width"] = int(d[0]), int(d[1])
        elif len(d) >= 2:
            if len(d) == 3 and d[0] < d[-2] and d[0] < d[-1]:
                normalized["height"], normalized["width"] = int(d[1]), int(d[2])
            else:
                normalized["height"], normalized["width"] = int(d[0]), int(d[1])
    if not have_hw():
        for v in list(normalized.values()):
            if isinstance(v, np.ndarray) and v.ndim >= 2:
                if v.ndim == 2:
                    normalized["height"], normalized["width"] = int(v.shape[0]), int(v.shape[1])
                elif v.ndim == 3:
                    normalized["height"], normalized["width"] = int(v.shape[1]), int(v.shape[2])
                if have_hw():
                    break
    if "height" in normalized and "width" in normalized and "dimensions" in normalized:
        del normalized["dimensions"]
    return normalized



def _select_planes(features_dict, feature_keys_to_use):
    """extract.py:510-566: metadata checks, automatic key selection, per-channel flattening."""
    if not features_dict or "height" not in features_dict or "width" not in features_dict:
        raise ValueError("特征字典为空或缺少图像尺寸信息 (height/width)。")
    shape = (features_dict["height"], features_dict["width"])
    if feature_keys_to_use is None:
        meta = ["transform", "crs", "width", "height", "dimensions", "geo_transform"]
        keys = [k for k, v in features_dict.items()
                if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == shape and k not in meta]
        if not keys:
            cands = ["ndvi", "ndwi", "ndbi", "texture_mean", "evi", "savi", "hierarchical_level_1", "hierarchical_level_2",
                     "hierarchical_all"]
            keys = [k for k in cands if k in features_dict and isinstance(features_dict[k], np.ndarray)
                    and ((features_dict[k].ndim == 2 and features_dict[k].shape == shape)
                         or (features_dict[k].ndim == 3 and features_dict[k].shape[:2] == shape))]
        feature_keys_to_use = keys
    if not feature_keys_to_use:
        raise ValueError("没有可用于K-Means的特征。请检查特征字典内容或手动指定 `feature_keys_to_use`。")
    planes = []
    for key in feature_keys_to_use:
        v = features_dict.get(key)
        if isinstance(v, np.ndarray) and v.ndim == 3 and v.shape[:2] == shape:
            planes.extend(v[:, :, i] for i in range(v.shape[2]))
        elif isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == shape:
            planes.append(v)
        # anything else is skipped with a warning in the reference (extract.py:558-562)
    if not planes:
        raise ValueError("未能准备任何特征数据进行K-Means分类。")
    return planes, shape


def unsupervised_kmeans_classification(features_dict, n_clusters=5, feature_keys_to_use=None):
    """extract.py:508-581 -> (H, W) int32 labels 0..n_clusters-1.  The stacked matrix is float32 iff every
    selected array is float32 (np.vstack promotion, extract.py:568), else float64."""
    planes, shape = _select_planes(features_dict, feature_keys_to_use)
    dt = np.result_type(*[p.dtype for p in planes])
    dt = np.float32 if dt == np.float32 else np.float64
    ctx = _ctx()
    dev = [ctx.to_device(np.ascontiguousarray(p, dtype=dt).reshape(-1)) for p in planes]
    labels, _ = ctx.kmeans_fit_predict(dev, int(n_clusters))
    return labels.cpu().numpy().reshape(shape)


def supervised_classification_predict(feature_array, classifier):
    """extract.py:690-719: NaN -> 0, classifier.predict over all pixels, (H, W) in the dtype of classes_."""
    if feature_array is None or getattr(feature_array, "ndim", 0) != 3:
        raise ValueError("输入的feature_array必须是3D NumPy数组 (height, width, n_features)。")
    from modules.supervised_classifiers import _predict_planes
    h, w, d = feature_array.shape
    planes = [np.nan_to_num(feature_array[:, :, i], nan=0.0) if np.isnan(feature_array[:, :, i]).any() else feature_array[:, :, i]
              for i in range(d)]
    return _predict_planes(classifier, planes).reshape(h, w)


# --------------------------------------------------------------------------------------------------
# rule-based classification (reference extract.py:299-505; merge of scripts/3_classification.py:335-375)
# --------------------------------------------------------------------------------------------------
def _mask_dev(mask):
    a = np.asarray(mask)
    return _ctx().to_device(np.ascontiguousarray(a != 0, dtype=np.uint8).reshape(-1)), a.shape


def advanced_post_processing(binary_mask, min_area=100, smooth_kernel_size=3, fill_holes=True):
    """extract.py:299-341: closing with the elliptical k x k element, removal of 8-connected components smaller than
    min_area, opening with the same element.  (Even kernel sizes fall back to scipy's binary_fill_holes in the
    reference; that branch is not reproduced: ValueError.)"""
    if binary_mask is None or np.asarray(binary_mask).size == 0:
        return binary_mask
    d, (h, w) = _mask_dev(binary_mask)
    return _post(_ctx(), d, h, w, min_area, smooth_kernel_size, fill_holes).cpu().numpy().reshape(h, w)


def _post(ctx, d, h, w, min_area, k, fill_holes=True):
    from rsseg import _lib as L
    if k > 0 and k % 2 == 0:
        raise ValueError("advanced_post_processing: even smooth_kernel_size (binary_fill_holes fallback) is not implemented")
    if k not in (0, 3, 5):
        raise ValueError("advanced_post_processing: smooth_kernel_size must be 3 or 5")
    if fill_holes and k > 0:
        d = ctx.morph_ellipse(d, h, w, k, L.MORPH_CLOSE)
    if min_area > 0:
        d = ctx.remove_small_components(d, h, w, int(min_area))
    if k > 0:
        d = ctx.morph_ellipse(d, h, w, k, L.MORPH_OPEN)
    return d


def threshold_segmentation(feature_image, threshold_value, above=True, otsu=False):
    """extract.py:344-404 (otsu=False, the only form the stage uses): NaN -> 0, then > / < threshold, uint8."""
    if feature_image is None:
        raise ValueError("输入的特征图像为空。")
    if otsu:
        raise ValueError("threshold_segmentation: otsu=True (cv2.threshold) is not implemented")
    a = np.asarray(feature_image)
    d = _ctx().to_device(np.ascontiguousarray(a, dtype=np.float32).reshape(-1))
    t = float(np.float32(threshold_value)) if a.dtype == np.float32 else float(threshold_value)
    m = _ctx().threshold_band(d, t, float("inf")) if above else _ctx().threshold_band(d, float("-inf"), t)
    return m.cpu().numpy().reshape(a.shape)


def _f32(features_dict, key, shape=None):
    v = features_dict.get(key)
    if v is None or (shape is not None and np.asarray(v).shape != shape):
        return None
    return _ctx().to_device(np.ascontiguousarray(v, dtype=np.float32).reshape(-1))


def _zeros_or_empty(features_dict):
    if "height" in features_dict and "width" in features_dict:
        return np.zeros((features_dict["height"], features_dict["width"]), dtype=np.uint8)
    return np.array([])


def extract_vegetation_by_threshold(features_dict, ndvi_threshold=0.2, post_process=True, min_area=100):
    """extract.py:406-418"""
    if features_dict.get("ndvi") is None:
        return _zeros_or_empty(features_dict)
    ctx = _ctx()
    h, w = np.asarray(features_dict["ndvi"]).shape
    m = ctx.threshold_band(_f32(features_dict, "ndvi"), float(np.float32(ndvi_threshold)), float("inf"))
    if post_process:
        m = _post(ctx, m, h, w, min_area, 3)
    return m.cpu().numpy().reshape(h, w)


def extract_water_by_threshold(features_dict, ndwi_threshold=0.0, mndwi_threshold=0.1, use_mndwi_if_available=True, post_process=True,
                               min_area=50):
    """extract.py:420-443: MNDWI when present (its own threshold), else NDWI."""
    ctx = _ctx()
    if use_mndwi_if_available and features_dict.get("mndwi") is not None:
        key, thr = "mndwi", mndwi_threshold
    elif features_dict.get("ndwi") is not None:
        key, thr = "ndwi", ndwi_threshold
    else:
        return _zeros_or_empty(features_dict)
    h, w = np.asarray(features_dict[key]).shape
    m = ctx.threshold_band(_f32(features_dict, key), float(np.float32(thr)), float("inf"))
    if post_process:
        m = _post(ctx, m, h, w, min_area, 3)
    return m.cpu().numpy().reshape(h, w)


def extract_builtup_by_threshold(features_dict, ndbi_threshold=0.0, ndvi_threshold_for_builtup=0.15, post_process=True, min_area=150):
    """extract.py:446-470: NDBI above its threshold and (when NDVI of the same shape exists) NDVI below its own;
    post-processing with the 5 x 5 element."""
    from rsseg import _lib as L
    if features_dict.get("ndbi") is None:
        return _zeros_or_empty(features_dict)
    ctx = _ctx()
    h, w = np.asarray(features_dict["ndbi"]).shape
    m = ctx.threshold_band(_f32(features_dict, "ndbi"), float(np.float32(ndbi_threshold)), float("inf"))
    nd = _f32(features_dict, "ndvi", (h, w))
    if nd is not None:
        m = ctx.mask_op(m, ctx.threshold_band(nd, float("-inf"), float(np.float32(ndvi_threshold_for_builtup))), L.MASK_AND)
    if post_process:
        m = _post(ctx, m, h, w, min_area, 5)
    return m.cpu().numpy().reshape(h, w)


def extract_bareland_by_rule(features_dict, vegetation_mask, water_mask, builtup_mask, ndvi_low_threshold=-0.1, ndvi_high_threshold=0.2,
                             ndbi_low_threshold=-0.2, ndbi_high_threshold=0.2, post_process=True, min_area=80):
    """extract.py:473-505: what the three masks leave, with NDVI and NDBI inside their bands."""
    from rsseg import _lib as L
    if "height" not in features_dict or "width" not in features_dict:
        return np.array([])
    ctx = _ctx()
    h, w = features_dict["height"], features_dict["width"]
    excl = ctx.to_device(np.zeros(h * w, np.uint8))
    for mk in (vegetation_mask, water_mask, builtup_mask):
        if mk is not None and np.asarray(mk).shape == (h, w):
            excl = ctx.mask_op(excl, _mask_dev(mk)[0], L.MASK_OR)
    m = ctx.mask_op(excl, None, L.MASK_NOT)
    for key, lo, hi in (("ndvi", ndvi_low_threshold, ndvi_high_threshold), ("ndbi", ndbi_low_threshold, ndbi_high_threshold)):
        d = _f32(features_dict, key, (h, w))
        if d is not None:
            m = ctx.mask_op(m, ctx.threshold_band(d, float(np.float32(lo)), float(np.float32(hi)), nan_as_zero=False), L.MASK_AND)  # NaN: False
    if post_process:
        m = _post(ctx, m, h, w, min_area, 3)
    return m.cpu().numpy().reshape(h, w)


def rule_based_classification(features):
    """The 'rule_based' branch of run_classification_stage (scripts/3_classification.py:335-375) on a dict holding 'ndvi',
    'ndbi', 'mndwi' / 'ndwi' planes and 'height' / 'width': thresholds 0.25 / 0.05 / (0.0, 0.2), minimum areas of
    0.05 % / 0.02 % / 0.1 % of the image, priority water > vegetation > built-up, then bare land (4) on what is left.
    Everything stays on the device between the steps; (H, W) uint8 with 0 = unclassified."""
    from rsseg import _lib as L
    ctx = _ctx()
    h, w = int(features["height"]), int(features["width"])
    n = h * w
    final = ctx.to_device(np.zeros(n, np.uint8))

    def plane(key):
        return _f32(features, key, (h, w))

    ndvi, ndbi = plane("ndvi"), plane("ndbi")
    water_src = plane("mndwi")
    water_thr = 0.1                                   # extract_water_by_threshold's mndwi_threshold default (the 0.05 passed
    if water_src is None:                             # by the script is the NDWI threshold, used only without MNDWI)
        water_src, water_thr = plane("ndwi"), 0.05
    veg = _post(ctx, ctx.threshold_band(ndvi, float(np.float32(0.25)), float("inf")), h, w, int(n * 0.0005), 3) if ndvi is not None else None
    water = _post(ctx, ctx.threshold_band(water_src, float(np.float32(water_thr)), float("inf")), h, w, int(n * 0.0002), 3) if water_src is not None else None
    built = None
    if ndbi is not None:
        built = ctx.threshold_band(ndbi, float(np.float32(0.0)), float("inf"))
        if ndvi is not None:
            built = ctx.mask_op(built, ctx.threshold_band(ndvi, float("-inf"), float(np.float32(0.2))), L.MASK_AND)
        built = _post(ctx, built, h, w, int(n * 0.001), 5)
    for mk, val in ((built, 3), (veg, 1), (water, 2)):     # lowest priority first (scripts/3:361-363)
        if mk is not None:
            ctx.mask_paint(final, mk, val)
    # bare land: not vegetation / water / built-up in the merged map, NDVI in (-0.1, 0.2), NDBI in (-0.2, 0.2)
    bare = ctx.mask_op(final, None, L.MASK_NOT)                                         # final == 0
    for d, lo, hi in ((ndvi, -0.1, 0.2), (ndbi, -0.2, 0.2)):
        if d is not None:
            bare = ctx.mask_op(bare, ctx.threshold_band(d, float(np.float32(lo)), float(np.float32(hi)), nan_as_zero=False), L.MASK_AND)  # extract.py:486-497
    bare = _post(ctx, bare, h, w, int(n * 0.0005), 3)
    ctx.mask_paint(final, bare, 4, only_unset=True)
    return final.cpu().numpy().reshape(h, w)

