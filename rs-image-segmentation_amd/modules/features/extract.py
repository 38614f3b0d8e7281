"""
Drop-in counterpart of the reference's modules/features/extract.py: every function scripts/3_classification.py reaches
through `from modules.features.extract import *` (scripts/3:25) under the reference's name, positional order, defaults,
return dtypes and error behaviour (ValueError / FileNotFoundError), NumPy in -> NumPy out, with the per-pixel work done by
librsseg_hip.so: KMeans (extract.py:508-581), forest inference (:690-719), the rule-based extractors (:299-505), the
feature-file plumbing (:32-295) and the GeoTIFF writer (:778-833).  Forest TRAINING (:585-687) stays scikit-learn on the
host, as in the reference.  The two plotting functions (:723-775, :840-) are outside the hot path: they exist so that a
script's call resolves, print one line and draw nothing.  `__all__` at the end of the file is the star-import surface;
tests/golden/star_import_names.json (oracle/gen_names.py) lists what the reference's scripts resolve through it.
"""
from __future__ import annotations

import numpy as np

from rsseg.runtime import default_context as _ctx

import os
import pickle

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
    """extract.py:299-341: closing with the elliptical k x k element (odd k; an even or zero k: scipy's binary_fill_holes
    instead, :314-316), removal of 8-connected components smaller than min_area, opening with the same element (odd k only;
    an even k prints the reference's warning and skips it, :337-338)."""
    if binary_mask is None or np.asarray(binary_mask).size == 0:
        print("警告: 输入的二值掩码为空，后处理跳过。")
        return binary_mask
    a = np.asarray(binary_mask)
    if a.ndim != 2:
        raise ValueError("advanced_post_processing expects a 2-D mask")
    # the reference works on the mask's VALUES as uint8 (cv2 min / max filters, :306); the library's planes hold 0 / 1,
    # which is what every caller passes (threshold masks)
    d, (h, w) = _mask_dev(a)
    return _post(_ctx(), d, h, w, min_area, smooth_kernel_size, fill_holes).cpu().numpy().reshape(h, w)


def _post(ctx, d, h, w, min_area, k, fill_holes=True):
    from rsseg import _lib as L
    odd = k > 0 and k % 2 == 1
    if odd and k > 31:
        from rsseg.runtime import RssegUnsupported
        raise RssegUnsupported(f"advanced_post_processing: smooth_kernel_size {k} > 31")
    if fill_holes and odd:
        d = ctx.morph_ellipse(d, h, w, k, L.MORPH_CLOSE) if k > 1 else d     # a 1 x 1 element changes nothing
    elif fill_holes:
        d = ctx.fill_holes(d, h, w)
    if min_area > 0:
        d = ctx.remove_small_components(d, h, w, int(min_area))
    if odd:
        d = ctx.morph_ellipse(d, h, w, k, L.MORPH_OPEN) if k > 1 else d
    elif k > 0:
        print(f"警告: 平滑核大小 {k} 不是奇数，形态学平滑可能效果不佳或出错。")
    return d


def threshold_segmentation(feature_image, threshold_value, above=True, otsu=False):
    """extract.py:344-404: NaN -> 0, then > / < threshold -> uint8 mask; otsu=True ignores threshold_value: the plane is
    stretched to 8 bits between its extrema and cut at cv2.threshold(THRESH_OTSU)'s level (:358-371; a plane without
    contrast gives all 0 / all 1).  float32 planes are compared in float32, anything else in float64, as NumPy does."""
    if feature_image is None:
        raise ValueError("输入的特征图像为空。")
    a = np.asarray(feature_image)
    dt = np.float32 if a.dtype == np.float32 else np.float64
    ctx = _ctx()
    if a.size == 0:
        return np.zeros(a.shape, np.uint8)
    d = ctx.to_device(np.ascontiguousarray(a, dtype=dt).reshape(-1))
    if otsu:
        m, level, _, _ = ctx.otsu_mask(d, above)
        if level < 0:
            print("警告: 特征图像所有值相同，Otsu无法应用。返回全黑或全白掩码。")
        return m.cpu().numpy().reshape(a.shape)
    t = float(np.float32(threshold_value)) if dt == np.float32 else float(threshold_value)
    m = ctx.threshold_band(d, t, float("inf")) if above else ctx.threshold_band(d, float("-inf"), t)
    return m.cpu().numpy().reshape(a.shape)


def _f32(features_dict, key, shape=None):
    v = features_dict.get(key)
    if v is None or (shape is not None and np.asarray(v).shape != shape):
        return None
    v = np.asarray(v)
    if v.dtype.kind == "f" and v.dtype.itemsize > 4:
        # NumPy compares a float64 plane with the rule's thresholds in float64 (0.2 is not float32(0.2)); these functions compare in
        # float32, the dtype of the index planes the feature stage writes.  Refused by name rather than narrowed silently;
        # threshold_segmentation itself takes float64 planes (its own float64 kernel).
        from rsseg.runtime import RssegUnsupported
        raise RssegUnsupported(f"feature '{key}' is {v.dtype}: the rule functions compare float32 planes (what the feature stage writes); "
                               "cast with .astype(np.float32), or call threshold_segmentation, which compares float64 planes in float64")
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


# --------------------------------------------------------------------------------------------------
# forest training helpers (reference extract.py:585-687): host-side scikit-learn, as in the reference — fitting is not
# on the accelerated path; inference with the fitted model is (supervised_classification_predict above)
# --------------------------------------------------------------------------------------------------
def prepare_training_samples(feature_array, labeled_roi_path):
    """extract.py:585-633: rows of the (H, W, F) stack under the non-zero pixels of a single-band label raster
    (read with rsseg.tiff instead of rasterio) -> (X, y); NaN features -> 0."""
    if not os.path.exists(labeled_roi_path):
        raise FileNotFoundError(f"标签ROI文件未找到: {labeled_roi_path}")
    if feature_array is None or getattr(feature_array, "ndim", 0) != 3:
        raise ValueError("输入的feature_array必须是3D NumPy数组 (height, width, n_features)。")
    from rsseg.tiff import read_tiff
    h, w, f = feature_array.shape
    labels = read_tiff(labeled_roi_path)[0]
    if labels.shape != (h, w):
        raise ValueError(f"标签ROI文件 '{labeled_roi_path}' (形状 {labels.shape}) 尺寸与特征数据 (期望的2D形状 {(h, w)}) 不匹配。")
    flat = labels.reshape(-1)
    keep = np.flatnonzero((flat != 0) & ~np.isnan(flat))
    if keep.size == 0:
        raise ValueError("未能从ROI中提取任何有效的训练样本。请检查标签ROI文件和类别定义。")
    X = feature_array.reshape(-1, f)[keep, :]
    if np.isnan(X).any():
        print("警告: 提取的训练样本中存在NaN值，将用0填充。")
        X = np.nan_to_num(X, nan=0.0)
    return X, flat[keep]


def train_random_forest_classifier(X_train, y_train, feature_names_for_training, n_estimators=100, test_size=0.3, random_state=42):
    """extract.py:635-687: stratified 70 / 30 split (when every class has two samples), RandomForestClassifier(n_estimators,
    random_state, n_jobs=-1).fit, validation accuracy / kappa / importances printed.  Returns the fitted classifier."""
    from sklearn.ensemble import RandomForestClassifier
    from sklearn.metrics import accuracy_score, classification_report, cohen_kappa_score
    from sklearn.model_selection import train_test_split
    classes, counts = np.unique(y_train, return_counts=True)
    stratify = y_train if len(classes) > 1 and counts.min() >= 2 else None
    if len(classes) > 1 and stratify is None:
        print(f"警告: 部分类别样本数少于2个 ({dict(zip(classes, counts))})。训练/验证分割时这些类别可能无法进行分层抽样。")
    X_t, X_val, y_t, y_val = train_test_split(X_train, y_train, test_size=test_size, random_state=random_state, stratify=stratify)
    print(f"训练样本数: {X_t.shape[0]}, 验证样本数: {X_val.shape[0]}")
    if X_t.shape[0] == 0:
        raise ValueError("没有足够的训练样本进行分割。")
    clf = RandomForestClassifier(n_estimators=n_estimators, random_state=random_state, n_jobs=-1)
    clf.fit(X_t, y_t)
    if X_val.shape[0] == 0:
        print("警告: 验证样本数为0，跳过验证评估。")
        return clf
    pred = clf.predict(X_val)
    print(f"随机森林分类器验证集性能:\n  准确率: {accuracy_score(y_val, pred):.4f}\n  Kappa系数: {cohen_kappa_score(y_val, pred):.4f}")
    print(classification_report(y_val, pred, labels=np.unique(np.concatenate((y_val, pred))), zero_division=0))
    imp = clf.feature_importances_
    if len(feature_names_for_training) == imp.size:
        for i in np.argsort(imp)[::-1]:
            print(f"    特征 '{feature_names_for_training[i]}': {imp[i]:.4f}")
    else:
        print("\n警告: 特征重要性无法显示。特征名称数量与分类器的特征重要性数组大小不匹配。")
    return clf


# --------------------------------------------------------------------------------------------------
# output (reference extract.py:778-833) and the plotting names (out of scope: SURVEY.md §2)
# --------------------------------------------------------------------------------------------------
def save_classification_as_geotiff(classification_result, features_meta, output_tif_path):
    """extract.py:778-833: one band, nodata 0, LZW in 256 x 256 tiles; uint8 when the labels fit, else uint16, else int32;
    float labels are rounded.  An empty result, incomplete metadata ('transform', 'crs', 'width', 'height'), a shape that
    contradicts the metadata or a write error are reported with print and the function returns None, like the reference.
    The file is written by rsseg.tiff (rasterio / GDAL are not needed)."""
    if classification_result is None or np.asarray(classification_result).size == 0:
        print("分类结果为空，无法保存为GeoTIFF。")
        return
    if not all(k in features_meta and features_meta[k] is not None for k in ("transform", "crs", "width", "height")):
        print("警告: 用于保存GeoTIFF的元数据不完整。需要 'transform', 'crs', 'width', 'height'。")
        print(f"可用的元数据键: {list(features_meta.keys())}")
        return
    try:
        a = np.asarray(classification_result)
        if a.max() <= 255 and a.min() >= 0:
            dt = np.uint8
        elif a.max() <= 65535 and a.min() >= 0:
            dt = np.uint16
        else:
            dt = np.int32
        if np.issubdtype(a.dtype, np.floating):
            print("警告: 分类结果包含浮点数标签，将转换为整数。")
            a = np.round(a).astype(dt)
        else:
            a = a.astype(dt)
        if a.ndim != 2 or a.shape[0] != features_meta["height"] or a.shape[1] != features_meta["width"]:
            print(f"Warning: Classification result shape {a.shape} does not match dimensions in metadata "
                  f"({features_meta['height']}, {features_meta['width']}). Skipping GeoTIFF save.")
            return
        from rsseg.stages import _epsg_of, _is_geographic
        from rsseg.tiff import write_tiff
        crs = features_meta["crs"]
        write_tiff(output_tif_path, a, transform=features_meta["transform"], epsg=_epsg_of(crs), geographic=_is_geographic(crs),
                   nodata=0, compress="lzw", tiled=True)
        print(f"分类结果已保存为GeoTIFF: {output_tif_path}")
    except Exception as e:  # noqa: BLE001 — extract.py:832-833 prints and returns
        print(f"保存GeoTIFF文件 '{output_tif_path}' 时出错: {e}")


def create_classification_map(classification_result, class_names_map, class_colors_map, save_path="classification_map.png", title="地物分类图"):
    """extract.py:723-775 draws a PNG with matplotlib: plotting is outside this path.  Kept as a name so that
    scripts/3:493 resolves; writes nothing."""
    print(f"[rsseg] create_classification_map: plotting is out of scope, '{save_path}' not written")


def visualize_combined_indices(features_dict, output_dir="visualization_outputs", save_path="combined_indices_map.png"):
    """extract.py:840- (matplotlib figure): plotting is outside this path; kept as a name (scripts/3:613), writes nothing."""
    print(f"[rsseg] visualize_combined_indices: plotting is out of scope, '{os.path.join(output_dir, save_path)}' not written")


PLOTTING_NAMES = ("create_classification_map", "visualize_combined_indices")   # resolve, draw nothing

# The star-import surface (scripts/3_classification.py:25).  The reference module has no __all__, so its import * also
# leaks its library imports; of those the script uses np, os, pickle (scripts/3:43, 145, 598) — and cv2 on one defensive
# line (:351, a mask of the wrong shape), which is not re-exported: cv2 is not a dependency of this implementation.
__all__ = [
    "load_features", "normalize_features_structure",
    "advanced_post_processing", "threshold_segmentation",
    "extract_vegetation_by_threshold", "extract_water_by_threshold", "extract_builtup_by_threshold", "extract_bareland_by_rule",
    "rule_based_classification",
    "unsupervised_kmeans_classification",
    "prepare_training_samples", "train_random_forest_classifier", "supervised_classification_predict",
    "create_classification_map", "save_classification_as_geotiff", "visualize_combined_indices",
    "np", "os", "pickle",
]
