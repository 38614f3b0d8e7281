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
        from rsseg.tiff import read_tiff
        arr = read_tiff(file_path)
        out["all_features"] = {f"band_{i + 1}": arr[i] for i in range(arr.shape[0])}
        out["transform"] = None
        out["crs"] = None
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
