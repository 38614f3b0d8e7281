"""
Drop-in counterpart of the classification entry points of the reference's modules/features/extract.py
that are on the hot path: unsupervised_kmeans_classification (extract.py:508-581) and
supervised_classification_predict (extract.py:690-719).  Same signatures, same error behaviour
(ValueError for empty / malformed inputs), NumPy in -> NumPy out.
"""
from __future__ import annotations

import numpy as np

from rsseg.runtime import default_context as _ctx

__all__ = ["unsupervised_kmeans_classification", "supervised_classification_predict", "np"]


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
