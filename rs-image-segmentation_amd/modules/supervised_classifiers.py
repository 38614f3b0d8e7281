"""
Drop-in counterpart of predict_image of the reference's modules/supervised_classifiers.py:99-115: a
fitted sklearn RandomForestClassifier (or an already flattened forest dict) applied to every pixel of an
(H, W, D) feature array by the K11 forest-walk kernel.  Like the reference, any failure is reported and
answered with an all-zero map (supervised_classifiers.py:113-115) — except a forest beyond the capacity of the
kernels (more than 64 features or classes): that is this library's limit, not a failure the reference would have had,
and raises rsseg.runtime.RssegUnsupported instead of returning an empty map.
"""
from __future__ import annotations

import numpy as np

from rsseg.forest import flatten_forest
from rsseg.runtime import RssegUnsupported
from rsseg.runtime import default_context as _ctx

__all__ = ["predict_image", "np"]


def _predict_planes(model, planes):
    forest = model if isinstance(model, dict) else flatten_forest(model)
    ctx = _ctx()
    ctx.forest_load(forest)
    dev = [ctx.to_device(np.ascontiguousarray(p, dtype=np.float32).reshape(-1)) for p in planes]  # _forest.py:640 float32 cast
    out = ctx.forest_predict(dev).cpu().numpy()
    return out.astype(np.asarray(forest["classes"]).dtype, copy=False)


def predict_image(model, features):
    try:
        h, w, d = features.shape
        planes = [features[:, :, i] for i in range(d)]
        return _predict_planes(model, planes).reshape(h, w)
    except RssegUnsupported:
        raise
    except Exception as e:  # noqa: BLE001 — reference behaviour
        print("❌ 预测失败:", e)
        return np.zeros(features.shape[:2], dtype=int)
