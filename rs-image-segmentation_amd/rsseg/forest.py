"""sklearn RandomForestClassifier -> the flat arrays rsseg_forest_load takes (host plumbing only).
tree_ node records: children_left/right (-1 = leaf), feature, threshold (float64), missing_go_to_left,
value (n_nodes, 1, n_classes) class fractions (sklearn >= 1.3) — SURVEY.md §8c item 2."""
from __future__ import annotations

import numpy as np


def flatten_forest(model) -> dict:
    offs, left, right, feat, thr, miss, val = [0], [], [], [], [], [], []
    n_classes = len(model.classes_)
    for est in model.estimators_:
        t = est.tree_
        left.append(np.asarray(t.children_left, np.int32))
        right.append(np.asarray(t.children_right, np.int32))
        feat.append(np.asarray(t.feature, np.int32))
        thr.append(np.asarray(t.threshold, np.float64))
        mg = getattr(t, "missing_go_to_left", None)
        miss.append(np.zeros(t.node_count, np.uint8) if mg is None else np.asarray(mg, np.uint8))
        v = np.asarray(t.value[:, 0, :n_classes], np.float64)
        s = v.sum(axis=1, keepdims=True)
        if not np.allclose(s[s > 0], 1.0):  # models pickled by sklearn < 1.3 store counts
            s[s == 0] = 1.0
            v = v / s
        val.append(v)
        offs.append(offs[-1] + t.node_count)
    return dict(tree_off=np.asarray(offs, np.int64), left=np.concatenate(left), right=np.concatenate(right),
                feature=np.concatenate(feat), threshold=np.concatenate(thr), missing_left=np.concatenate(miss),
                value=np.ascontiguousarray(np.concatenate(val)), classes=np.asarray(model.classes_),
                n_features=int(model.n_features_in_))
