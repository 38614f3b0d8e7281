"""
ORACLE — TEST INFRASTRUCTURE ONLY: the CPU counterpart harness of BASELINE.md §3.

What bench.py times as `cpu_baseline`: the reference's NumPy glue restated (oracle/ref_np.py), calling the SAME
scikit-learn estimators the reference calls — RobustScaler + PCA (reference modules/features/indices.py:226-238),
MinMaxScaler + KMeans(random_state=42, n_init='auto') (modules/features/extract.py:568-579),
RandomForestClassifier.predict (modules/supervised_classifiers.py:111) — and the C / NumPy restatements of the
cv2 / scikit-image stages (OpenCV and scikit-image are not installed).  Run once with every host core and once with
one thread (threadpoolctl).  Never imported by the product path.
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Sequence

import numpy as np

from . import ref_np as O

GLCM_KEYS = ("contrast", "dissimilarity", "homogeneity", "energy", "correlation")


def perform_pca_sklearn(bands: Sequence[np.ndarray], n_components=None):
    """perform_pca with the scikit-learn objects themselves (indices.py:205-246)."""
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import RobustScaler
    h, w = bands[0].shape
    X = np.vstack([b.flatten() for b in bands]).T
    Xs = RobustScaler().fit_transform(X)
    pca = PCA(n_components=n_components)
    T = pca.fit_transform(Xs)
    return [T[:, i].reshape(h, w) for i in range(T.shape[1])], pca.explained_variance_ratio_, pca


def kmeans_sklearn(planes: Sequence[np.ndarray], k: int):
    """unsupervised_kmeans_classification's estimator calls (extract.py:553-579)."""
    from sklearn.cluster import KMeans
    from sklearn.preprocessing import MinMaxScaler
    h, w = planes[0].shape
    X = np.vstack([np.nan_to_num(p, nan=0.0).flatten() for p in planes]).T
    Xs = MinMaxScaler().fit_transform(X)
    km = KMeans(n_clusters=k, random_state=42, n_init="auto")
    labels = km.fit_predict(Xs)
    return labels.reshape(h, w), int(km.n_iter_)


def thread_info() -> dict:
    from threadpoolctl import threadpool_info
    info = threadpool_info()
    return dict(host_cores=os.cpu_count(), pools=[dict(api=p.get("user_api"), lib=p.get("internal_api"), threads=p.get("num_threads")) for p in info])


def _timed(fn):
    t0 = time.perf_counter()
    out = fn()
    return out, time.perf_counter() - t0


def config23(bands: List[np.ndarray], cfg: str, k: int, glcm_step: int, threads) -> Dict:
    """BASELINE configs[1] / [2] on host arrays: per-stage wall seconds.  `threads`: None = every core, 1 = one thread
    (threadpoolctl limits OpenBLAS, scikit-learn's OpenMP and the OpenMP loop of the GLCM restatement in oracle.c)."""
    from threadpoolctl import threadpool_limits
    st: Dict[str, float] = {}
    with threadpool_limits(limits=threads):
        norm, st["robust_normalize"] = _timed(lambda: [O.robust_normalize(x) for x in bands])
        bl, g, r, n, s = norm[:5]
        feats, st["indices"] = _timed(lambda: [O.calculate_ndvi(n, r), O.calculate_evi(n, r, bl), O.calculate_msavi(n, r), O.calculate_ndwi(g, n),
                                               O.calculate_mndwi(g, s), O.calculate_ndbi(s, n), O.calculate_bsi(bl, r, n, s)])
        if cfg == "c3":
            (pcs, _, _), st["pca"] = _timed(lambda: perform_pca_sklearn(norm, 3))
            (gl, _), st["glcm"] = _timed(lambda: O.calculate_glcm_features(norm[3], 32, 7, glcm_step))
            feats = feats + [gl[x] for x in GLCM_KEYS] + list(pcs)
        (labels, n_iter), st["kmeans"] = _timed(lambda: kmeans_sklearn(feats, k))
    st["n_iter"] = n_iter
    return st


def config5(bands: List[np.ndarray], model, threads) -> Dict:
    """BASELINE configs[4] on host arrays: the oracle's 19-feature stack, then forest.predict."""
    from threadpoolctl import threadpool_limits
    st: Dict[str, float] = {}
    with threadpool_limits(limits=threads):
        (_, hier), st["features"] = _timed(lambda: O.run_feature_extraction_stage(bands, pca_fn=perform_pca_sklearn))
        model.n_jobs = None if threads == 1 else -1
        _, st["forest"] = _timed(lambda: model.predict(hier["all"].reshape(-1, 19)))
    return st
