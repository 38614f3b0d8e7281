"""ORACLE — TEST INFRASTRUCTURE ONLY.  How far does "bit-identical to scikit-learn" reach on inputs that are NOT the reference's
feature data?  The KMeans restatement (oracle.c / kmeans_impl.h) keeps scikit-learn's per-pixel arithmetic and replaces every
reduction over all pixels by an exact fixed-point sum, because scikit-learn's own result depends on BLAS / OpenMP summation order
there.  On the reference's feature data that reproduces the reference's labels (tests/golden).  This
script runs the random matrices of tests/test_gpu_fuzz.py::test_fuzz_kmeans_entry_point (uniform noise, coarse grids with exact
ties, blobs, NaNs; 1..64 features, 1..64 clusters, float32 / float64) through
    A  scikit-learn, one thread            (threadpool_limits(1))
    B  scikit-learn, all cores
    C  the oracle
and counts the cases whose label maps differ:  python oracle/campaign_kmeans_vs_sklearn.py LO HI  ->  one JSON line."""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from oracle import ref_np as oracle  # noqa: E402


def gen(seed):
    rng = np.random.default_rng(7000 + seed)
    F = int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 19, 32, 33, 40, 64]))
    dt = np.float64 if rng.random() < 0.3 else np.float32
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 5000), rng.integers(5000, 60000)]))
    k = int(min(n, rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 32, 33, 64])))
    X = rng.random((n, F))
    style = str(rng.choice(["continuous", "grid", "blobs", "nan"]))
    if style == "grid":
        X = np.round(X * int(rng.integers(1, 5))) / 4.0
    elif style == "blobs":
        c = rng.random((max(k, 2), F)) * 4
        X = c[rng.integers(0, c.shape[0], n)] + rng.normal(0, 0.05, (n, F))
    elif style == "nan":
        X[rng.random((n, F)) < 0.02] = np.nan
    return [np.ascontiguousarray(X[:, f]).astype(dt) for f in range(F)], k, dt, style


def main():
    from sklearn.cluster import KMeans
    from sklearn.preprocessing import MinMaxScaler
    from threadpoolctl import threadpool_limits
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    ab = ac = both = 0
    by_style = {}
    for seed in range(lo, hi):
        planes, k, dt, style = gen(seed)
        Xm = MinMaxScaler().fit_transform(np.nan_to_num(np.stack(planes, 1).astype(dt), nan=0.0))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with threadpool_limits(limits=1):
                a = KMeans(n_clusters=k, random_state=42, n_init="auto").fit_predict(Xm)
            b = KMeans(n_clusters=k, random_state=42, n_init="auto").fit_predict(Xm)
        c, _ = oracle.kmeans_fit_planes(planes, k)
        s, o = not np.array_equal(a, b), not np.array_equal(a, c)
        ab += s
        ac += o
        both += s and o
        st = by_style.setdefault(style, [0, 0, 0])
        st[0] += 1
        st[1] += s
        st[2] += o
    print(json.dumps({"seeds": [lo, hi], "cases": hi - lo, "sklearn_1_thread_vs_all_cores_differ": ab, "oracle_vs_sklearn_1_thread_differ": ac,
                      "both": both, "by_style_cases_selfdiffer_oraclediffer": by_style}))


if __name__ == "__main__":
    main()
