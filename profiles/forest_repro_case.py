import numpy as np
from sklearn.ensemble import ExtraTreesClassifier, RandomForestClassifier
def case(seed):
    rng = np.random.default_rng(9000 + seed)
    F = int(rng.choice([1, 2, 3, 5, 8, 19, 31, 32, 33, 55, 64]))
    ncls = int(rng.choice([2, 3, 4, 5, 8, 9, 16, 17, 33, 40]))
    ntr = int(rng.choice([30, 300, 3000, 30000]))
    Xtr = rng.random((ntr, F)).astype(np.float32)
    if rng.random() < 0.4:
        Xtr = (np.round(Xtr * 8) / 8).astype(np.float32)
    lab = np.sort(rng.choice(1000, ncls, replace=False)) - 500
    ytr = lab[((Xtr[:, 0] * ncls).astype(np.int64) + (rng.random(ntr) < 0.3) * rng.integers(0, ncls, ntr)) % ncls]
    nan_fit = rng.random() < 0.4
    if nan_fit:
        Xtr[rng.random((ntr, F)) < 0.03] = np.nan
    depth = None if rng.random() < 0.25 else int(rng.integers(1, 15))
    kw = dict(n_estimators=int(rng.integers(1, 31)), max_depth=depth, random_state=int(seed), n_jobs=4)
    use_et = rng.random() < 0.25 and not nan_fit
    model = (ExtraTreesClassifier(**kw) if use_et else RandomForestClassifier(bootstrap=bool(rng.random() < 0.7), **kw)).fit(Xtr, ytr)
    n = int(rng.choice([1, 63, 64, 65, 1023, 1025, 5003, 40001]))
    X = rng.random((n, F)).astype(np.float32)
    r1 = rng.random() < 0.5
    if r1:
        X = (np.round(X * 8) / 8).astype(np.float32)
    r2 = rng.random() < 0.6
    if r2:
        X[rng.random((n, F)) < 0.02] = np.nan
    r3 = rng.random() < 0.3
    if r3:
        X[rng.random((n, F)) < 0.01] = np.float32(3.0e38)
        X[rng.random((n, F)) < 0.01] = np.float32(-3.0e38)
    tag = dict(seed=seed, F=F, classes=len(model.classes_), trees=kw["n_estimators"], depth=depth, n=n, nan_fit=nan_fit, et=use_et, grid=r1, nanrows=r2, huge=r3, ntr=ntr)
    return model, X, tag
