"""GPU suite, randomised differential cases: raster shapes, value distributions and parameters drawn from a seeded
generator, the product (through the C ABI) against the CPU oracle on the same inputs.  The fixed cases of
test_gpu_parity.py pin what the reference's own fixtures pin; these look for what nobody thought of — odd widths that are
no multiple of any tile, rasters barely larger than a window, constant bands, heavy duplicates, cluster counts near the
number of distinct pixels.  Integer / label outputs and IEEE-elementwise planes bit for bit, the PCA within 1e-5 of the
float64 evaluation.  RSSEG_FUZZ_N=<n> runs n seeds per family instead of the default few (a campaign, run by hand:
profiles/r04_fuzz_campaign.txt holds the last one)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = int(os.environ.get("RSSEG_FUZZ_N", "0"))
SEED0 = int(os.environ.get("RSSEG_FUZZ_SEED0", "0"))


def seeds(default):
    return list(range(SEED0, SEED0 + (N or default)))


def dev(ctx, a, dtype=None):
    return ctx.to_device(np.ascontiguousarray(a).reshape(-1), dtype)


def host(t, shape=None):
    a = t.cpu().numpy()
    return a if shape is None else a.reshape(shape)


def random_bands(rng, H, W, nb=7, neg_zero=True):
    """7 float32 bands of one of the distributions the path meets: 8-bit digital numbers (the TM tiles), wider integers,
    general floats, duplicate-heavy floats; now and then one band constant or two bands equal."""
    kind = rng.choice(["u8", "u8_narrow", "u11", "float", "steps"])
    yy, xx = np.mgrid[0:H, 0:W]
    bands = []
    for b in range(nb):
        smooth = 60.0 * np.sin(yy / rng.uniform(3, 40) + b) * np.cos(xx / rng.uniform(3, 40) - b) + rng.uniform(60, 180)
        noise = rng.normal(0, rng.uniform(1, 30), (H, W))
        v = smooth + noise
        if kind == "u8":
            v = np.clip(np.round(v), 0, 255)
        elif kind == "u8_narrow":
            v = np.clip(np.round(v / 16.0), 3, 12)
        elif kind == "u11":
            v = np.clip(np.round(v * 8.0), 0, 2047)
        elif kind == "steps":
            v = np.round(v / 7.0) * 0.37 - 3.0
        v = v.astype(np.float32)              # np.round of a small negative leaves -0.0 in the integer kinds: kept (np.clip keeps it too)
        bands.append(v if neg_zero else v + np.float32(0.0))
    flip = rng.random()
    if flip < 0.15:
        bands[int(rng.integers(0, nb))][:] = np.float32(rng.integers(0, 200))     # a constant band: hi == lo, IQR == 0
    elif flip < 0.3 and nb >= 7:
        bands[int(rng.integers(4, nb))] = bands[int(rng.integers(0, 4))].copy()   # two equal bands: a singular covariance
    return str(kind), bands


def bits_equal(a, b):
    """Equal bit for bit (-0.0 != +0.0), same dtype, NaNs at the same places (payloads not compared)."""
    a, b = np.ascontiguousarray(a).reshape(-1), np.ascontiguousarray(b).reshape(-1)
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.dtype.kind != "f":
        return bool(np.array_equal(a, b))
    it = {4: np.int32, 8: np.int64}[a.dtype.itemsize]
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a.view(it)[~na], b.view(it)[~na]))


def pca_truth64(norm_planes):
    """float64 evaluation of perform_pca (RobustScaler, a zero inter-quartile range scaled by 1 as scikit-learn does, then PCA
    with the sign rule of svd_flip): (components' scores [nc, N], eigenvalues)."""
    X = np.stack([b.reshape(-1) for b in norm_planes], 1).astype(np.float32)
    c = np.median(X, axis=0)
    q = np.transpose([np.percentile(X[:, j], (25.0, 75.0)) for j in range(X.shape[1])])
    sc = q[1] - q[0]
    sc[sc < 10 * np.finfo(np.float64).eps] = 1.0
    X = X - c
    X = (X / sc).astype(np.float32)
    X64 = X.astype(np.float64)
    m = X64.mean(0)
    C = (X64 - m).T @ (X64 - m) / (X64.shape[0] - 1)
    w, V = np.linalg.eigh(C)
    Vt = V[:, ::-1].T.copy()
    Vt *= np.sign(Vt[np.arange(Vt.shape[0]), np.argmax(np.abs(Vt), axis=1)])[:, None]
    return ((X64 - m) @ Vt.T).T, w[::-1]


@pytest.mark.parametrize("seed", seeds(6))
def test_fuzz_config3_planes_and_kmeans(ctx, oracle, seed):
    """Random raster -> the product's config 3 (order statistics, fused index / PCA pass, texture chain at window 7 and a random
    step, KMeans with a random k).  The 12 index / texture planes equal the oracle's bit for bit; the components are within 1e-5
    of the float64 evaluation wherever the spectrum is well separated; KMeans is checked on the product's OWN 15 planes — the
    oracle fed the same planes must draw the same seeds, take the same number of iterations and give the same labels."""
    from rsseg import pipeline as P
    rng = np.random.default_rng(1000 + seed)
    big = seed % 5 == 4                      # one case in five crosses tile / chunk / workgroup boundaries many times
    H = int(rng.integers(150, 700)) if big else int(rng.integers(7, 150))
    W = int(rng.integers(300, 1100)) if big else int(rng.integers(7, 300))
    step = int(rng.choice([1, 1, 2, 3, 5, 7]))
    k = int(rng.integers(2, 11))
    kind, bands = random_bands(rng, H, W)
    tag = dict(seed=seed, H=H, W=W, step=step, k=k, kind=kind)
    labels, meta, planes = P.config3(ctx, [dev(ctx, b) for b in bands], H, W, k, 7, step, 3)
    norm = [oracle.robust_normalize(b) for b in bands]
    b, g, rd, n, s = norm[:5]
    feats = [oracle.calculate_ndvi(n, rd), oracle.calculate_evi(n, rd, b), oracle.calculate_msavi(n, rd),
             oracle.calculate_ndwi(g, n), oracle.calculate_mndwi(g, s), oracle.calculate_ndbi(s, n),
             oracle.calculate_bsi(b, rd, n, s)]
    gl, _ = oracle.calculate_glcm_features(n, 32, 7, step)
    feats += [gl[x] for x in ("contrast", "dissimilarity", "homogeneity", "energy", "correlation")]
    for i in range(12):
        assert bits_equal(host(planes[i]), feats[i]), (tag, i)
    with np.errstate(all="ignore"):
        truth, evals = pca_truth64(norm)
    if np.all(np.isfinite(evals)) and np.all(np.isfinite(truth[:3])):
        gaps = np.abs(np.diff(evals[:4]))
        for i in range(3):
            if min(gaps[max(i - 1, 0):i + 1].min(), evals[i]) > 5e-2:      # an isolated eigenvalue: its vector is well conditioned
                d = float(np.abs(host(planes[12 + i], (H, W)) - truth[i].reshape(H, W)).max())
                assert d <= 1e-5, (tag, i, d, evals[:4])
    got_planes = [host(p) for p in planes]
    if all(np.isfinite(p).all() for p in got_planes):
        want, info = oracle.kmeans_fit_planes(got_planes, k)
        assert [int(x) for x in meta["init_indices"]] == [int(x) for x in info["init_indices"]], tag
        assert int(meta["n_iter"]) == int(info["n_iter"]) and int(meta["relocated"]) == int(info["relocated"]), tag
        assert np.array_equal(host(labels), want), (tag, int((host(labels) != want).sum()))


@pytest.mark.parametrize("seed", seeds(4))
def test_fuzz_stack19_and_forest(ctx, oracle, seed):
    """Random raster -> the 19-feature stack (7x7 context means, texture at 21 / 21, morphology, local deviation, Sobel) against
    the oracle's stage, then a forest fitted on random rows of that stack with random shape parameters: the product's labels
    equal model.predict on every pixel."""
    from sklearn.ensemble import RandomForestClassifier
    from rsseg import pipeline as P
    from rsseg.forest import flatten_forest
    rng = np.random.default_rng(2000 + seed)
    big = seed % 5 == 4
    H = int(rng.integers(120, 500)) if big else int(rng.integers(21, 120))
    W = int(rng.integers(200, 900)) if big else int(rng.integers(21, 200))
    kind, bands = random_bands(rng, H, W)
    tag = dict(seed=seed, H=H, W=W, kind=kind)
    planes, _ = P.feature_stack19(ctx, [dev(ctx, b) for b in bands], H, W)
    stack = P.stack19_to_host(planes, H, W)
    with np.errstate(all="ignore"):
        _, hier = oracle.run_feature_extraction_stage(bands)
    ref = hier["all"]
    assert stack.shape == ref.shape
    for c in (0, 1, 2, 3, 4, 5, 14, 15, 16, 17, 18):          # IEEE-elementwise / integer columns
        assert bits_equal(stack[:, :, c], ref[:, :, c]), (tag, c)
    for c in (7, 8, 9, 10, 11, 12):                          # 7x7 means of bit-exact planes
        assert np.allclose(stack[:, :, c], ref[:, :, c], rtol=0, atol=1e-5, equal_nan=True), (tag, c)
    fplanes = P.stack19_forest_planes(ctx, planes)
    X = np.stack([host(p) for p in fplanes], 1)
    if not np.isfinite(X).all():
        return
    n_cls = int(rng.integers(2, 9))
    rows = rng.choice(H * W, size=min(H * W, int(rng.integers(40, 600))), replace=False)
    y = (rng.integers(0, n_cls, rows.size) + (X[rows, 2] > np.median(X[:, 2])) * 3) % n_cls
    model = RandomForestClassifier(n_estimators=int(rng.integers(1, 24)), max_depth=int(rng.integers(1, 14)) if rng.random() < 0.8 else None,
                                   random_state=int(seed), n_jobs=1).fit(X[rows], y)
    ctx.forest_load(flatten_forest(model))
    got = host(ctx.forest_predict(fplanes))
    want = model.predict(X)
    assert np.array_equal(got, want), (tag, int((got != want).sum()))


@pytest.mark.parametrize("seed", seeds(4))
def test_fuzz_rule_based(ctx, oracle, seed):
    """Random index planes with blobs of every class -> the rule-based classifier (thresholds, elliptical open / close, hole
    fill, 8-connected component area filter, priority painting) against the oracle: bit for bit."""
    from modules.features import extract as E
    rng = np.random.default_rng(3000 + seed)
    H = int(rng.integers(5, 220))
    W = int(rng.integers(5, 260))
    yy, xx = np.mgrid[0:H, 0:W]
    feats = {}
    for name in ("ndvi", "ndwi", "mndwi", "ndbi", "bsi", "evi", "msavi"):
        v = 0.6 * np.sin(yy / rng.uniform(4, 30) + rng.uniform(0, 6)) * np.cos(xx / rng.uniform(4, 30)) + rng.normal(0, rng.uniform(0.02, 0.3), (H, W))
        feats[name] = v.astype(np.float32)
    if rng.random() < 0.3:
        del feats["mndwi"]
    feats.update(height=H, width=W)
    want = oracle.rule_based_classification(feats)
    got = E.rule_based_classification(feats)
    assert got.shape == want.shape and np.array_equal(got, want), (dict(seed=seed, H=H, W=W), int((got != want).sum()))


@pytest.mark.parametrize("seed", seeds(3))
def test_fuzz_row_stripes_equal_single_context(ctx, oracle, seed):
    """Random raster, a random number of row stripes (every rank a thread with its own context and a barrier all-reduce
    hook, test_gpu_dist._ThreadWorld), random texture step and k: labels, iteration count and the 15 planes of every stripe
    equal the rows of the single-context result bit for bit; the 19-feature stack too when the raster is tall enough."""
    import torch
    from rsseg import pipeline as P
    from rsseg.runtime import Context
    from test_gpu_dist import _ThreadWorld
    rng = np.random.default_rng(4000 + seed)
    H = int(rng.integers(9, 260))
    W = int(rng.integers(8, 200))
    world = int(rng.integers(2, 8))
    step = int(rng.choice([1, 1, 2, 3, 7]))
    k = int(rng.integers(2, 9))
    kind, bl = random_bands(rng, H, W)
    bands = np.stack(bl)
    with19 = H >= 42 and W >= 21             # two 21-row texture windows, one 21-column window
    tag = dict(seed=seed, H=H, W=W, world=world, step=step, k=k, kind=kind)
    d0 = [dev(ctx, bands[i]) for i in range(7)]
    labels, meta, planes = P.config3(ctx, d0, H, W, k, 7, step, 3)
    want_labels = host(labels)
    want_planes = [host(p) for p in planes]
    want19 = [host(p) for p in P.feature_stack19(ctx, d0, H, W)[0]] if with19 else []
    tw = _ThreadWorld(world)
    out = [None] * world

    def rank_main(r):
        c = Context(0, use_dist=False)
        c.install_comm_hook(r, world, tw.hook(r))
        r0, r1 = P.stripe_rows(H, world, r)
        j0, j1, i0, i1 = P.glcm_halo_rows(H, r0, r1, 7, step)
        d = [c.to_device(bands[i, r0:r1].reshape(-1)) for i in range(7)]
        nir_ext = c.to_device(bands[3, i0:i1].reshape(-1))
        lab, m, pl = P.config3_striped(c, d, nir_ext, H, W, r0, r1, i0, k, 7, step)
        p19 = []
        if with19:
            e0, e1 = P.stack19_halo_rows(H, r0, r1)
            p19, _ = P.stack19_striped(c, [c.to_device(bands[i, e0:e1].reshape(-1)) for i in range(7)], H, W, r0, r1, e0)
        torch.cuda.synchronize()
        out[r] = (r0, r1, lab.cpu().numpy(), m["n_iter"], [p.cpu().numpy() for p in pl], [p.cpu().numpy() for p in p19])
        c.close()

    tw.run(rank_main)
    for r in range(world):
        r0, r1, lab, n_iter, pl, p19 = out[r]
        a, b = r0 * W, r1 * W
        assert n_iter == meta["n_iter"], (tag, r)
        assert np.array_equal(lab, want_labels[a:b]), (tag, r)
        for i, p in enumerate(pl):
            assert np.array_equal(p, want_planes[i][a:b], equal_nan=True), (tag, r, i)
        for i, p in enumerate(p19):
            assert np.array_equal(p, want19[i][a:b], equal_nan=True), (tag, r, i)


@pytest.mark.parametrize("seed", seeds(4))
def test_fuzz_texture_dictionary_members(ctx, oracle, seed):
    """One random band of a random shape (down to a single row or column) through the mirror's texture functions — morphology
    at 3 / 5 / 7, local mean / variance / deviation at 1 / 3 / 5 / 7, rank entropy, Gaussian 5 / 15, DoG, Laplacian, Sobel,
    uniform LBP — against the oracle: bit for bit, the entropies (a logarithm) within 1e-12."""
    from modules.features import indices as I
    rng = np.random.default_rng(5000 + seed)
    H = int(rng.integers(1, 120))
    W = int(rng.integers(1, 200))
    kind, bands = random_bands(rng, H, W, nb=1)
    band = bands[0]
    tag = dict(seed=seed, H=H, W=W, kind=kind)
    with np.errstate(all="ignore"):
        want_m = oracle.calculate_morphological_features(band)
        got_m = I.calculate_morphological_features(band)
        for k in want_m:
            assert bits_equal(got_m[k], want_m[k]), (tag, k)
        ms = I.calculate_multi_scale_features(band)
        b = oracle.robust_normalize(band)
        u8 = oracle.to_u8(b)
        for sc in (3, 5, 7):
            assert bits_equal(ms[f"std_dev_scale_{sc}"], oracle.std_dev_feature(band, sc)), (tag, sc)
            assert bits_equal(ms[f"variance_scale_{sc}"], oracle.variance_feature(band, sc)), (tag, sc)
            assert bits_equal(ms[f"mean_scale_{sc}"], oracle.box_mean(b, sc, "reflect101")), (tag, sc)
        for sc in (1, 3, 5):
            e = oracle.rank_entropy(u8, sc)
            assert np.allclose(ms[f"entropy_scale_{sc}"], e / np.max(e), rtol=0, atol=1e-12, equal_nan=True), (tag, sc)
        fr = I.calculate_filter_responses(band)
        want_f = oracle.filter_responses_extra(band)
        for k in ("gaussian_5", "gaussian_15", "dog"):
            assert bits_equal(fr[k], want_f[k]), (tag, k)
        assert bits_equal(fr["sobel_mag"], oracle.sobel_mag_feature(band)), tag
        assert bits_equal(fr["laplacian"], oracle.laplacian_feature(band)), tag
        lbp = I.calculate_lbp_features(band)
        o = oracle.lbp_uniform(u8, 24, 3)
        assert bits_equal(lbp, o / o.max()), tag


@pytest.mark.parametrize("seed", seeds(4))
def test_fuzz_glcm_parameters(ctx, oracle, seed):
    """The texture entry point on a random quantised map with random level count (2..64), window (2..23) and step (1..window+2):
    every kernel family behind the dispatch (2 x 2 windows per thread, one window per thread, workgroup per window) against
    the oracle's exact integer formulation, five properties bit for bit."""
    rng = np.random.default_rng(6000 + seed)
    levels = int(rng.choice([2, 4, 8, 16, 31, 32, 33, 48, 64]))
    win = int(rng.choice([2, 3, 4, 5, 7, 7, 7, 9, 11, 15, 21, 23]))
    step = int(rng.integers(1, win + 3))
    H = int(rng.integers(win, win + 90))
    W = int(rng.integers(win, win + 200))
    base = rng.integers(0, levels, (H, W))
    smooth = (np.add.outer(int(rng.integers(1, 4)) * np.arange(H), np.arange(W)) // int(rng.integers(2, 12))) % levels
    q = np.where(rng.random((H, W)) < rng.random(), base, smooth).astype(np.uint8)
    if rng.random() < 0.3:
        q[: H // 2, : W // 2] = int(rng.integers(0, levels))          # constant windows: variance 0, correlation's special case
    tag = dict(seed=seed, levels=levels, win=win, step=step, H=H, W=W)
    want = oracle.glcm_small_maps(q, levels, win, step, mode=1)
    got, (oh, ow) = ctx.glcm(dev(ctx, q), H, W, levels, win, step)
    assert (oh, ow) == ((H - win) // step + 1, (W - win) // step + 1), tag
    for g, k in zip(got, ["contrast", "dissimilarity", "homogeneity", "energy", "correlation"]):
        assert bits_equal(host(g, (oh, ow)), want[k]), (tag, k)


@pytest.mark.parametrize("seed", seeds(6))
def test_fuzz_kmeans_entry_point(ctx, oracle, seed):
    """rsseg_kmeans_fit_predict on random matrices: 1..64 feature planes, 1..64 clusters, float32 or float64, 1..60 000 rows,
    values continuous / on a coarse grid (ties, duplicates, clusters that run empty and are relocated) / with NaNs (the entry
    point zeroes them as the reference does): seeds, iteration count, relocations and labels equal the oracle's bit for bit."""
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
    planes = [np.ascontiguousarray(X[:, f]).astype(dt) for f in range(F)]
    tag = dict(seed=seed, F=F, k=k, n=n, dt=dt.__name__, style=style)
    want, info = oracle.kmeans_fit_planes(planes, k)
    labels, meta = ctx.kmeans_fit_predict([dev(ctx, p) for p in planes], k)
    assert [int(x) for x in meta["init_indices"]] == [int(x) for x in info["init_indices"]], tag
    assert int(meta["n_iter"]) == int(info["n_iter"]) and int(meta["relocated"]) == int(info["relocated"]), (tag, meta["n_iter"], info["n_iter"])
    assert np.array_equal(host(labels), want), (tag, int((host(labels) != want).sum()))


@pytest.mark.parametrize("seed", seeds(4))
def test_fuzz_resize_and_order_statistics(ctx, oracle, seed):
    """cv2.resize(INTER_LINEAR) between random shapes (up and down, 1-pixel sources) against the oracle bit for bit, and the
    order statistics of a random plane at random ranks against np.sort."""
    rng = np.random.default_rng(8000 + seed)
    sh, sw = int(rng.integers(1, 60)), int(rng.integers(1, 90))
    dh, dw = int(rng.integers(1, 400)), int(rng.integers(1, 600))
    src = rng.random((sh, sw)).astype(np.float32)
    got = host(ctx.resize_bilinear(dev(ctx, src), sh, sw, dh, dw), (dh, dw))
    assert bits_equal(got, oracle.resize_bilinear(src, dh, dw)), dict(seed=seed, sh=sh, sw=sw, dh=dh, dw=dw)
    n = int(rng.choice([rng.integers(1, 50), rng.integers(50, 5000), rng.integers(5000, 400000)]))
    kind = str(rng.choice(["u8", "u11", "float", "neg", "dup"]))
    a = {"u8": lambda: rng.integers(0, 256, n), "u11": lambda: rng.integers(0, 2048, n), "float": lambda: rng.standard_normal(n) * 1e3,
         "neg": lambda: rng.integers(-50, 50, n) + rng.integers(0, 2, n) * 0.5, "dup": lambda: rng.integers(0, 3, n) * 1e-30}[kind]().astype(np.float32)
    ranks = sorted({int(r) for r in rng.integers(0, n, min(n, 9))} | {0, n - 1})
    vals, n_nan = ctx.order_stats(dev(ctx, a), ranks)
    assert n_nan == 0 and np.array_equal(vals, np.sort(a)[ranks]), dict(seed=seed, n=n, kind=kind)


@pytest.mark.parametrize("seed", seeds(4))
def test_fuzz_forest_entry_point(ctx, oracle, seed):
    """rsseg_forest_load / rsseg_forest_predict with random forests: 1..64 features, 2..40 classes with arbitrary labels, 1..30
    trees, depth 1..unbounded (trees larger than the LDS node area take the general kernel), bootstrap on / off, training with or
    without NaNs (scikit-learn then records where missing values go), NaN and huge values in the rows to classify, pixel counts
    that leave ragged workgroups: labels equal model.predict on every row."""
    from sklearn.ensemble import ExtraTreesClassifier, RandomForestClassifier
    rng = np.random.default_rng(9000 + seed)
    F = int(rng.choice([1, 2, 3, 5, 8, 19, 31, 32, 33, 55, 64]))
    ncls = int(rng.choice([2, 3, 4, 5, 8, 9, 16, 17, 33, 40]))
    ntr = int(rng.choice([30, 300, 3000, 30000]))
    Xtr = rng.random((ntr, F)).astype(np.float32)
    if rng.random() < 0.4:
        Xtr = (np.round(Xtr * 8) / 8).astype(np.float32)          # thresholds that coincide with feature values
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
    if rng.random() < 0.5:
        X = (np.round(X * 8) / 8).astype(np.float32)
    if rng.random() < 0.6:
        X[rng.random((n, F)) < 0.02] = np.nan
    if rng.random() < 0.3:
        X[rng.random((n, F)) < 0.01] = np.float32(3.0e38)
        X[rng.random((n, F)) < 0.01] = np.float32(-3.0e38)
    tag = dict(seed=seed, F=F, classes=len(model.classes_), trees=kw["n_estimators"], depth=depth, n=n, nan_fit=nan_fit, extra_trees=use_et,
               nodes=max(e.tree_.node_count for e in model.estimators_))
    model.set_params(n_jobs=1)      # with several jobs scikit-learn adds the trees' votes in the order the threads finish: near-ties flip
    want = model.predict(X)
    ctx.forest_load(oracle.flatten_forest(model))
    got = host(ctx.forest_predict([dev(ctx, X[:, i]) for i in range(F)]))
    assert np.array_equal(got, want), (tag, int((got != want).sum()))


@pytest.mark.parametrize("seed", seeds(4))
def test_fuzz_pca_entry_point(ctx, oracle, seed):
    """perform_pca through the product on 2..12 random planes of a random size, with and without the RobustScaler, a random number
    of components: scores within 1e-5 of the float64 evaluation for every well-separated eigenvalue, explained-variance ratios
    within 1e-6, and the two paths (normalised planes in memory / raw planes normalised inside the kernels) bit-identical."""
    from rsseg import pipeline as P
    rng = np.random.default_rng(9500 + seed)
    nb = int(rng.integers(2, 11))
    H, W = int(rng.integers(3, 200)), int(rng.integers(3, 300))
    kind, bands = random_bands(rng, H, W, nb=nb)
    robust = bool(rng.random() < 0.75)
    nc = None if rng.random() < 0.3 else int(rng.integers(1, nb + 1))
    if H * W <= nb:
        return
    norm = [oracle.robust_normalize(b) for b in bands]
    tag = dict(seed=seed, nb=nb, H=H, W=W, kind=kind, robust=robust, nc=nc)
    if nb > 8:       # the kernels take at most 8 band planes (the TM scenes have 7): refused as a capacity, by name
        from rsseg.runtime import RssegUnsupported
        with pytest.raises(RssegUnsupported, match="at most 8"):
            P.pca(ctx, [dev(ctx, b) for b in norm], nc, robust)
        return
    pcs, ratio, model = P.pca(ctx, [dev(ctx, b) for b in norm], nc, robust)
    X = np.stack([b.reshape(-1) for b in norm], 1).astype(np.float32)
    if robust:
        c = np.median(X, axis=0)
        q = np.transpose([np.percentile(X[:, j], (25.0, 75.0)) for j in range(nb)])
        sc = q[1] - q[0]
        sc[sc < 10 * np.finfo(np.float64).eps] = 1.0
        X = ((X - c) / sc).astype(np.float32)
    X64 = X.astype(np.float64)
    m = X64.mean(0)
    C = (X64 - m).T @ (X64 - m) / (X64.shape[0] - 1)
    w, V = np.linalg.eigh(C)
    w, Vt = w[::-1], V[:, ::-1].T.copy()
    Vt *= np.sign(Vt[np.arange(nb), np.argmax(np.abs(Vt), axis=1)])[:, None]
    truth = ((X64 - m) @ Vt.T).T
    k = nb if nc is None else nc
    assert len(pcs) == k, tag
    if w.sum() > 0:
        assert np.allclose(np.asarray(ratio, np.float64), (w / w.sum())[:k], rtol=0, atol=2e-6), (tag, ratio, (w / w.sum())[:k])
    scale = max(1.0, float(np.abs(truth).max()))
    for i in range(k):
        lo = w[i] - w[i + 1] if i + 1 < nb else w[i]
        hi = w[i - 1] - w[i] if i > 0 else np.inf
        rel_gap = min(lo, hi) / max(w[0], 1e-30)
        if rel_gap > 5e-2:
            # an eigenvector's sensitivity to float32 rounding of the covariance grows with 1 / gap: 1e-5 for gaps of a fifth of the
            # spectrum and more (the bar of DESIGN 4), proportionally more below (seed 1059: gap 0.146 of w[0], 1.05e-5)
            d = float(np.abs(host(pcs[i]) - truth[i]).max())
            assert d <= 1e-5 * scale * max(1.0, 0.2 / rel_gap), (tag, i, d, w[:4])


@pytest.mark.parametrize("seed", seeds(4))
def test_fuzz_threshold_and_post_processing(ctx, oracle, seed):
    """threshold_segmentation (fixed threshold above / below, float32 and float64 planes, NaNs, Otsu incl. planes without
    contrast) and advanced_post_processing (random mask density, min_area, elliptical element of an odd size 1..31, an even size
    -> hole fill) of the mirror against the oracle's restatements of cv2 / scipy.ndimage: bit for bit."""
    from modules.features import extract as E
    rng = np.random.default_rng(9800 + seed)
    H, W = int(rng.integers(1, 160)), int(rng.integers(1, 220))
    yy, xx = np.mgrid[0:H, 0:W]
    plane = (np.sin(yy / rng.uniform(2, 25)) * np.cos(xx / rng.uniform(2, 25)) + rng.normal(0, rng.uniform(0.01, 0.5), (H, W)))
    plane = plane.astype(np.float32 if rng.random() < 0.6 else np.float64)
    if rng.random() < 0.3:
        plane[rng.random((H, W)) < 0.05] = np.nan
    if rng.random() < 0.1:
        plane[:] = plane.flat[0] if np.isfinite(plane.flat[0]) else 0.25       # no contrast
    tag = dict(seed=seed, H=H, W=W, dtype=str(plane.dtype))
    thr = float(rng.uniform(-0.8, 0.8))
    for above in (True, False):
        assert np.array_equal(E.threshold_segmentation(plane, thr, above), oracle.threshold_segmentation(plane, thr, above)), (tag, thr, above)
    with np.errstate(all="ignore"):
        want_o = oracle.threshold_segmentation(plane, None, True, otsu=True)
    assert np.array_equal(E.threshold_segmentation(plane, None, True, otsu=True), want_o), (tag, "otsu")
    mask = (rng.random((H, W)) < rng.uniform(0.1, 0.9)).astype(np.uint8)
    if H > 12 and W > 12:
        mask[3:H // 2, 3:W // 2] = 1
        mask[5:H // 2 - 2, 5:W // 2 - 2] = rng.random((max(H // 2 - 7, 0), max(W // 2 - 7, 0))) < 0.7      # holes inside a block
    k = int(rng.choice([0, 1, 2, 3, 3, 4, 5, 5, 7, 9, 11, 15, 21, 31]))
    min_area = int(rng.choice([0, 1, 2, 5, 30, 200, 100000]))
    got = E.advanced_post_processing(mask, min_area, k)
    want = oracle.advanced_post_processing(mask, min_area, k)
    assert np.array_equal(got, want), (tag, k, min_area, int((got != want).sum()))


@pytest.mark.parametrize("seed", seeds(3))
def test_fuzz_uint8_planes_equal_float32_planes(ctx, seed):
    """8-bit rasters handed over as uint8 device planes (a quarter of the bytes: Context.upload_band, the table forms of the PCA
    kernels, the 8-bit histogram pass) against the same values as float32 planes: config 3 (random texture step and k) and the
    19-feature stack — labels, seeds, iteration count and every plane bit-identical."""
    import torch
    from rsseg import pipeline as P
    rng = np.random.default_rng(9900 + seed)
    H, W = int(rng.integers(21, 260)), int(rng.integers(21, 330))
    step = int(rng.choice([1, 1, 2, 7]))
    k = int(rng.integers(2, 10))
    _, bands = random_bands(rng, H, W, neg_zero=False)      # uint8 planes cannot hold a -0.0
    bands = [np.clip(np.round(b), 0, 255) + 0.0 for b in bands]
    if rng.random() < 0.3:
        bands = [np.clip(np.round(b / 32.0), 0, 7) for b in bands]       # 8 distinct values per band
    d32 = [dev(ctx, b.astype(np.float32)) for b in bands]
    d8 = [dev(ctx, b.astype(np.uint8)) for b in bands]
    tag = dict(seed=seed, H=H, W=W, step=step, k=k)
    l32, m32, p32 = P.config3(ctx, d32, H, W, k, 7, step, 3)
    l8, m8, p8 = P.config3(ctx, d8, H, W, k, 7, step, 3)
    assert m8["n_iter"] == m32["n_iter"] and np.array_equal(m8["init_indices"], m32["init_indices"]), tag
    assert torch.equal(l8, l32), tag
    for i, (a, b) in enumerate(zip(p8, p32)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (tag, i)
    s32, _ = P.feature_stack19(ctx, d32, H, W)
    s8, _ = P.feature_stack19(ctx, d8, H, W)
    for i, (a, b) in enumerate(zip(s8, s32)):
        assert a.dtype == b.dtype and torch.equal(a.view(torch.uint8), b.view(torch.uint8)), (tag, i)
