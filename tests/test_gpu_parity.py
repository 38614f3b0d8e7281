"""GPU suite: every HIP kernel, called through the C ABI (ctypes), against the CPU oracle on the same
seeded inputs, against the golden vectors of the reference, and — at sizes the oracle cannot reach —
through size-independent properties.  Integer / label outputs must be bit-exact; float features must be
bit-exact where the arithmetic is IEEE-elementwise and within 1e-5 where the reference itself goes
through BLAS / LAPACK (PCA)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ["ndvi", "evi", "msavi", "ndwi", "mndwi", "ndbi", "bsi"]


@pytest.fixture(scope="module")
def crop(golden_dir):
    return np.load(os.path.join(golden_dir, "crop96.npz"))


@pytest.fixture(scope="module")
def scene(golden_dir):
    return np.load(os.path.join(golden_dir, "scene_aa.npz"))


def dev(ctx, a, dtype=None):
    return ctx.to_device(np.ascontiguousarray(a).reshape(-1), dtype)


def host(t, shape=None):
    a = t.cpu().numpy()
    return a if shape is None else a.reshape(shape)


# ------------------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("n", [1, 5, 1000, 9216, 360000, 1 << 20])
def test_order_stats_exact(ctx, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 256, n).astype(np.float32)
    frac = rng.random(n) < 0.2
    a[frac] += rng.random(int(frac.sum())).astype(np.float32)
    a = (a - 100.0).astype(np.float32)  # negative values, zeros, duplicates
    s = np.sort(a)
    ranks = sorted({0, n - 1, n // 2, int(0.02 * (n - 1)), int(0.98 * (n - 1)), min(n - 1, int(0.02 * (n - 1)) + 1)})
    vals, n_nan = ctx.order_stats(dev(ctx, a), ranks)
    assert n_nan == 0
    assert np.array_equal(vals, s[ranks])


def test_order_stats_general_floats_and_nan(ctx):
    rng = np.random.default_rng(3)
    a = (rng.standard_normal(200003) * 1e3).astype(np.float32)
    a[::1001] = np.nan
    a[5] = -0.0
    a[6] = 0.0
    a[7] = np.inf
    a[8] = -np.inf
    s = np.sort(a)  # NaN last
    nn = int(np.isnan(a).sum())
    ranks = [0, 1, 17, 100001, a.size - nn - 1, a.size - nn, a.size - 1]
    vals, n_nan = ctx.order_stats(dev(ctx, a), ranks)
    assert n_nan == nn
    assert np.array_equal(vals, s[ranks], equal_nan=True)
    with pytest.raises(ValueError):
        ctx.order_stats(dev(ctx, a), [a.size])


def test_order_stats_small_integer_fast_path_and_fallback(ctx):
    """Planes of small non-negative integers (the reference's uint8 digital numbers stored as float32) resolve in one pass;
    any other value — a fraction, a negative, 2048 or more, an infinity — anywhere in the plane sends the call through the
    three radix passes.  Both routes against np.sort, with NaNs (sorted last) and -0.0."""
    rng = np.random.default_rng(11)
    n = 300007
    base = rng.integers(0, 256, n).astype(np.float32)
    base[5] = -0.0
    ranks = [0, 1, n // 50, n // 2, n - n // 50, n - 1]
    cases = {"ints": base.copy()}
    for name, (pos, val) in {"fraction_late": (n - 3, 7.5), "negative": (n // 3, -1.0), "too_big": (17, 2048.0), "inf": (n // 2, np.inf),
                             "max_ok": (9, 2047.0)}.items():
        a = base.copy()
        a[pos] = val
        cases[name] = a
    with_nan = base.copy()
    with_nan[::5003] = np.nan
    cases["ints_with_nan"] = with_nan
    for name, a in cases.items():
        s = np.sort(a)
        vals, n_nan = ctx.order_stats(dev(ctx, a), ranks)
        assert n_nan == int(np.isnan(a).sum()), name
        assert np.array_equal(vals, s[ranks], equal_nan=True), name
    planes = [cases["ints"], cases["fraction_late"], cases["ints_with_nan"]]       # a group with one general plane: all fall back
    vals, nans = ctx.order_stats_multi([dev(ctx, p) for p in planes], [ranks] * 3)
    for i, p in enumerate(planes):
        assert np.array_equal(vals[i], np.sort(p)[ranks], equal_nan=True), i


def test_band_percentiles_equal_numpy(ctx, scene, oracle):
    from rsseg.quantiles import band_percentiles, robust_scaler_stats
    bands = oracle.stage1_preprocess(scene["dn"])
    for b in bands[:3]:
        lo, hi = band_percentiles(ctx, dev(ctx, b), (2, 98))
        assert lo == np.percentile(b, 2) and hi == np.percentile(b, 98)
        nb = oracle.robust_normalize(b)
        c, s = robust_scaler_stats(ctx, dev(ctx, nb))
        assert c == np.nanmedian(nb)
        q = np.nanpercentile(nb, (25.0, 75.0))
        assert s == q[1] - q[0]


# ------------------------------------------------------------------------------------------------ K2
def test_indices_bitexact_vs_reference_goldens(ctx, crop):
    from rsseg import pipeline as P
    bands = [dev(ctx, b) for b in crop["bands"]]
    lohi = P.band_lohi(ctx, bands)
    idx, norms = P.spectral_indices(ctx, bands, lohi, want_norm=(True,) * 5)
    for i in range(5):
        assert np.array_equal(host(norms[i], (96, 96)), crop["norm"][i])
    for k in KEYS:
        assert np.array_equal(host(idx[k], (96, 96)), crop["idx_" + k]), k
    for i in range(7):
        got = host(ctx.normalize(bands[i], float(lohi[i, 0]), float(lohi[i, 1])), (96, 96))
        assert np.array_equal(got, crop["norm"][i])


def test_indices_edge_cases_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(11)
    n = 4099  # ragged tail (n % 4 != 0)
    b = [rng.random(n).astype(np.float32) for _ in range(5)]
    b[3][:50] = 0.0
    b[2][:50] = 0.0          # den == 0
    b[1][60:70] = np.nan
    b[4][80:90] = 0.0004     # den just under 0.001
    b[3][80:90] = 0.0005
    idx, _ = ctx.spectral_indices([dev(ctx, x) for x in b], None)
    blue, green, red, nir, swir = b
    want = [oracle.calculate_ndvi(nir, red), oracle.calculate_evi(nir, red, blue), oracle.calculate_msavi(nir, red),
            oracle.calculate_ndwi(green, nir), oracle.calculate_mndwi(green, swir), oracle.calculate_ndbi(swir, nir),
            oracle.calculate_bsi(blue, red, nir, swir)]
    for g, w, k in zip(idx, want, KEYS):
        assert np.array_equal(host(g), w, equal_nan=True), k


def test_quantize_truncates_like_astype_uint8(ctx):
    x = np.linspace(0, 1, 100003, dtype=np.float32)
    for mult in (31.0, 255.0):
        q = host(ctx.quantize_u8(dev(ctx, x), mult))
        assert np.array_equal(q, (x * np.float32(mult)).astype(np.uint8))


# ------------------------------------------------------------------------------------------------ K3
def _pca_truth64(norm_planes):
    """float64 evaluation of the same estimator, used to rank two float32 results."""
    X = np.stack([b.reshape(-1) for b in norm_planes], 1).astype(np.float32)
    c = np.nanmedian(X, axis=0)
    q = np.transpose([np.nanpercentile(X[:, j], (25.0, 75.0)) for j in range(X.shape[1])])
    X = X - c
    X = (X / (q[1] - q[0])).astype(np.float32)
    X64 = X.astype(np.float64)
    m = X64.mean(0)
    C = (X64 - m).T @ (X64 - m) / (X64.shape[0] - 1)
    w, V = np.linalg.eigh(C)
    Vt = V[:, ::-1].T.copy()
    Vt *= np.sign(Vt[np.arange(Vt.shape[0]), np.argmax(np.abs(Vt), axis=1)])[:, None]
    return ((X64 - m) @ Vt.T).T, w[::-1]


def test_pca_vs_reference_goldens(ctx, crop):
    """Within 1e-5 of the reference (sklearn float32 sgemm + LAPACK) on every well-conditioned component.
    The last two eigenvalues of this crop are 2.27e-2 and 2.13e-2: their eigenvectors amplify float32
    covariance noise by 1/gap, so sklearn itself is ~9e-5 away from the float64 answer there; for those
    the check is that the GPU result is closer to float64 than the reference is, and within 2e-4 of it."""
    from rsseg import pipeline as P
    norm = [dev(ctx, b) for b in crop["norm"]]
    pcs, ratio, model = P.pca(ctx, norm, None, True)
    got = np.stack([host(p, (96, 96)) for p in pcs])
    truth, evals = _pca_truth64(list(crop["norm"]))
    truth = truth.reshape(7, 96, 96)
    gaps = np.minimum(np.abs(np.diff(evals, prepend=np.inf)), np.abs(np.diff(evals, append=-np.inf)))
    for c in range(7):
        d_ref = np.abs(got[c] - crop["pca7"][c]).max()
        if gaps[c] > 1e-2:
            assert d_ref <= 1e-5, (c, d_ref)
        else:
            assert d_ref <= 2e-4, (c, d_ref)
            assert np.abs(got[c] - truth[c]).max() <= np.abs(crop["pca7"][c] - truth[c]).max(), c
    assert np.allclose(ratio, crop["pca7_ratio"], atol=1e-6)
    assert np.allclose(model["components"][:5], crop["pca7_components"][:5], atol=1e-5)
    pcs3, ratio3, _ = P.pca(ctx, norm, 3, True)
    got3 = np.stack([host(p, (96, 96)) for p in pcs3])
    assert np.allclose(got3, crop["pca3"], rtol=0, atol=1e-5)
    assert np.allclose(ratio3, crop["pca3_ratio"], atol=1e-6)


@pytest.mark.parametrize("nb", [1, 2, 3, 5, 8])
def test_pca_other_band_counts_vs_float64(ctx, crop, nb):
    """The Gram / projection kernels are specialised on the band count: every instantiation other than 7 against the
    float64 evaluation of the same estimator (components with a clear eigenvalue gap: 1e-5; all: variance ratios)."""
    from rsseg import pipeline as P
    rng = np.random.default_rng(nb)
    planes = [crop["norm"][i % 7] if i < 7 else (crop["norm"][0] * 0.5 + rng.random((96, 96), dtype=np.float32) * 0.5).astype(np.float32)
              for i in range(nb)]
    pcs, ratio, model = P.pca(ctx, [dev(ctx, b) for b in planes], None, True)
    truth, evals = _pca_truth64(planes)
    assert len(pcs) == nb and np.isclose(ratio.sum(), 1.0, atol=1e-5)
    assert np.allclose(ratio, evals / evals.sum(), atol=1e-5)
    gaps = np.minimum(np.abs(np.diff(evals, prepend=np.inf)), np.abs(np.diff(evals, append=-np.inf)))
    for c in range(nb):
        if nb == 1 or gaps[c] > 5e-2:
            assert np.abs(host(pcs[c], (96, 96)).reshape(-1) - truth[c]).max() <= 1e-5, (nb, c)


def test_pca_full_scene_vs_reference(ctx, scene, oracle, golden_dir):
    """600 x 600 scene.  At N = 360 000 the reference's float32 mean / sgemm accumulation is itself ~5e-5
    away from the float64 answer, so the 1e-5 bar is checked against float64 and the reference's own
    output is required to be no closer to float64 than the GPU result is."""
    from rsseg import pipeline as P
    ref = np.load(os.path.join(golden_dir, "scene_aa_ref_outputs.npz"))
    bands = oracle.stage1_preprocess(scene["dn"])
    normh = [oracle.robust_normalize(b) for b in bands]
    norm = [dev(ctx, b) for b in normh]
    pcs, ratio, model = P.pca(ctx, norm, None, True)
    assert np.allclose(ratio, ref["pca_ratio"], atol=1e-5)
    assert np.allclose(model["components"], ref["pca_components"], atol=1e-4)
    truth, _ = _pca_truth64(normh)
    t0 = truth[0].reshape(600, 600)[::7, ::7]
    pc0 = host(pcs[0], (600, 600))[::7, ::7]
    d_gpu, d_ref = np.abs(pc0 - t0).max(), np.abs(ref["pc0_sample"] - t0).max()
    assert d_gpu <= 1e-5, d_gpu
    assert d_gpu <= d_ref
    assert np.abs(pc0 - ref["pc0_sample"]).max() <= 2e-4


# ------------------------------------------------------------------------------------------------ K4-K8
@pytest.mark.parametrize("win,step", [(7, 1), (7, 7), (5, 2), (3, 1), (21, 21), (9, 4)])
def test_glcm_bitexact_vs_oracle(ctx, oracle, win, step):
    rng = np.random.default_rng(win * 100 + step)
    H, W = 70, 131
    base = rng.integers(0, 32, (H, W))
    smooth = (np.add.outer(np.arange(H), np.arange(W)) // 9) % 32
    q = np.where(rng.random((H, W)) < 0.5, base, smooth).astype(np.uint8)
    q[:12, :12] = 7  # constant windows
    want = oracle.glcm_small_maps(q, 32, win, step, mode=1)
    got, (oh, ow) = ctx.glcm(dev(ctx, q), H, W, 32, win, step)
    for g, k in zip(got, ["contrast", "dissimilarity", "homogeneity", "energy", "correlation"]):
        assert np.array_equal(host(g, (oh, ow)), want[k]), (k, win, step)


@pytest.mark.parametrize("levels,win,step", [(64, 7, 1), (64, 5, 3), (48, 3, 1), (16, 7, 2), (64, 11, 5)])
def test_glcm_other_level_counts_bitexact_vs_oracle(ctx, oracle, levels, win, step):
    """Level counts other than 32: up to 32 levels the window kernel pre-scales the pixels by 8, up to 64 by 4 (a second
    instantiation with its own key layout and Hq table); larger windows take the workgroup-per-window kernel."""
    rng = np.random.default_rng(levels * 1000 + win * 10 + step)
    H, W = 61, 97
    base = rng.integers(0, levels, (H, W))
    smooth = (np.add.outer(3 * np.arange(H), np.arange(W)) // 7) % levels
    q = np.where(rng.random((H, W)) < 0.5, base, smooth).astype(np.uint8)
    q[30:45, 40:70] = levels - 1
    want = oracle.glcm_small_maps(q, levels, win, step, mode=1)
    got, (oh, ow) = ctx.glcm(dev(ctx, q), H, W, levels, win, step)
    for g, k in zip(got, ["contrast", "dissimilarity", "homogeneity", "energy", "correlation"]):
        assert np.array_equal(host(g, (oh, ow)), want[k]), (k, levels, win, step)


@pytest.fixture(params=["quad", "pair"])
def dense_kernel(request):
    """Dense 7x7 / step 1 / 32 levels: k4_glcm_quad (a 2 x 2 block of windows per thread, r04, the default) and
    k4_glcm_pair (two adjacent windows per thread, r02; RSSEG_GLCM_DENSE=pair) — both must equal the oracle bit for bit."""
    old = os.environ.get("RSSEG_GLCM_DENSE")
    os.environ["RSSEG_GLCM_DENSE"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("RSSEG_GLCM_DENSE", None)
    else:
        os.environ["RSSEG_GLCM_DENSE"] = old


def test_glcm_dense_pair_kernel_edge_maps(ctx, oracle, dense_kernel):
    """The dense kernels on maps of width / height 1, 2, 3 (threads with fewer windows than their block), exactly one
    workgroup strip (128 windows wide, 8 rows tall), one strip + 1 and + 2, single rows and columns, odd and even sizes."""
    rng = np.random.default_rng(5)
    for H, W in ((7, 7), (7, 8), (8, 7), (8, 8), (8, 9), (9, 134), (7, 135), (12, 136), (11, 20), (14, 21), (15, 263), (16, 13), (23, 7)):
        q = rng.integers(0, 32, (H, W)).astype(np.uint8)
        q[:, : W // 2] = (q[:, : W // 2] // 8) * 8   # few distinct levels on the left: many equal keys
        want = oracle.glcm_small_maps(q, 32, 7, 1, mode=1)
        got, (oh, ow) = ctx.glcm(dev(ctx, q), H, W, 32, 7, 1)
        assert (oh, ow) == (H - 6, W - 6)
        for g, k in zip(got, ["contrast", "dissimilarity", "homogeneity", "energy", "correlation"]):
            assert np.array_equal(host(g, (oh, ow)), want[k]), (k, H, W)


def test_glcm_dense_pair_kernel_bitexact_vs_oracle(ctx, oracle, dense_kernel):
    """Dense 7x7 / step 1 / 32 levels on a map wider than one workgroup strip (128 windows) and taller than one
    workgroup (4 / 8 rows): the dense kernels against the oracle, all five properties bit for bit."""
    rng = np.random.default_rng(77)
    H, W = 150, 300
    base = rng.integers(0, 32, (H, W))
    smooth = (np.add.outer(np.arange(H), 2 * np.arange(W)) // 11) % 32
    q = np.where(rng.random((H, W)) < 0.4, base, smooth).astype(np.uint8)
    q[40:70, 100:160] = 31  # flat region: every pair on one diagonal bin (largest counter values)
    q[90:110, :] = (np.arange(W) % 2 * 31).astype(np.uint8)  # two-level stripes
    want = oracle.glcm_small_maps(q, 32, 7, 1, mode=1)
    got, (oh, ow) = ctx.glcm(dev(ctx, q), H, W, 32, 7, 1)
    assert (oh, ow) == (144, 294)
    for g, k in zip(got, ["contrast", "dissimilarity", "homogeneity", "energy", "correlation"]):
        assert np.array_equal(host(g, (oh, ow)), want[k]), k


def test_glcm_vs_skimage_docstring_image_and_literal_restatement(ctx, oracle):
    """(a) The 4 x 4 example image of scikit-image's graycomatrix / graycoprops docstrings through the HIP library
    (window 4: the workgroup-per-window kernel; levels 4): the mean contrast over the four angles equals the value that
    follows from the published co-occurrence matrices, (7/12 + 16/9 + 1 + 4/9) / 4, and all five properties equal the
    literal (graycoprops-formula) restatement, oracle mode 0, which tests/test_oracle.py pins to those vectors.
    (b) The register kernels (windows 7 / 5 / 3) and the workgroup kernel against oracle MODE 0 within 1e-6 on all five
    properties — the HIP path's exact integer formulation against the literal float64 formulas."""
    img = np.array([[0, 0, 1, 1], [0, 0, 1, 1], [0, 2, 2, 2], [2, 2, 3, 3]], np.uint8)
    got, (oh, ow) = ctx.glcm(dev(ctx, img), 4, 4, 4, 4, 1)
    assert (oh, ow) == (1, 1)
    lit = oracle.glcm_small_maps(img, 4, 4, 1, mode=0)
    names = ["contrast", "dissimilarity", "homogeneity", "energy", "correlation"]
    for g, k in zip(got, names):
        assert abs(float(host(g)[0]) - float(lit[k][0, 0])) <= 1e-6, k
    assert abs(float(host(got[0])[0]) - (7 / 12 + 16 / 9 + 1 + 4 / 9) / 4) <= 1e-6
    rng = np.random.default_rng(2026)
    H, W = 64, 90
    base = rng.integers(0, 32, (H, W))
    smooth = (np.add.outer(np.arange(H), np.arange(W)) // 5) % 32
    q = np.where(rng.random((H, W)) < 0.5, base, smooth).astype(np.uint8)
    q[:9, :9] = 3
    for win, step in ((7, 1), (5, 2), (3, 1), (21, 21), (9, 3)):
        lit = oracle.glcm_small_maps(q, 32, win, step, mode=0)
        got, (oh, ow) = ctx.glcm(dev(ctx, q), H, W, 32, win, step)
        for g, k in zip(got, names):
            assert np.allclose(host(g, (oh, ow)), lit[k], rtol=1e-6, atol=1e-6), (k, win, step)


def test_resize_bilinear_bitexact_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(5)
    # up-sampling by 21 (config 5's maps), by ~1 (config 3's), mixed, a 1 x 1 source, the identity; r04 (the kernel walks
    # SOURCE rows, several destination rows per source row or none): down-sampling by 6 / 4.4 / 1.003, a source row count
    # below the look-ahead, and destination strips taller than one workgroup's share
    for (sh, sw, dh, dw) in [(28, 28, 600, 600), (13, 7, 40, 55), (594, 594, 600, 600), (1, 1, 8, 8), (5, 9, 5, 9),
                             (600, 600, 100, 90), (57, 33, 13, 40), (300, 300, 299, 301), (40, 40, 2000, 300), (3, 500, 700, 20),
                             (1000, 16, 3, 16), (2, 2, 1, 1)]:
        src = rng.random((sh, sw)).astype(np.float32)
        got = host(ctx.resize_bilinear(dev(ctx, src), sh, sw, dh, dw), (dh, dw))
        assert np.array_equal(got, oracle.resize_bilinear(src, dh, dw)), (sh, sw, dh, dw)


def test_window_ops_bitexact_vs_oracle(ctx, oracle):
    from rsseg import _lib as L
    rng = np.random.default_rng(9)
    H, W = 67, 201
    x = rng.random((H, W)).astype(np.float32)
    x[10:20, 10:20] = 0.5
    for k, border, name in [(7, L.BORDER_REFLECT, "reflect"), (5, L.BORDER_REFLECT101, "reflect101"), (3, L.BORDER_REFLECT, "reflect")]:
        got = host(ctx.box_mean(dev(ctx, x), H, W, k, border), (H, W))
        assert np.array_equal(got, oracle.box_mean(x, k, name)), (k, name)
    mean = oracle.box_mean(x, 5, "reflect101")
    msq = oracle.box_mean(x * x, 5, "reflect101")
    var = msq - mean * mean
    var[var < 0] = 0
    assert np.array_equal(host(ctx.local_std(dev(ctx, x), H, W, 5), (H, W)), np.sqrt(var))
    u8 = (x * 255).astype(np.uint8)
    assert np.array_equal(host(ctx.morph_gradient(dev(ctx, u8), H, W, 5), (H, W)), oracle.morph_gradient_u8(u8, 5))
    # sobel_mag_feature re-normalises; feed a plane whose p2/p98 normalisation is the identity
    xn = oracle.robust_normalize(x)
    xn2 = oracle.robust_normalize(xn)
    q = (xn2 * 255).astype(np.uint8)
    got = host(ctx.sobel_mag(dev(ctx, q), H, W), (H, W))
    assert np.array_equal(got, oracle.sobel_mag_feature(xn)), np.abs(got - oracle.sobel_mag_feature(xn)).max()
    # tiny images: borders reflect more than once
    t = rng.random((2, 3)).astype(np.float32)
    assert np.array_equal(host(ctx.box_mean(dev(ctx, t), 2, 3, 3, L.BORDER_REFLECT), (2, 3)), oracle.box_mean(t, 3, "reflect"))


def test_morphology_variance_laplacian_bitexact_vs_oracle(ctx, oracle):
    """The remaining uint8 / float32 window members of the feature dict (erosion, dilation, opening, closing at 3/5/7,
    variance_scale_k, laplacian) against the oracle, bit for bit, on an odd-sized plane."""
    from rsseg import _lib as L
    rng = np.random.default_rng(21)
    H, W = 67, 201
    u8 = rng.integers(0, 256, (H, W)).astype(np.uint8)
    u8[20:40, 50:90] = 17
    ops = (("erosion", L.MORPH_ERODE), ("dilation", L.MORPH_DILATE), ("opening", L.MORPH_OPEN), ("closing", L.MORPH_CLOSE),
           ("gradient", L.MORPH_GRADIENT))
    for k in (3, 5, 7):
        for name, op in ops:
            assert np.array_equal(host(ctx.morph(dev(ctx, u8), H, W, k, op), (H, W)), oracle.morph_u8(u8, k, name)), (name, k)
    x = rng.random((H, W)).astype(np.float32)
    for k in (3, 5, 7):
        mean = oracle.box_mean(x, k, "reflect101")
        var = oracle.box_mean(x * x, k, "reflect101") - mean * mean
        var[var < 0] = 0
        assert np.array_equal(host(ctx.local_var(dev(ctx, x), H, W, k), (H, W)), var), k
    xn = oracle.robust_normalize(x)
    q = (oracle.robust_normalize(xn) * 255).astype(np.uint8)
    got = host(ctx.laplacian_norm(dev(ctx, q), H, W), (H, W))
    assert np.array_equal(got, oracle.laplacian_feature(xn)), np.abs(got - oracle.laplacian_feature(xn)).max()
    with pytest.raises(Exception):
        ctx.morph(dev(ctx, u8), H, W, 4, L.MORPH_ERODE)


def test_stack19_and_class_map_end_to_end(ctx, scene, oracle, golden_dir):
    """Bundled scene -> 19-feature stack on the GPU -> bundled forest on the GPU == the reference's
    committed class_map.npy; the stack equals the oracle's (float columns within 1e-5)."""
    from rsseg import pipeline as P
    bands = oracle.stage1_preprocess(scene["dn"])
    planes, extras = P.feature_stack19(ctx, [dev(ctx, b) for b in bands], 600, 600)
    stack = P.stack19_to_host(planes, 600, 600)
    _, hier = oracle.run_feature_extraction_stage(bands)
    assert stack.shape == hier["all"].shape and stack.dtype == np.float64
    for c in range(19):
        # columns 6 and 13 are PC0 and its 7x7 mean: float32 BLAS noise of the CPU path at N = 360 000
        # (see test_pca_full_scene_vs_reference); every other column within 1e-5
        tol = 2e-4 if c in (6, 13) else 1e-5
        assert np.allclose(stack[:, :, c], hier["all"][:, :, c], rtol=0, atol=tol), (c, np.abs(stack[:, :, c] - hier["all"][:, :, c]).max())
    for c in (0, 1, 2, 3, 4, 5, 14, 15, 16, 17, 18):  # IEEE-elementwise / integer columns: bit-exact
        assert np.array_equal(stack[:, :, c], hier["all"][:, :, c]), c
    f = dict(np.load(os.path.join(golden_dir, "rf_samples_model_flat.npz")))
    ctx.forest_load(f)
    fplanes = P.stack19_forest_planes(ctx, planes)
    # gradient_5 is float64 uint8/255.0 in the reference and is cast to float32 by sklearn
    assert np.array_equal(host(fplanes[16]), (host(planes[16]).astype(np.float64) / 255.0).astype(np.float32))
    cm = host(ctx.forest_predict(fplanes), (600, 600))
    assert cm.dtype == np.int64
    assert float(np.mean(cm == scene["class_map"])) >= 0.999
    for (x, y), lab in zip(scene["sample_coords"], scene["sample_labels"]):
        assert cm[y, x] == lab


# ------------------------------------------------------------------------------------------------ K9/K10
@pytest.mark.parametrize("k", [6, 7, 8])
def test_kmeans_labels_bitexact_vs_oracle_and_sklearn(ctx, crop, oracle, k):
    planes = [crop["idx_" + n] for n in KEYS]
    want, info = oracle.kmeans_fit_planes(planes, k)
    labels, meta = ctx.kmeans_fit_predict([dev(ctx, p) for p in planes], k)
    got = host(labels)
    assert got.dtype == np.int32
    assert np.array_equal(meta["init_indices"], info["init_indices"])
    assert meta["n_iter"] == info["n_iter"]
    assert np.array_equal(got, want)
    assert np.array_equal(got, crop[f"kmeans_idx7_k{k}"].reshape(-1))  # the reference function's labels
    assert np.array_equal(meta["scale"], info["scale"]) and np.array_equal(meta["mean"], info["mean"])
    assert meta["tol"] == info["tol"]


@pytest.mark.parametrize("k", [6, 8])
def test_kmeans_float64_stack19(ctx, crop, oracle, k):
    st = crop["stack19"]
    planes = [np.ascontiguousarray(st[:, :, i]) for i in range(19)]
    want, info = oracle.kmeans_fit_planes(planes, k)
    labels, meta = ctx.kmeans_fit_predict([dev(ctx, p) for p in planes], k)
    assert meta["n_iter"] == info["n_iter"]
    assert np.array_equal(host(labels), want)
    assert np.array_equal(host(labels), crop[f"kmeans_stack19_k{k}"].reshape(-1))


def test_kmeans_nan_and_ragged_and_constant_feature(ctx, crop, oracle):
    planes = [crop["idx_" + n].reshape(-1)[:9001].copy() for n in KEYS]  # n % 4 != 0, partial tile
    planes[0][[5, 77, 4000]] = np.nan
    planes.append(np.full(9001, 0.25, np.float32))  # zero range -> scale 1
    want, info = oracle.kmeans_fit_planes(planes, 5)
    labels, meta = ctx.kmeans_fit_predict([dev(ctx, p) for p in planes], 5)
    assert np.array_equal(host(labels), want)
    assert meta["n_iter"] == info["n_iter"]


def test_pipelines_with_nodata_pixels_in_a_band(ctx, oracle):
    """scripts/2 turns nodata into NaN (scripts/2_feature_extraction.py:160-175); np.percentile of such a band is NaN, so
    robust_normalize makes the WHOLE band NaN (indices.py:38-45), every index that reads it is NaN, KMeans clusters NaN as 0
    (extract.py:560-566) and PCA refuses the stack like scikit-learn.  Config 2 with NaNs in the green band: planes (NaN
    pattern included) and labels against the oracle; config 3 and the 19-feature stack raise instead of computing."""
    from rsseg import pipeline as P
    H, W = 128, 160
    r = oracle.synthetic_raster(H, W).copy()
    r[1, 10:14, 20:40] = np.nan
    bands = [dev(ctx, r[i].reshape(-1)) for i in range(7)]
    labels, meta, planes = P.config2(ctx, bands, 6)
    b, g, rd, n, s = [oracle.robust_normalize(r[i]) for i in range(5)]
    assert np.isnan(g).all()
    ref_planes = [oracle.calculate_ndvi(n, rd), oracle.calculate_evi(n, rd, b), oracle.calculate_msavi(n, rd), oracle.calculate_ndwi(g, n),
                  oracle.calculate_mndwi(g, s), oracle.calculate_ndbi(s, n), oracle.calculate_bsi(b, rd, n, s)]
    for name, gp, rp in zip(P.INDEX_NAMES, planes, ref_planes):
        assert np.array_equal(host(gp, (H, W)), rp, equal_nan=True), name
    want, info = oracle.kmeans_fit_planes([p.reshape(-1) for p in ref_planes], 6)
    assert meta["n_iter"] == info["n_iter"] and np.array_equal(host(labels), want)
    with pytest.raises(Exception, match="NaN"):
        P.config3(ctx, bands, H, W, 8, 7, 1, 3)
    with pytest.raises(Exception, match="NaN"):
        P.feature_stack19(ctx, bands, H, W)


def test_kmeans_full_scene_bitexact_vs_oracle(ctx, scene, oracle):
    bands = oracle.stage1_preprocess(scene["dn"])
    norm = [oracle.robust_normalize(b) for b in bands]
    b, g, r, n, s = norm[:5]
    planes = [oracle.calculate_ndvi(n, r), oracle.calculate_evi(n, r, b), oracle.calculate_msavi(n, r),
              oracle.calculate_ndwi(g, n), oracle.calculate_mndwi(g, s), oracle.calculate_ndbi(s, n),
              oracle.calculate_bsi(b, r, n, s)]
    for k in (6, 8):
        want, info = oracle.kmeans_fit_planes(planes, k)
        labels, meta = ctx.kmeans_fit_predict([dev(ctx, p) for p in planes], k)
        assert meta["n_iter"] == info["n_iter"], (meta["n_iter"], info["n_iter"])
        assert np.array_equal(host(labels), want)
        assert np.allclose(meta["centers"] - meta["mean"], info["centers"], rtol=0, atol=1e-6)


def test_kmeans_full_scene_vs_reference_goldens(ctx, scene, oracle, golden_dir):
    """GPU labels on the 600 x 600 scene against the labels the REFERENCE function produced there
    (tests/golden/scene_aa_ref_outputs.npz, scikit-learn 1.7.2, one thread), k = 6 and k = 8.  scikit-learn's float32
    accumulation order moves ~1e-4 of the labels between its own runs (1 thread vs 8: 34 / 71 labels); the
    GPU path differs from the golden in no more labels than were recorded when the goldens were made (47 / 66), and
    every differing pixel is a near-tie between exactly the two centres involved."""
    import json
    ref = np.load(os.path.join(golden_dir, "scene_aa_ref_outputs.npz"))
    report = json.load(open(os.path.join(golden_dir, "PIN_REPORT.json")))
    bands = oracle.stage1_preprocess(scene["dn"])
    norm = [oracle.robust_normalize(b) for b in bands]
    b, g, r, n, s = norm[:5]
    planes = [oracle.calculate_ndvi(n, r), oracle.calculate_evi(n, r, b), oracle.calculate_msavi(n, r),
              oracle.calculate_ndwi(g, n), oracle.calculate_mndwi(g, s), oracle.calculate_ndbi(s, n),
              oracle.calculate_bsi(b, r, n, s)]
    X = np.stack([p.reshape(-1) for p in planes], 1).astype(np.float64)
    for k in (6, 8):
        labels, meta = ctx.kmeans_fit_predict([dev(ctx, p) for p in planes], k)
        got = host(labels)
        refl = ref[f"kmeans_idx7_k{k}"].reshape(-1).astype(np.int32)
        bad = np.nonzero(got != refl)[0]
        assert bad.size <= report[f"kmeans_scene_idx7_k{k}"]["mismatch"], (k, bad.size)
        Xs = X * meta["scale"] + meta["min"] - meta["mean"]
        C = meta["centers"] - meta["mean"]                      # centres in the centred space
        d = ((Xs[bad, None, :] - C[None, :, :]) ** 2).sum(-1)
        ar = np.arange(bad.size)
        gap = np.abs(d[ar, got[bad]] - d[ar, refl[bad]])
        assert gap.max() < 2e-3, (k, gap.max())
        # ... and the two labels are the pixel's two nearest centres (in either order: the gap is below float32 resolution)
        order = np.sort(np.argsort(d, axis=1)[:, :2], axis=1)
        pair = np.sort(np.stack([got[bad], refl[bad]], 1), axis=1)
        assert np.array_equal(order, pair), k


def test_config2_at_its_real_size_equals_the_oracle(ctx, oracle):
    """BASELINE configs[1] at its real size, exact: the 4096 x 4096 synthetic TM raster (SURVEY 8d generator), percentile
    normalisation + 7 spectral indices + KMeans(k = 6) — label map, seeds, iteration count and centres against the CPU
    oracle (about a minute of CPU time; the one full-size case the oracle reaches)."""
    from rsseg import pipeline as P
    H = W = 4096
    r = oracle.synthetic_raster(H, W, bands=5)
    labels, meta, planes = P.config2(ctx, [dev(ctx, r[i].reshape(-1)) for i in range(5)], 6)
    got = host(labels)
    b, g, rd, n, s = [oracle.robust_normalize(r[i]) for i in range(5)]
    ref_planes = [oracle.calculate_ndvi(n, rd), oracle.calculate_evi(n, rd, b), oracle.calculate_msavi(n, rd), oracle.calculate_ndwi(g, n),
                  oracle.calculate_mndwi(g, s), oracle.calculate_ndbi(s, n), oracle.calculate_bsi(b, rd, n, s)]
    for name, gp, rp in zip(P.INDEX_NAMES, planes, ref_planes):
        assert np.array_equal(host(gp).view(np.int32), rp.reshape(-1).view(np.int32)), name
    want, info = oracle.kmeans_fit_planes([p.reshape(-1) for p in ref_planes], 6)
    assert meta["n_iter"] == info["n_iter"]
    assert np.array_equal(meta["init_indices"], info["init_indices"])
    assert np.array_equal(got, want)
    assert np.allclose(meta["centers"] - meta["mean"], info["centers"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("n,F,k", [(50, 3, 1), (6, 2, 6), (7, 1, 3), (1000, 1, 2), (5000, 64, 2), (2, 1, 2), (300, 5, 64)])
def test_kmeans_corner_sizes(ctx, oracle, n, F, k):
    """One cluster, as many clusters as points, one feature, the widest plane count, k = RSSEG's 64: labels, seeds and
    iteration count against the oracle (which equals scikit-learn on such inputs, tests/test_oracle.py)."""
    rng = np.random.default_rng(n * 1000 + F * 10 + k)
    planes = [rng.random(n).astype(np.float32) for _ in range(F)]
    want, info = oracle.kmeans_fit_planes(planes, k)
    labels, meta = ctx.kmeans_fit_predict([dev(ctx, p) for p in planes], k)
    assert meta["n_iter"] == info["n_iter"]
    assert np.array_equal(meta["init_indices"], info["init_indices"])
    assert np.array_equal(host(labels), want)


@pytest.mark.parametrize("n,F,k", [(400, 12, 40), (600, 20, 64), (300, 5, 64), (500, 31, 33)])
def test_kmeans_float64_planes_with_more_than_32_clusters(ctx, oracle, n, F, k):
    """float64 planes, 33 ... 64 clusters: 9 ... 32 planes take the feature-blocked Lloyd kernel (the register-resident form
    would need scratch memory there), up to 8 planes the register-resident one; labels, seeds and iteration
    count against the oracle, and the float32 twin of the same data for comparison of the code paths."""
    rng = np.random.default_rng(n + F + k)
    centres = rng.random((k, F))
    planes = [np.ascontiguousarray((centres[rng.integers(0, k, n), f] + rng.normal(0, 0.02, n))) for f in range(F)]
    want, info = oracle.kmeans_fit_planes(planes, k)
    labels, meta = ctx.kmeans_fit_predict([dev(ctx, p) for p in planes], k)
    assert planes[0].dtype == np.float64 and meta["n_iter"] == info["n_iter"]
    assert np.array_equal(meta["init_indices"], info["init_indices"])
    assert np.array_equal(host(labels), want)
    p32 = [p.astype(np.float32) for p in planes]
    want32, info32 = oracle.kmeans_fit_planes(p32, k)
    labels32, meta32 = ctx.kmeans_fit_predict([dev(ctx, p) for p in p32], k)
    assert meta32["n_iter"] == info32["n_iter"] and np.array_equal(host(labels32), want32)


def test_kmeans_errors(ctx):
    x = dev(ctx, np.zeros(3, np.float32))
    with pytest.raises(ValueError):
        ctx.kmeans_fit_predict([x], 5)  # n_samples < n_clusters
    from rsseg.runtime import RssegUnsupported
    with pytest.raises(RssegUnsupported):
        ctx.kmeans_fit_predict([dev(ctx, np.zeros(300, np.float32))] * 65, 2)  # more planes than RSSEG_MAX_FEATURES (64)


def test_producer_minmax_tags_and_kmeans_shortcut(ctx, crop, oracle):
    """Planes written while rsseg_ctx_collect_minmax is on carry their exact (min, max), NaN counted as 0, and KMeans
    given such planes (no MinMaxScaler pass) returns what it returns for untagged copies."""
    import torch
    from rsseg import pipeline as P
    bands = [dev(ctx, b) for b in crop["bands"]]
    ctx.collect_minmax(True)
    try:
        lohi = P.band_lohi(ctx, bands[:5])
        idx, _ = P.spectral_indices(ctx, bands, lohi)
        planes = [idx[n] for n in P.INDEX_NAMES]
        up = ctx.resize_bilinear(planes[0][:90 * 96].contiguous(), 90, 96, 96, 96)
        normd = [ctx.normalize(b, 0.0, 255.0) for b in bands]
        pcs, *_ = ctx.pca_fit_transform(normd, None, None, 3)
    finally:
        ctx.collect_minmax(False)
    for t in planes + [up] + list(pcs):
        mn, mx = t._rsseg_minmax
        z = torch.nan_to_num(t, nan=0.0)
        assert mn == float(z.min()) and mx == float(z.max())
    feats = planes + list(pcs)
    lab_a, meta_a = ctx.kmeans_fit_predict(feats, 6)
    lab_b, meta_b = ctx.kmeans_fit_predict([t.clone() for t in feats], 6)   # clones carry no tag: full scaler pass
    assert torch.equal(lab_a, lab_b) and meta_a["n_iter"] == meta_b["n_iter"]
    assert np.array_equal(meta_a["centers"], meta_b["centers"]) and np.array_equal(meta_a["scale"], meta_b["scale"])
    # a plane produced with collection off is not tagged
    assert not hasattr(ctx.resize_bilinear(planes[0][:90 * 96].contiguous(), 90, 96, 96, 96), "_rsseg_minmax")


def test_kmeans_large_properties(ctx, oracle):
    """2048 x 2048 synthetic raster: beyond what the CPU oracle finishes quickly, so check
    size-independent properties: determinism, every label is the argmin of its distance row, the centres
    are the means of their members, labels invariant to a permutation of pixel order within chunks."""
    import torch
    from rsseg import pipeline as P
    H = W = 2048
    r = oracle.synthetic_raster(H, W)
    bands = [dev(ctx, r[i]) for i in range(7)]
    labels, meta, planes = P.config2(ctx, bands, 6)
    labels2, meta2, _ = P.config2(ctx, bands, 6)
    assert torch.equal(labels, labels2) and meta["n_iter"] == meta2["n_iter"]
    lab = host(labels)
    assert lab.min() == 0 and lab.max() == 5
    X = np.stack([host(p) for p in planes], 1).astype(np.float64)
    Xs = X * meta["scale"] + meta["min"]
    C = meta["centers"]
    # centres = member means (the final centres come from the last M-step; labels from the E-step after it)
    cnt = np.bincount(lab, minlength=6)
    assert cnt.min() > 0
    sub = np.random.default_rng(0).choice(lab.size, 200000, replace=False)
    d = ((Xs[sub, None, :] - C[None]) ** 2).sum(-1)
    best = d.min(1)
    mine = d[np.arange(sub.size), lab[sub]]
    assert np.all(mine - best <= 1e-5)


# ------------------------------------------------------------------------------------------------ K11
def test_forest_bitexact_vs_sklearn_goldens(ctx, crop, golden_dir, oracle):
    f = dict(np.load(os.path.join(golden_dir, "rf_samples_model_flat.npz")))
    ctx.forest_load(f)
    X = crop["rf_X"]
    out = host(ctx.forest_predict([dev(ctx, X[:, i]) for i in range(19)]))
    assert out.dtype == np.int64
    assert np.array_equal(out, crop["rf_pred_image"].reshape(-1))
    Xn = crop["rf_X_nan"]
    out = host(ctx.forest_predict([dev(ctx, Xn[:, i]) for i in range(19)]))
    assert np.array_equal(out, crop["rf_pred_nan_native"].reshape(-1))
    with pytest.raises(ValueError):
        ctx.forest_predict([dev(ctx, X[:, i]) for i in range(5)])


def test_forest_deep_synthetic_vs_oracle(ctx, oracle):
    """100 trees, depth up to 16 (BASELINE config 5 schema), fitted on host with scikit-learn."""
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.default_rng(42)
    Xtr = rng.random((20000, 19)).astype(np.float32)
    ytr = (Xtr[:, 0] * 3 + Xtr[:, 5] * 2 + Xtr[:, 11] > 2.6).astype(np.int64) + (Xtr[:, 7] > 0.8) * 2
    flip = rng.random(20000) < 0.1
    ytr[flip] = rng.integers(0, 4, flip.sum())
    model = RandomForestClassifier(n_estimators=100, max_depth=16, random_state=42, n_jobs=8).fit(Xtr, ytr)
    f = oracle.flatten_forest(model)
    X = rng.random((30011, 19)).astype(np.float32)
    want = model.predict(X)
    assert np.array_equal(oracle.rf_predict_planes(f, [X[:, i] for i in range(19)]), want)
    ctx.forest_load(f)
    got = host(ctx.forest_predict([dev(ctx, X[:, i]) for i in range(19)]))
    assert np.array_equal(got, want)


def test_order_stats_multi_equals_single(ctx, scene, oracle):
    """The grouped select (all planes advance pass by pass together) returns what one select per plane returns,
    including a plane with NaNs and planes whose ranks fall into the same and into different radix prefixes."""
    bands = oracle.stage1_preprocess(scene["dn"])
    planes = [bands[i].reshape(-1).copy() for i in range(7)]
    planes[2][::97] = np.nan
    planes.append(np.linspace(-3.0, 5.0, planes[0].size, dtype=np.float32))
    n = planes[0].size
    dv = [dev(ctx, p) for p in planes]
    ranks = [[0, n // 50, n // 2, n // 2 + 1, n - 1 - 4000, (7 * i) % (n - 4000)] for i in range(len(planes))]
    vals, nans = ctx.order_stats_multi(dv, ranks)
    for i, d in enumerate(dv):
        v1, nn1 = ctx.order_stats(d, ranks[i])
        assert np.array_equal(vals[i], v1, equal_nan=True) and nans[i] == nn1, i
        srt = np.sort(planes[i])
        assert np.array_equal(vals[i], srt[ranks[i]], equal_nan=True), i
    with pytest.raises(ValueError):
        ctx.order_stats_multi(dv + dv[:1], ranks + ranks[:1])  # more than 8 planes


def test_forest_six_classes_and_nan_rows_vs_sklearn(ctx, oracle):
    """More than four classes (the wider vote-accumulator instantiation), impure leaves, NaN features in some rows
    (missing-value rule) and a pixel count that is not a multiple of the workgroup size."""
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.default_rng(7)
    Xtr = rng.random((6000, 19)).astype(np.float32)
    ytr = (Xtr[:, 2] * 6).astype(np.int64) % 6
    flip = rng.random(6000) < 0.2
    ytr[flip] = rng.integers(0, 6, flip.sum())
    model = RandomForestClassifier(n_estimators=25, max_depth=9, min_samples_leaf=3, random_state=1, n_jobs=4).fit(Xtr, ytr)
    f = oracle.flatten_forest(model)
    X = rng.random((5003, 19)).astype(np.float32)
    X[::37, 2] = np.nan
    X[5::53, 11] = np.nan
    want = model.predict(X)
    ctx.forest_load(f)
    got = host(ctx.forest_predict([dev(ctx, X[:, i]) for i in range(19)]))
    assert len(f["classes"]) == 6 and np.array_equal(got, want)


def test_forest_nan_threshold_splits_vs_sklearn(ctx, oracle):
    """Forests fitted on data WITH missing values: scikit-learn (>= 1.4) may then choose a split that only separates missing from
    non-missing values and records it with threshold NaN — `x <= NaN` is false, every non-missing value goes right, a missing one
    where missing_go_to_left says.  (Found by tests/test_gpu_fuzz.py in round 4: the loader used to store the NaN as it was, and
    NaN thresholds mark leaves in the walk.)  Rows with and without NaNs, both kernels' class widths, against model.predict and
    the oracle."""
    from sklearn.ensemble import RandomForestClassifier
    found = 0
    for seed, ncls, F in ((0, 3, 5), (1, 13, 19), (2, 8, 55), (3, 9, 19), (4, 4, 3), (5, 20, 7)):
        rng = np.random.default_rng(500 + seed)
        Xtr = rng.random((30, F)).astype(np.float32)
        Xtr[rng.random((30, F)) < 0.08] = np.nan
        ytr = rng.integers(0, ncls, 30) * 7 - 20
        model = RandomForestClassifier(n_estimators=25, max_depth=6, random_state=seed, n_jobs=1).fit(Xtr, ytr)
        n_nan_thr = sum(int(np.isnan(e.tree_.threshold[e.tree_.children_left >= 0]).sum()) for e in model.estimators_)
        found += n_nan_thr
        f = oracle.flatten_forest(model)
        ctx.forest_load(f)
        for with_nan in (False, True):
            X = rng.random((5003, F)).astype(np.float32)
            if with_nan:
                X[rng.random((5003, F)) < 0.05] = np.nan
            got = host(ctx.forest_predict([dev(ctx, X[:, i]) for i in range(F)]))
            assert np.array_equal(got, model.predict(X)), (seed, with_nan, n_nan_thr)
            assert np.array_equal(got, oracle.rf_predict_planes(f, [np.ascontiguousarray(X[:, i]) for i in range(F)])), (seed, with_nan)
    assert found > 0          # the case is really exercised


def test_forest_degenerate_shapes_vs_sklearn(ctx, oracle):
    """A forest whose trees are single leaves (one class in the training set), a single tree, a single stump, one pixel
    and an empty raster: the walk has nothing to walk, the vote table has one row per leaf."""
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.default_rng(77)
    Xtr = rng.random((200, 5)).astype(np.float32)
    X = rng.random((1031, 5)).astype(np.float32)
    cases = {
        "one_class": RandomForestClassifier(n_estimators=7, random_state=0).fit(Xtr, np.full(200, 3)),
        "one_tree": RandomForestClassifier(n_estimators=1, max_depth=6, random_state=0).fit(Xtr, (Xtr[:, 1] * 4).astype(int)),
        "stumps": RandomForestClassifier(n_estimators=5, max_depth=1, random_state=0).fit(Xtr, (Xtr[:, 0] > 0.5).astype(int) * 7 + 2),
    }
    for name, model in cases.items():
        ctx.forest_load(oracle.flatten_forest(model))
        got = host(ctx.forest_predict([dev(ctx, X[:, i]) for i in range(5)]))
        assert np.array_equal(got, model.predict(X)), name
        one = host(ctx.forest_predict([dev(ctx, X[:1, i]) for i in range(5)]))
        assert np.array_equal(one, model.predict(X[:1])), name


@pytest.mark.parametrize("F", [1, 2, 8, 32, 33, 55, 64])
def test_forest_feature_counts_vs_sklearn(ctx, oracle, F):
    """Feature counts 1, even, 32 (the last count with 1024 pixels per workgroup) and 33 / 55 / 64 (512 pixels per
    workgroup; 55 = every 2-D plane of the stage-2 dictionary, the reference's non-hierarchical forest branch,
    scripts/3_classification.py:425-437): a pixel's features are one LDS row of F | 1 floats and a node's byte 3 is
    4 x feature; NaN rows included; a pixel count that leaves the last workgroup ragged."""
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.default_rng(100 + F)
    Xtr = rng.random((4000, F)).astype(np.float32)
    ytr = ((Xtr.sum(1) * 3).astype(np.int64) + (rng.random(4000) < 0.1) * rng.integers(0, 4, 4000)) % 4
    model = RandomForestClassifier(n_estimators=9, max_depth=8, random_state=F, n_jobs=4).fit(Xtr, ytr)
    X = rng.random((3001, F)).astype(np.float32)
    X[::29, F - 1] = np.nan
    X[3::31, 0] = np.nan
    ctx.forest_load(oracle.flatten_forest(model))
    got = host(ctx.forest_predict([dev(ctx, X[:, i]) for i in range(F)]))
    assert np.array_equal(got, model.predict(X))


def test_forest_large_trees_many_classes_and_infinities_vs_sklearn(ctx, oracle):
    """(a) Unpruned trees with more nodes than the LDS holds (the general kernel: upper levels in LDS, deeper nodes by
    global loads), with NaN rows; (b) 11 and 20 classes (sklearn has no class limit; the wider vote tables);
    (c) +/-inf features walk like any other value (x <= thr), as sklearn's tree does."""
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.default_rng(3)
    Xtr = rng.random((60000, 7)).astype(np.float32)
    ytr = rng.integers(0, 3, 60000)                         # pure noise: trees grow until every leaf is pure
    big = RandomForestClassifier(n_estimators=6, max_depth=None, random_state=0, n_jobs=8).fit(Xtr, ytr)
    assert max(e.tree_.node_count for e in big.estimators_) > 20000
    X = rng.random((40001, 7)).astype(np.float32)
    X[::41, 3] = np.nan
    X[7::43, 0] = np.inf
    X[9::47, 5] = -np.inf
    ctx.forest_load(oracle.flatten_forest(big))
    got = host(ctx.forest_predict([dev(ctx, X[:, i]) for i in range(7)]))
    Xs = X.copy()
    fin = np.isfinite(Xs) | np.isnan(Xs)
    Xs[~fin] = np.sign(Xs[~fin]) * 3.0e38                    # sklearn refuses inf input; the same comparisons with huge values
    assert np.array_equal(got, big.predict(Xs))
    for ncls in (11, 20, 40, 64):                              # 33 .. 64 classes: 512 pixels per workgroup, 64-wide vote rows
        Xt = rng.random((8000, 5)).astype(np.float32)
        yt = ((Xt[:, 1] * ncls).astype(np.int64) + (rng.random(8000) < 0.15) * rng.integers(0, ncls, 8000)) % ncls * 3 + 100  # labels 100, 103, ...
        m = RandomForestClassifier(n_estimators=15, max_depth=10, random_state=2, n_jobs=4).fit(Xt, yt)
        assert len(m.classes_) == ncls
        Xp = rng.random((3001, 5)).astype(np.float32)
        ctx.forest_load(oracle.flatten_forest(m))
        assert np.array_equal(host(ctx.forest_predict([dev(ctx, Xp[:, i]) for i in range(5)])), m.predict(Xp)), ncls
    # beyond the capacity of the kernels the library refuses loudly (no all-zero map): 65 classes
    f65 = dict(oracle.flatten_forest(m))
    f65["value"] = np.concatenate([f65["value"], np.zeros((f65["value"].shape[0], 65 - f65["value"].shape[1]))], 1)
    f65["classes"] = np.arange(65)
    from rsseg.runtime import RssegUnsupported
    with pytest.raises(RssegUnsupported):
        ctx.forest_load(f65)


def test_quantile_bundle_equals_separate_selects(ctx, scene, oracle):
    """One select per band (config-3 fast path) gives the same six statistics as the separate NumPy-style
    calls on the raw and on the normalised band."""
    from rsseg import pipeline as P
    from rsseg.quantiles import band_percentiles, robust_scaler_stats
    bands = oracle.stage1_preprocess(scene["dn"])
    for b in (bands[0], bands[3], bands[6]):
        d = dev(ctx, b)
        q = P.band_quantile_bundle(ctx, d)
        lo, hi = band_percentiles(ctx, d, (2, 98))
        assert q["lo"] == lo and q["hi"] == hi
        nd = ctx.normalize(d, float(lo), float(hi))
        c, s = robust_scaler_stats(ctx, nd)
        assert q["center"] == c and q["scale"] == s
        lo2, hi2 = band_percentiles(ctx, nd, (2, 98))
        assert q["lo2"] == lo2 and q["hi2"] == hi2
        nb = oracle.robust_normalize(b)
        assert q["lo2"] == np.percentile(nb, 2) and q["hi2"] == np.percentile(nb, 98)
    # NaN band: falls back
    x = bands[1].copy()
    x[3, 4] = np.nan
    q = P.band_quantile_bundle(ctx, dev(ctx, x))
    assert np.isnan(q["lo"]) and q["center"] is None


@pytest.mark.parametrize("H,W", [(192, 160), (2048, 2048)])
def test_config3_pipeline_vs_oracle(ctx, oracle, H, W):
    """BASELINE config 3 on a 192 x 160 synthetic tile and on a 2048 x 2048 one (4 Mpixel: 256 k-means chunks, multi-block
    reductions everywhere — the size bench.py's CPU baseline runs): features and labels against the CPU oracle."""
    from rsseg import pipeline as P
    r = oracle.synthetic_raster(H, W)
    labels, meta, planes = P.config3(ctx, [dev(ctx, r[i]) for i in range(7)], H, W, 8, 7, 1, 3)
    norm = [oracle.robust_normalize(r[i]) for i in range(7)]
    b, g, rd, n, s = norm[:5]
    feats = [oracle.calculate_ndvi(n, rd), oracle.calculate_evi(n, rd, b), oracle.calculate_msavi(n, rd),
             oracle.calculate_ndwi(g, n), oracle.calculate_mndwi(g, s), oracle.calculate_ndbi(s, n),
             oracle.calculate_bsi(b, rd, n, s)]
    gl, _ = oracle.calculate_glcm_features(norm[3], 32, 7, 1)
    feats += [gl[x] for x in ("contrast", "dissimilarity", "homogeneity", "energy", "correlation")]
    for i in range(12):
        assert np.array_equal(host(planes[i], (H, W)), feats[i]), i
    truth, _ = _pca_truth64(norm)  # float64 evaluation; the float32 CPU path carries ~1e-5 BLAS noise itself
    pcs, _, _ = oracle.perform_pca(norm, n_components=3)
    for i in range(3):
        got = host(planes[12 + i], (H, W))
        assert np.abs(got - truth[i].reshape(H, W)).max() <= 1e-5
        assert np.abs(got - pcs[i]).max() <= (1e-4 if H * W < 100000 else 1e-3)   # scikit-learn's float32 Gram: its noise grows with N
    # labels: bit-exact against the oracle KMeans run on the GPU's own feature planes
    want, info = oracle.kmeans_fit_planes([host(p, (H, W)) for p in planes], 8)
    assert meta["n_iter"] == info["n_iter"]
    assert np.array_equal(host(labels), want)


@pytest.mark.parametrize("as_u8", [False, True])
def test_texture_planes_from_the_nir_band_alone_equal_config3(ctx, oracle, as_u8):
    """pipeline.texture_planes (the texture chain from the NIR band and its quantile bundle alone: what a caller that uploads NIR
    first runs while the other bands are still crossing PCIe) gives config 3's five texture planes bit for bit, and config3
    handed those planes and the per-band bundles gives the same labels, seeds and iteration count as config3 on its own."""
    import torch
    from rsseg import pipeline as P
    H, W = 300, 260
    r = oracle.synthetic_raster(H, W)
    bands = [ctx.to_device(np.ascontiguousarray(r[i].astype(np.uint8) if as_u8 else r[i]).reshape(-1)) for i in range(7)]
    labels, meta, planes = P.config3(ctx, bands, H, W, 8, 7, 1, 3)
    qb = [P.band_quantile_bundle(ctx, b) for b in bands]
    tex = P.texture_planes(ctx, bands[3], qb[3], H, W, 7, 1)
    for i, name in enumerate(P.GLCM_NAMES):
        assert torch.equal(tex[name].view(torch.int32), planes[7 + i].view(torch.int32)), name
        assert getattr(tex[name], "_rsseg_minmax", None) == getattr(planes[7 + i], "_rsseg_minmax", None) is not None
    labels2, meta2, planes2 = P.config3(ctx, bands, H, W, 8, 7, 1, 3, qb=qb, glcm=tex)
    assert torch.equal(labels, labels2) and meta["n_iter"] == meta2["n_iter"] and np.array_equal(meta["init_indices"], meta2["init_indices"])
    for a, b in zip(planes, planes2):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def _label_diff_report(got, want, planes64, meta):
    """How two label maps differ, judged in the product's scaled and centred space: count, the largest squared-distance gap
    between the two labels' centres at a differing pixel, and how many differing pixels have exactly their two nearest
    centres as the two labels (a near-tie)."""
    bad = np.nonzero(got != want)[0]
    rep = {"differing": int(bad.size)}
    if bad.size:
        Xs = planes64[bad] * meta["scale"] + meta["min"] - meta["mean"]
        C = meta["centers"] - meta["mean"]
        d = ((Xs[:, None, :] - C[None, :, :]) ** 2).sum(-1)
        ar = np.arange(bad.size)
        rep["max_gap"] = float(np.abs(d[ar, got[bad]] - d[ar, want[bad]]).max())
        order = np.sort(np.argsort(d, axis=1)[:, :2], axis=1)
        pair = np.sort(np.stack([got[bad], want[bad]], 1), axis=1)
        rep["near_ties"] = int((order == pair).all(1).sum())
    return rep


@pytest.mark.parametrize("kind", ["easy", "hard"])
def test_bench_raster_whole_path_vs_cpu_oracle(ctx, oracle, kind):
    """The raster bench.py TIMES (bench.synth_rows: SURVEY 8d's prototypes with the noise drawn on the device; 'hard': the
    continuously mixed form) tied to the oracle at 2048 x 2048: generated on the GPU, copied to the host, pushed through the
    WHOLE CPU path — oracle feature planes -> oracle KMeans — and compared with the product's config 3: the 12 non-PCA
    planes bit for bit, the components within 1e-5 of the float64 evaluation, and three label comparisons
    (written to gpurun_out/r04/bench_raster_whole_path_<kind>.json; DESIGN 2 quotes them):
      A  product vs the CPU path with scikit-learn's own float32 PCA (the literal whole path);
      B  that CPU path vs the same CPU path fed the float64-exact components instead — the REFERENCE's sensitivity to its
         own ~3e-4 float32 PCA noise;
      C  product vs the CPU path fed the float64-exact components.
    C must be the same seeds, the same iteration count and near-ties only.  A is bounded by the recorded count when the
    seeds agree (easy raster: 4 labels of 4 194 304); when k-means++ draws a different seed in A, the same must happen in
    B — the reference then disagrees with ITSELF under a perturbation of the size of its own rounding noise, so no
    implementation can match it label for label — and A is reported, not bounded."""
    import json
    import torch
    import bench
    from rsseg import pipeline as P
    H = W = 2048
    dbands = bench.synth_rows(torch, ctx.device, W, 0, H, kind=kind)
    r = np.stack([b.cpu().numpy().reshape(H, W) for b in dbands])
    assert r.dtype == np.float32 and np.array_equal(r, np.round(r)) and r.min() >= 0 and r.max() <= 255
    labels, meta, planes = P.config3(ctx, dbands, H, W, 8, 7, 1, 3)
    norm = [oracle.robust_normalize(r[i]) for i in range(7)]
    b, g, rd, n, s = norm[:5]
    feats = [oracle.calculate_ndvi(n, rd), oracle.calculate_evi(n, rd, b), oracle.calculate_msavi(n, rd),
             oracle.calculate_ndwi(g, n), oracle.calculate_mndwi(g, s), oracle.calculate_ndbi(s, n),
             oracle.calculate_bsi(b, rd, n, s)]
    gl, _ = oracle.calculate_glcm_features(norm[3], 32, 7, 1)
    feats += [gl[x] for x in ("contrast", "dissimilarity", "homogeneity", "energy", "correlation")]
    for i in range(12):
        assert np.array_equal(host(planes[i], (H, W)), feats[i]), i
    truth, _ = _pca_truth64(norm)
    pcs, _, _ = oracle.perform_pca(norm, n_components=3)
    dev_truth, dev_sk = [], []
    for i in range(3):
        gp = host(planes[12 + i], (H, W))
        dev_truth.append(float(np.abs(gp - truth[i].reshape(H, W)).max()))
        dev_sk.append(float(np.abs(gp - pcs[i]).max()))
        assert dev_truth[-1] <= 1e-5
    cpu_sk, info_sk = oracle.kmeans_fit_planes(feats + list(pcs), 8)                                         # the literal CPU path
    cpu_ex, info_ex = oracle.kmeans_fit_planes(feats + [truth[i].reshape(H, W).astype(np.float32) for i in range(3)], 8)
    got = host(labels)
    X64 = np.stack([host(p).astype(np.float64) for p in planes], 1)
    seeds = lambda m: [int(x) for x in m["init_indices"]]   # noqa: E731
    rep = {"raster": f"bench.synth_rows kind={kind}, {H}x{W}", "pixels": H * W,
           "pc_max_abs_dev_from_float64": dev_truth, "pc_max_abs_dev_from_sklearn_float32": dev_sk,
           "A_product_vs_cpu_path": dict(_label_diff_report(got, cpu_sk, X64, meta), same_seeds=seeds(meta) == seeds(info_sk),
                                         n_iter=[int(meta["n_iter"]), int(info_sk["n_iter"])]),
           "B_cpu_path_vs_cpu_path_with_exact_pca": dict(_label_diff_report(cpu_sk, cpu_ex, X64, meta), same_seeds=seeds(info_sk) == seeds(info_ex),
                                                         n_iter=[int(info_sk["n_iter"]), int(info_ex["n_iter"])]),
           "C_product_vs_cpu_path_with_exact_pca": dict(_label_diff_report(got, cpu_ex, X64, meta), same_seeds=seeds(meta) == seeds(info_ex),
                                                        n_iter=[int(meta["n_iter"]), int(info_ex["n_iter"])])}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04")
    os.makedirs(out, exist_ok=True)
    json.dump(rep, open(os.path.join(out, f"bench_raster_whole_path_{kind}.json"), "w"), indent=1)
    print(json.dumps(rep))
    A, B, Cc = rep["A_product_vs_cpu_path"], rep["B_cpu_path_vs_cpu_path_with_exact_pca"], rep["C_product_vs_cpu_path_with_exact_pca"]
    assert Cc["same_seeds"] and Cc["n_iter"][0] == Cc["n_iter"][1], rep
    assert Cc["differing"] <= 64 and Cc.get("near_ties", 0) == Cc["differing"] and Cc.get("max_gap", 0.0) < 2e-3, rep
    if A["same_seeds"]:
        assert A["differing"] <= BENCH_RASTER_LABEL_BOUND[kind] and A.get("near_ties", 0) == A["differing"] and A.get("max_gap", 0.0) < 2e-3, rep
    else:
        assert not B["same_seeds"], rep     # the reference's own float32 PCA noise moves a seed: it disagrees with itself just the same


# labels of the 4 194 304 that may differ between the product and the literal CPU path on the bench rasters at 2048^2 when both
# draw the same seeds (recorded: easy 4; profiles/r04_bench_raster_whole_path_*.json), each pixel proven a near-tie
BENCH_RASTER_LABEL_BOUND = {"easy": 16, "hard": 4096}


@pytest.mark.parametrize("seed", [20, 22, 25, 26, 29, 30])
def test_kmeans_empty_cluster_relocation(ctx, oracle, seed):
    """Duplicate-heavy data with more clusters than distinct points: clusters run empty and are re-seeded
    with the farthest pixels (_relocate_empty_clusters_dense).  GPU == oracle bit for bit (and the oracle
    equals scikit-learn on these inputs, oracle/gen_golden-style check in the docstring of DESIGN.md §4)."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(20, 120)); F = int(rng.integers(1, 4)); k = int(rng.integers(6, 14))
    X = rng.random((n, F)).astype(np.float32)
    X = (np.round(X * rng.integers(2, 6)) / 4.0).astype(np.float32)
    planes = [np.ascontiguousarray(X[:, f]) for f in range(F)]
    want, info = oracle.kmeans_fit_planes(planes, k)
    assert info["relocated"] > 0
    labels, meta = ctx.kmeans_fit_predict([dev(ctx, p) for p in planes], k)
    assert meta["relocated"] == info["relocated"]
    assert meta["n_iter"] == info["n_iter"]
    assert np.array_equal(host(labels), want)
    from sklearn.cluster import KMeans
    from sklearn.preprocessing import MinMaxScaler
    from threadpoolctl import threadpool_limits
    import warnings
    with threadpool_limits(limits=1), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = KMeans(n_clusters=k, random_state=42, n_init="auto").fit_predict(MinMaxScaler().fit_transform(X))
    assert np.array_equal(host(labels), ref)


@pytest.mark.parametrize("k,F,dt", [(12, 5, np.float32), (20, 20, np.float32), (40, 3, np.float32), (33, 9, np.float64), (2, 1, np.float32),
                                    (1, 4, np.float32), (16, 32, np.float32),
                                    (5, 33, np.float32), (8, 55, np.float64), (12, 64, np.float32), (20, 40, np.float64), (40, 48, np.float32)])
def test_kmeans_many_clusters_and_features(ctx, oracle, k, F, dt):
    """Every kernel instantiation (KMAX 8/16/32/64 x 8/16/32 register-resident features, and the feature-blocked
    kernels for 33..64 features; float32 and float64) against the oracle, on clumpy data so that the iteration count is
    non-trivial."""
    rng = np.random.default_rng(k * 100 + F)
    n = 20011
    cent = rng.random((k + 3, F))
    X = (cent[rng.integers(0, k + 3, n)] + rng.normal(0, 0.08, (n, F))).astype(dt)
    planes = [np.ascontiguousarray(X[:, f]) for f in range(F)]
    want, info = oracle.kmeans_fit_planes(planes, k)
    labels, meta = ctx.kmeans_fit_predict([dev(ctx, p) for p in planes], k)
    assert np.array_equal(meta["init_indices"], info["init_indices"])
    assert meta["n_iter"] == info["n_iter"]
    assert np.array_equal(host(labels), want)


# ------------------------------------------------------------------------------------------------ full size
def _pattern_np(y0, y1, x0, x1, b):
    """Integer-valued synthetic DN in [0, 250]: blocks of 64 x 64 with a per-band level plus a hash texture."""
    y = np.arange(y0, y1, dtype=np.int64)[:, None]
    x = np.arange(x0, x1, dtype=np.int64)[None, :]
    base = (((y // 64) * 7 + (x // 64) * 3 + b * 5) % 8) * 24 + 20
    tex = ((y * 131 + x * 71 + b * 37) ^ ((y >> 3) * (x >> 2) + b)) % 23
    return (base + tex).astype(np.float32)


def _pattern_gpu(torch, device, H, W, b):
    out = torch.empty(H * W, dtype=torch.float32, device=device)
    x = torch.arange(W, device=device, dtype=torch.int64)[None, :]
    for y0 in range(0, H, 1024):
        y1 = min(H, y0 + 1024)
        y = torch.arange(y0, y1, device=device, dtype=torch.int64)[:, None]
        base = (((y // 64) * 7 + (x // 64) * 3 + b * 5) % 8) * 24 + 20
        tex = ((y * 131 + x * 71 + b * 37) ^ ((y >> 3) * (x >> 2) + b)) % 23
        out[y0 * W:y1 * W] = (base + tex).to(torch.float32).reshape(-1)
    return out


def test_full_size_16384_properties(ctx, oracle):
    """BASELINE's full size (16384 x 16384 x 7, 268 Mpx): far beyond the CPU oracle, so
      * order statistics against a device sort (exact),
      * the LOCAL stages (normalise + indices, quantise + GLCM small maps) against the oracle on a crop taken at the
        FAR corner of the raster, where any 32-bit index overflow would show (bit-exact; the global percentiles are
        handed to the oracle),
      * the row-striped upsample against the whole-plane upsample on the last rows (same values),
      * config 3 end to end: deterministic, every cluster centre is the float64 mean of its members' scaled features to
        1e-6, sampled labels are arg-min labels."""
    import torch
    from rsseg import pipeline as P
    H = W = 16384
    n = H * W
    bands = [_pattern_gpu(torch, ctx.device, H, W, b) for b in range(7)]
    # -- order statistics
    ranks = [int(0.02 * (n - 1)), int(0.02 * (n - 1)) + 1, n // 2, int(0.98 * (n - 1)), n - 1, 0]
    got, nn = ctx.order_stats(bands[3], ranks)
    assert nn == 0
    srt = torch.sort(bands[3]).values
    for r, g in zip(ranks, got):
        assert float(srt[r]) == float(g), r
    del srt
    # -- local stages at the far corner
    lohi = P.band_lohi(ctx, bands)
    idx, norms = P.spectral_indices(ctx, bands, lohi, want_norm=(True,) * 5)
    ch, cw = 48, 80
    y0, x0 = H - ch, W - cw

    def crop(t):
        return t.reshape(H, W)[y0:, x0:].cpu().numpy()

    cb = [_pattern_np(y0, H, x0, W, b) for b in range(5)]
    for b in range(5):
        assert np.array_equal(crop(bands[b]), cb[b])
    cn = []
    for b in range(5):
        lo, hi = lohi[b]
        c = np.minimum(np.maximum(cb[b], lo), hi)
        cn.append(((c - lo) / (hi - lo + np.float32(1e-10))).astype(np.float32))
        assert np.array_equal(crop(norms[b]), cn[b]), b
    bl, g, r, nir, s = cn
    want = {"ndvi": oracle.calculate_ndvi(nir, r), "evi": oracle.calculate_evi(nir, r, bl), "msavi": oracle.calculate_msavi(nir, r),
            "ndwi": oracle.calculate_ndwi(g, nir), "mndwi": oracle.calculate_mndwi(g, s), "ndbi": oracle.calculate_ndbi(s, nir),
            "bsi": oracle.calculate_bsi(bl, r, nir, s)}
    for k, v in want.items():
        assert np.array_equal(crop(idx[k]), v), k
    nir2 = P.renormalize(ctx, norms[3])
    q = ctx.quantize_u8(nir2, 31.0)
    small, (oh, ow) = ctx.glcm(q, H, W, 32, 7, 1)
    assert (oh, ow) == (H - 6, W - 6)
    qc = q.reshape(H, W)[y0:, x0:].cpu().numpy()
    wantg = oracle.glcm_small_maps(qc, 32, 7, 1, mode=1)
    for gpl, k in zip(small, ["contrast", "dissimilarity", "homogeneity", "energy", "correlation"]):
        gc = gpl.reshape(oh, ow)[y0:, x0:].cpu().numpy()   # windows whose top-left corner lies in the crop
        assert np.array_equal(gc, wantg[k]), k
    # -- upsample: last 40 rows through the striped entry point == the same rows of the whole-plane call
    up = ctx.resize_bilinear(small[0], oh, ow, H, W)
    rows0 = H - 40
    s0 = oh - 48
    part = ctx.resize_bilinear_rows(small[0].reshape(oh, ow)[s0:].reshape(-1).contiguous(), oh - s0, ow, s0, oh, 40, W, rows0, H)
    assert torch.equal(part, up.reshape(H, W)[rows0:].reshape(-1))
    del up, part, idx, norms, nir2, q, small
    torch.cuda.empty_cache()
    # -- config 3 end to end
    k = 8
    labels, meta, planes = P.config3(ctx, bands, H, W, k, 7, 1, 3)
    labels2, meta2, _ = P.config3(ctx, bands, H, W, k, 7, 1, 3)
    assert torch.equal(labels, labels2) and meta["n_iter"] == meta2["n_iter"] and np.array_equal(meta["centers"], meta2["centers"])
    assert int(labels.min()) == 0 and int(labels.max()) == k - 1
    F = len(planes)
    scale = torch.tensor(meta["scale"], dtype=torch.float64, device=ctx.device)
    mn = torch.tensor(meta["min"], dtype=torch.float64, device=ctx.device)
    cnt = torch.bincount(labels.to(torch.int64), minlength=k).to(torch.float64)
    assert float(cnt.min()) > 0
    lab64 = labels.to(torch.int64)
    masks = [labels == j for j in range(k)]
    for f in (0, 7, 11, 14):
        # float32 steps of MinMaxScaler.transform as the library applies them, summed in float64
        xs = (planes[f] * scale[f].to(torch.float32) + mn[f].to(torch.float32)).to(torch.float64)
        zero = torch.zeros((), dtype=torch.float64, device=ctx.device)
        sums = torch.stack([torch.where(masks[j], xs, zero).sum() for j in range(k)])
        mean_f = (sums / cnt).cpu().numpy()
        # the last M-step precedes the final E-step, so a few pixels may have moved since: tolerance, not equality
        assert np.allclose(mean_f, meta["centers"][:, f], rtol=0, atol=1e-3), (f, np.abs(mean_f - meta["centers"][:, f]).max())
        del xs
    del masks
    sub = torch.randint(0, n, (200000,), device=ctx.device, generator=torch.Generator(device=ctx.device).manual_seed(5))
    Xs = torch.stack([(planes[f][sub] * scale[f].to(torch.float32) + mn[f].to(torch.float32)).to(torch.float64) for f in range(F)], 1)
    C = torch.tensor(meta["centers"], dtype=torch.float64, device=ctx.device)
    d = ((Xs[:, None, :] - C[None]) ** 2).sum(-1)
    mine = d[torch.arange(sub.numel(), device=ctx.device), lab64[sub]]
    assert float((mine - d.min(1).values).max()) <= 1e-5


def test_full_size_16384_window_ops_and_forest(ctx, oracle, golden_dir):
    """Config-5 kernels at the full size: 7x7 context mean, 5x5 std / morphological gradient, Sobel (global max) and the
    forest walk, each checked at the far corner against the oracle (window operators: on a crop with a 4-pixel margin,
    compared away from the artificial crop edge; the bottom / right edges are true image borders)."""
    import torch
    from rsseg import _lib as L
    H = W = 16384
    ch, cw, m = 64, 96, 4
    y0, x0 = H - ch, W - cw
    x = _pattern_gpu(torch, ctx.device, H, W, 3) / 255.0   # float32 in [0, 1]
    xc = (_pattern_np(y0, H, x0, W, 3) / np.float32(255.0)).astype(np.float32)

    def far(t):
        return t.reshape(H, W)[y0 + m:, x0 + m:].cpu().numpy()

    assert np.array_equal(far(x), xc[m:, m:])
    assert np.array_equal(far(ctx.box_mean(x, H, W, 7, L.BORDER_REFLECT)), oracle.box_mean(xc, 7, "reflect")[m:, m:])
    mean = oracle.box_mean(xc, 5, "reflect101")
    var = oracle.box_mean(xc * xc, 5, "reflect101") - mean * mean
    var[var < 0] = 0
    assert np.array_equal(far(ctx.local_std(x, H, W, 5)), np.sqrt(var)[m:, m:])
    q = ctx.quantize_u8(x, 255.0)
    qc = (xc * 255).astype(np.uint8)
    assert np.array_equal(far(q), qc[m:, m:])
    assert np.array_equal(far(ctx.morph_gradient(q, H, W, 5)), oracle.morph_gradient_u8(qc, 5)[m:, m:])
    # Sobel: the normalising maximum is global; rebuild the oracle's crop result with the device's maximum
    sob = ctx.sobel_mag(q, H, W)
    p = np.pad(qc.astype(np.int32), 1, mode="reflect")
    gx = (p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])
    gy = (p[2:, :-2] - p[:-2, :-2]) + 2 * (p[2:, 1:-1] - p[:-2, 1:-1]) + (p[2:, 2:] - p[:-2, 2:])
    sx, sy = gx.astype(np.float32) / np.float32(255.0), gy.astype(np.float32) / np.float32(255.0)
    mag = np.sqrt(sx * sx + sy * sy)
    mx = np.float32(float((sob.max())))  # == 1.0 after normalisation unless the plane is flat
    assert mx == np.float32(1.0)
    ratio = far(sob) / np.where(mag[m:, m:] > 0, mag[m:, m:], 1).astype(np.float32)
    nz = mag[m:, m:] > 0
    assert nz.any() and np.allclose(ratio[nz], ratio[nz][0], rtol=1e-6)  # one common divisor (global max + 1e-10)
    del sob, q, x
    torch.cuda.empty_cache()
    # forest: bundled model on 19 planes, far end of the pixel range against the oracle's walk
    f = dict(np.load(os.path.join(golden_dir, "rf_samples_model_flat.npz")))
    ctx.forest_load(f)
    g = torch.Generator(device=ctx.device).manual_seed(11)
    planes = [torch.rand(H * W, device=ctx.device, generator=g) for _ in range(19)]
    out = ctx.forest_predict(planes)
    tail = 4099
    want = oracle.rf_predict_planes(f, [pl[-tail:].cpu().numpy() for pl in planes])
    assert np.array_equal(out[-tail:].cpu().numpy(), want)
    head = oracle.rf_predict_planes(f, [pl[:tail].cpu().numpy() for pl in planes])
    assert np.array_equal(out[:tail].cpu().numpy(), head)


def test_raw_band_pca_and_fused_quantise_equal_two_step_forms(ctx, crop, oracle):
    """rsseg_pca_fit_transform_raw_f32 on raw bands + percentiles == rsseg_pca_fit_transform_f32 on the normalised planes,
    and rsseg_normalize_quantize_u8 == normalise then quantise (bit for bit)."""
    import torch
    from rsseg import pipeline as P
    bands = [dev(ctx, b) for b in crop["bands"]]
    qb = P.band_quantile_bundles(ctx, bands)
    lohi = np.array([[q["lo"], q["hi"]] for q in qb], np.float32)
    stats = [(q["center"], q["scale"]) for q in qb]
    normd = [ctx.normalize(b, float(lohi[i, 0]), float(lohi[i, 1])) for i, b in enumerate(bands)]
    a, ra, ma = P.pca(ctx, normd, 3, True, None, stats)
    b, rb, mb = P.pca(ctx, bands, 3, True, None, stats, lohi=lohi)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert np.array_equal(ra, rb) and np.array_equal(ma["components"], mb["components"])
    c, rc, _ = P.pca(ctx, normd, 7, False)
    d, rd, _ = P.pca(ctx, bands, 7, False, lohi=lohi)
    assert all(torch.equal(x, y) for x, y in zip(c, d)) and np.array_equal(rc, rd)
    lo2, hi2 = float(qb[3]["lo2"]), float(qb[3]["hi2"])
    two = ctx.quantize_u8(ctx.normalize(normd[3], lo2, hi2), 31.0)
    one = ctx.normalize_quantize_u8(normd[3], lo2, hi2, 31.0)
    assert torch.equal(one, two)
    with pytest.raises(ValueError):
        P.pca(ctx, bands, 3, True, lohi=lohi)  # raw bands need the precomputed RobustScaler statistics


@pytest.mark.parametrize("n", [16 * 4096 + 0, 100003, 37])
def test_uint8_planes_equal_the_widened_float32_planes(ctx, oracle, n):
    """8-bit band planes inside K1 / K2 / K3 (1 byte per pixel instead of 4): order statistics, the seven indices (+ the
    normalised bands), and the PCA (fit on a sub-range with an odd offset, projection of everything) are bit for bit what
    the float32 entry points return on the widened planes — the 256-entry tables are filled with the float path's own
    operations.  Ragged lengths (n % 16, n % 4 != 0) included."""
    import torch
    from rsseg import pipeline as P
    rng = np.random.default_rng(n)
    raw = [np.clip(rng.normal(90 + 20 * b, 35, n), 0, 255).astype(np.uint8) for b in range(7)]
    raw[5][: n // 3] = 7                                   # a band with a huge tie
    d8 = [ctx.to_device(b) for b in raw]
    d32 = [ctx.to_device(b.astype(np.float32)) for b in raw]
    assert d8[0].dtype == torch.uint8
    ranks = sorted({0, n - 1, n // 2, (n - 1) // 2, int(0.02 * (n - 1)), int(0.98 * (n - 1)), n // 4, (3 * n) // 4})
    v8, nan8 = ctx.order_stats_multi(d8, [ranks] * 7)
    v32, nan32 = ctx.order_stats_multi(d32, [ranks] * 7)
    assert np.array_equal(v8, v32) and not nan8.any()
    for b in range(7):
        assert np.array_equal(v8[b], np.sort(raw[b])[ranks].astype(np.float32)), b
    q8, q32 = P.band_quantile_bundles(ctx, d8, n), P.band_quantile_bundles(ctx, d32, n)
    for a, b in zip(q8, q32):
        assert all(np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes() for k in a), (a, b)
    lohi = np.array([[q["lo"], q["hi"]] for q in q32], np.float32)
    for lh in (lohi, None):
        o8, n8 = ctx.spectral_indices(d8[:5], None if lh is None else lh[:5], want_norm=(True,) * 5)
        o32, n32 = ctx.spectral_indices(d32[:5], None if lh is None else lh[:5], want_norm=(True,) * 5)
        for a, b in zip(o8 + n8, o32 + n32):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    center = np.array([q["center"] for q in q32], np.float32)
    scale = np.array([q["scale"] for q in q32], np.float64)
    fit = (3, n - 8) if n > 64 else None
    r8 = ctx.pca_fit_transform(d8, center, scale, 3, lohi, fit=fit)
    r32 = ctx.pca_fit_transform(d32, center, scale, 3, lohi, fit=fit)
    for a, b in zip(r8[0], r32[0]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    for a, b in zip(r8[1:], r32[1:]):
        assert np.array_equal(a, b)
    r8n = ctx.pca_fit_transform(d8, None, None, 7)           # no RobustScaler, no normalisation: raw DN (range known: 0..255)
    r32n = ctx.pca_fit_transform(d32, None, None, 7)
    assert np.array_equal(r8n[1], r32n[1]) and all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(r8n[0], r32n[0]))


@pytest.mark.parametrize("n,u8", [(100003, False), (4096 * 16, True), (100003, True), (41, False)])
def test_fused_indices_pca_equals_the_two_separate_calls(ctx, oracle, n, u8):
    """rsseg_indices_pca_* (one Gram pass + one pass writing indices, normalised bands and components) against
    rsseg_spectral_indices_evi_* + rsseg_pca_fit_transform_ext_*: every plane, the model and the extrema tags bit for bit;
    float32 and uint8 bands, ragged lengths, a fit range with an odd offset, 3 and 7 components, non-default EVI coefficients."""
    import torch
    from rsseg import pipeline as P
    rng = np.random.default_rng(n + u8)
    raw = [np.clip(rng.normal(80 + 15 * b, 30, n), 0, 255).astype(np.uint8) for b in range(7)]
    if not u8:
        raw = [(b.astype(np.float32) + rng.random(n).astype(np.float32) * 0.7).astype(np.float32) for b in raw]   # general floats
    d = [ctx.to_device(b) for b in raw]
    qb = P.band_quantile_bundles(ctx, d, n)
    lohi = np.array([[q["lo"], q["hi"]] for q in qb], np.float32)
    center = np.array([q["center"] for q in qb], np.float32)
    scale = np.array([q["scale"] for q in qb], np.float64)
    fit = (5, n - 9) if n > 64 else None
    ctx.collect_minmax(True)
    try:
        for nc, evi in ((3, None), (7, (0.5, 5.0, 6.5, 2.0))):
            idx_s, norm_s = ctx.spectral_indices(d[:5], lohi[:5], want_norm=(True,) * 5, evi_coef=evi)
            tags_s = [t._rsseg_minmax for t in idx_s]
            pc_s, comp_s, ratio_s, mean_s, ev_s = ctx.pca_fit_transform(d, center, scale, nc, lohi, fit=fit)
            tags_s += [t._rsseg_minmax for t in pc_s]
            lo2, hi2 = float(qb[3]["lo2"]), float(qb[3]["hi2"])
            idx_f, norm_f, pc_f, comp_f, ratio_f, mean_f, ev_f = ctx.indices_pca(d, lohi, center, scale, nc, want_norm=(True,) * 5, fit=fit, evi_coef=evi,
                                                                                 quantize=(lo2, hi2, 31.0))
            tags_f = [t._rsseg_minmax for t in idx_f + pc_f]
            # the texture chain's input written by the same pass == the separate re-normalise + truncate kernel on the normalised NIR band
            assert torch.equal(ctx.last_quantized, ctx.normalize_quantize_u8(norm_s[3], lo2, hi2, 31.0))
            for a, b in zip(idx_s + norm_s + pc_s, idx_f + norm_f + pc_f):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32))
            assert np.array_equal(comp_s, comp_f) and np.array_equal(ratio_s, ratio_f) and np.array_equal(mean_s, mean_f) and np.array_equal(ev_s, ev_f)
            assert tags_s == tags_f
    finally:
        ctx.collect_minmax(False)


@pytest.mark.parametrize("n", [4099, 300001])
def test_gram_table_pass_for_byte_valued_float_planes(ctx, n):
    """float32 planes that the select has just seen to hold only the integers 0..255 take the Gram pass with 256-entry
    tables (k3_gram<NB, false, true>); the kernel verifies every value.  (1) the model equals the general kernel's bit for
    bit (the hint is dropped by a select on some other plane); (2) a STALE hint — planes changed after the select — costs
    a second pass, never a result; (3) so does a plane with values above 255 or a NaN."""
    import torch
    from rsseg import pipeline as P
    rng = np.random.default_rng(n)
    raw = [np.clip(rng.normal(80 + 15 * b, 30, n), 0, 255).astype(np.uint8).astype(np.float32) for b in range(7)]
    d = [ctx.to_device(b) for b in raw]
    other = ctx.to_device(rng.random(1000).astype(np.float32))

    def model(planes, hinted, fit=None):
        qb = P.band_quantile_bundles(ctx, planes, n)               # leaves the hint for these planes
        lohi = np.array([[q["lo"], q["hi"]] for q in qb], np.float32)
        center = np.array([q["center"] for q in qb], np.float32)
        scale = np.array([q["scale"] for q in qb], np.float64)
        if not hinted:
            ctx.order_stats(other, [0])                            # a select on another plane drops it
        ctx.prof_enable(True)
        ctx.prof_reset()
        out = ctx.pca_fit_transform(planes, center, scale, 3, lohi, fit=fit)
        _, launches = ctx.prof_get("gram")
        ctx.prof_enable(False)
        return out, launches, (lohi, center, scale)

    for fit in (None, (5, n - 9)):
        (pc_t, *m_t), l_t, _ = model(d, True, fit)
        (pc_g, *m_g), l_g, _ = model(d, False, fit)
        assert l_t == l_g                                          # one pass each (a head segment adds a launch to both)
        for x, y in zip(m_t, m_g):
            assert np.array_equal(x, y)
        for x, y in zip(pc_t, pc_g):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    # stale hint: the planes change between the select and the PCA
    qb = P.band_quantile_bundles(ctx, d, n)
    lohi = np.array([[q["lo"], q["hi"]] for q in qb], np.float32)
    center = np.array([q["center"] for q in qb], np.float32)
    scale = np.array([q["scale"] for q in qb], np.float64)
    d[2].add_(torch.rand(n, device=d[2].device) * 0.5)
    d[5][n // 2] = 300.0
    ctx.prof_enable(True)
    ctx.prof_reset()
    pc_s, *m_s = ctx.pca_fit_transform(d, center, scale, 3, lohi)
    _, l_s = ctx.prof_get("gram")
    ctx.prof_enable(False)
    ctx.order_stats(other, [0])
    ctx.prof_enable(True)
    ctx.prof_reset()
    pc_r, *m_r = ctx.pca_fit_transform(d, center, scale, 3, lohi)
    _, l_r = ctx.prof_get("gram")
    ctx.prof_enable(False)
    assert l_s == 2 * l_r                                          # the table pass, then the general one
    for x, y in zip(m_s, m_r):
        assert np.array_equal(x, y)
    for x, y in zip(pc_s, pc_r):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    # a NaN under a stale hint is still reported
    P.band_quantile_bundles(ctx, [ctx.to_device(b) for b in raw], n)
    d2 = [ctx.to_device(b) for b in raw]
    P.band_quantile_bundles(ctx, d2, n)
    d2[1][7] = float("nan")
    with pytest.raises(Exception, match="NaN"):
        ctx.pca_fit_transform(d2, center, scale, 3, lohi)


def test_config3_on_uint8_bands_equals_float32_bands(ctx, oracle):
    """The whole config-3 pipeline fed with uint8 band planes: labels, seeds, iteration count and feature planes equal
    those of the float32 planes."""
    import torch
    from rsseg import pipeline as P
    H, W = 160, 208
    r = oracle.synthetic_raster(H, W)
    d32 = [ctx.to_device(r[i].reshape(-1)) for i in range(7)]
    d8 = [ctx.to_device(r[i].reshape(-1).astype(np.uint8)) for i in range(7)]
    l32, m32, p32 = P.config3(ctx, d32, H, W, 8, 7, 1, 3)
    l8, m8, p8 = P.config3(ctx, d8, H, W, 8, 7, 1, 3)
    assert m8["n_iter"] == m32["n_iter"] and np.array_equal(m8["init_indices"], m32["init_indices"])
    assert torch.equal(l8, l32)
    for a, b in zip(p8, p32):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    s32, _ = P.feature_stack19(ctx, d32, H, W)
    s8, _ = P.feature_stack19(ctx, d8, H, W)
    for a, b in zip(s8, s32):
        assert torch.equal(a, b) or torch.equal(a.view(torch.int32), b.view(torch.int32))


def bits_equal(a, b):
    """Equal bit for bit (so that -0.0 != +0.0), NaNs at the same places (their payloads are not compared)."""
    a, b = np.ascontiguousarray(a, np.float32).reshape(-1), np.ascontiguousarray(b, np.float32).reshape(-1)
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a.view(np.int32)[~na], b.view(np.int32)[~na]))


def test_signed_zeros_follow_numpy(ctx, oracle):
    """Bands that hold -0.0 (np.round of a small negative number; a float raster can carry them, 8-bit digital numbers cannot):
    np.clip keeps a -0.0 that equals the lower bound, so robust_normalize returns -0.0 there and the ratio indices carry the sign
    on (found by tests/test_gpu_fuzz.py in round 4 — the oracle's np.maximum / np.minimum form lost it; the kernels had it right).
    Normalised planes and the seven indices of the product, separate kernels and the fused pass, against the oracle BIT for bit."""
    from rsseg import pipeline as P
    rng = np.random.default_rng(41)
    H, W = 60, 77
    bands = [np.clip(np.round(rng.normal(1.0, 2.0, (H, W))), 0, 6).astype(np.float32) for _ in range(7)]    # -0.0 where round(-0.3) landed
    assert sum(int(np.signbit(b[b == 0]).sum()) for b in bands) > 50
    norm = [oracle.robust_normalize(b) for b in bands]
    assert any(np.signbit(n[n == 0]).any() for n in norm)
    d = [dev(ctx, b) for b in bands]
    lohi = P.band_lohi(ctx, d, H * W)
    for i in range(7):
        assert bits_equal(host(ctx.normalize(d[i], float(lohi[i, 0]), float(lohi[i, 1]))), norm[i]), i
    b, g, r, n, s = norm[:5]
    want = [oracle.calculate_ndvi(n, r), oracle.calculate_evi(n, r, b), oracle.calculate_msavi(n, r), oracle.calculate_ndwi(g, n),
            oracle.calculate_mndwi(g, s), oracle.calculate_ndbi(s, n), oracle.calculate_bsi(b, r, n, s)]
    idx, _ = P.spectral_indices(ctx, d, lohi)
    for name, w in zip(P.INDEX_NAMES, want):
        assert bits_equal(host(idx[name]), w), name
    _, _, planes = P.config3(ctx, d, H, W, 4, 7, 1, 3)          # the fused index / PCA pass
    for i, w in enumerate(want):
        assert bits_equal(host(planes[i]), w), ("fused", i)


def test_normalise_division_is_the_ieee_quotient(ctx):
    """robust_normalize's (clip(x) - lo) / (hi - lo + 1e-10) on the device: bit-equal to NumPy's float32 quotient for
    ordinary, tiny, huge and degenerate ranges, and for numerators down to the denormals (no flush to zero, no
    reciprocal shortcut)."""
    rng = np.random.default_rng(11)
    n = 1 << 18
    f32 = np.float32
    cases = [(0.0, 255.0), (3.0, 141.0), (-17.5, 90.25), (0.0, 1.0), (5.0, 5.0), (0.0, 1e-20), (0.0, 1e30), (-1e25, 1e25),
             (0.0, float(np.nextafter(f32(2.0), f32(0.0)))),       # divisor with an all-ones significand
             (0.0, 3e-19), (1e-3, 1.0000001e-3), (0.0, 8.6e18)]
    for lo, hi in cases:
        lo, hi = f32(lo), f32(hi)
        span = float(hi) - float(lo)
        x = np.concatenate([
            (rng.random(n // 4) * span * 1.2 + float(lo) - 0.1 * span).astype(f32),            # across and beyond the range
            (float(lo) + rng.random(n // 4) * span * rng.choice([1e-3, 1e-9, 1e-20, 1e-33, 1e-38], n // 4)).astype(f32),  # just above lo
            rng.integers(0, 256, n // 4).astype(f32),                                            # DN values
            np.frombuffer(rng.integers(0, 2**32, n // 4, dtype=np.uint32).tobytes(), f32),       # any bit pattern (NaN, inf, denormals)
        ])
        with np.errstate(all="ignore"):
            want = (np.clip(x, lo, hi) - lo) / (hi - lo + f32(1e-10))
        assert want.dtype == np.float32
        got = ctx.normalize(dev(ctx, x), float(lo), float(hi)).cpu().numpy()
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (lo, hi, x[~same][:4], got[~same][:4], want[~same][:4])


def test_full_size_depth16_forest_vs_sklearn(ctx):
    """BASELINE config 5 at its real size with its real forest: RandomForestClassifier(100 trees, max_depth 16) fitted on
    the 19-feature stack (bench.fit_c5_forest, the same object the bench walks), the 16384 x 16384 stack classified on
    the GPU, 30 000 pixels drawn over the whole raster (the last rows included) against model.predict on the same rows."""
    import torch
    import bench
    from rsseg import pipeline as P
    H = W = 16384
    fm = bench.fit_c5_forest(torch, None, ctx.device, P, 0, 1, W)
    depths = [e.tree_.max_depth for e in fm["model"].estimators_]
    assert len(depths) == 100 and max(depths) == 16
    ctx.forest_load(fm["flat"])
    bands = bench.synth_rows(torch, ctx.device, W, 0, H)
    planes, _ = P.feature_stack19(ctx, bands, H, W)
    fp = P.stack19_forest_planes(ctx, planes)
    got = ctx.forest_predict(fp)
    rng = np.random.default_rng(16)
    idx = np.concatenate([rng.choice(H * W, 29000, replace=False), np.arange(H * W - 1000, H * W)])
    ti = torch.from_numpy(idx).to(ctx.device)
    X = np.stack([p[ti].cpu().numpy() for p in fp], 1)
    want = fm["model"].predict(X)
    assert np.array_equal(got[ti].cpu().numpy(), want)
    assert len(np.unique(want)) >= 6   # a real multi-class map, not one label
    del planes, fp, bands, got
    torch.cuda.empty_cache()
