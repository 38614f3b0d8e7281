"""CPU suite, part 1: the oracle restatement against the golden vectors produced by the reference
itself (oracle/gen_golden.py) and against the reference's committed artefacts."""
import json
import os

import numpy as np
import pytest

KEYS = ["ndvi", "evi", "msavi", "ndwi", "mndwi", "ndbi", "bsi"]


@pytest.fixture(scope="module")
def crop(golden_dir):
    return np.load(os.path.join(golden_dir, "crop96.npz"))


@pytest.fixture(scope="module")
def scene(golden_dir):
    return np.load(os.path.join(golden_dir, "scene_aa.npz"))


def _indices(O, norm):
    b, g, r, n, s = norm[:5]
    return {"ndvi": O.calculate_ndvi(n, r), "evi": O.calculate_evi(n, r, b), "msavi": O.calculate_msavi(n, r),
            "ndwi": O.calculate_ndwi(g, n), "mndwi": O.calculate_mndwi(g, s), "ndbi": O.calculate_ndbi(s, n),
            "bsi": O.calculate_bsi(b, r, n, s)}


def test_normalize_and_indices_bitexact(oracle, crop):
    norm = [oracle.robust_normalize(b) for b in crop["bands"]]
    for a, b in zip(norm, crop["norm"]):
        assert a.dtype == np.float32 and np.array_equal(a, b)
    idx = _indices(oracle, norm)
    for k in KEYS:
        assert idx[k].dtype == np.float32
        assert np.array_equal(idx[k], crop["idx_" + k]), k


def test_indices_nan_and_zero_denominator(oracle):
    a = np.array([[0.0, 0.0005, np.nan, 0.3]], np.float32)
    b = np.array([[0.0, 0.0004, 0.2, np.nan]], np.float32)
    out = oracle.calculate_ndvi(a, b)
    assert np.array_equal(out, np.zeros_like(out))  # den <= 0.001 or NaN -> 0


def test_pca_matches_reference(oracle, crop):
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):
        pcs, ratio, model = oracle.perform_pca(list(crop["norm"]))
        pcs3, ratio3, _ = oracle.perform_pca(list(crop["norm"]), n_components=3)
    assert np.allclose(np.stack(pcs), crop["pca7"], rtol=0, atol=1e-5)
    assert np.allclose(ratio, crop["pca7_ratio"], atol=1e-6)
    assert np.allclose(np.stack(pcs3), crop["pca3"], rtol=0, atol=1e-5)
    assert np.allclose(model["components"], crop["pca7_components"], atol=1e-5)


@pytest.mark.parametrize("k", [6, 7, 8])
def test_kmeans_idx7_labels_equal_sklearn(oracle, crop, k):
    planes = [crop["idx_" + n] for n in KEYS]
    labels, info = oracle.kmeans_fit_planes(planes, k)
    assert labels.dtype == np.int32
    assert np.array_equal(labels, crop[f"kmeans_idx7_k{k}"].reshape(-1))
    assert info["relocated"] == 0


@pytest.mark.parametrize("k", [6, 8])
def test_kmeans_stack19_f64_labels_equal_sklearn(oracle, crop, k):
    st = crop["stack19"]
    assert st.dtype == np.float64
    labels, _ = oracle.kmeans_fit_planes([st[:, :, i] for i in range(19)], k)
    assert np.array_equal(labels, crop[f"kmeans_stack19_k{k}"].reshape(-1))


def test_kmeans_default_selection_55_planes_equals_reference(oracle, golden_dir):
    """The reference's DEFAULT call (feature_keys_to_use=None -> every 2-D plane of a stage-2-shaped dictionary, 55 mixed
    float32 / float64 planes, extract.py:516-522, 568): the restatement reproduces the labels the reference function
    returned (tests/golden/crop96_stage2.npz, written by oracle/gen_golden.py), 0 mismatches for k = 5 (the default) and 8."""
    g = np.load(os.path.join(golden_dir, "crop96_stage2.npz"))
    d = {str(k): g[f"plane_{i:02d}"] for i, k in enumerate(g["keys"])}
    d["height"], d["width"] = int(g["height"]), int(g["width"])
    assert len(oracle.select_feature_planes(d, None)) == 55
    for k in (5, 8):
        lab, info = oracle.unsupervised_kmeans_classification(d, k, None)
        assert np.array_equal(lab, g[f"kmeans_auto_k{k}"]), k


def test_full_features_dict_key_order_is_the_reference_key_rule(oracle, golden_dir):
    """flatten_features_dict(full_features_dict(...)) names the 55 planes as the reference's normalize_features_structure
    named them when the fixture was written."""
    g = np.load(os.path.join(golden_dir, "crop96_stage2.npz"))
    keys = [str(k) for k in g["keys"]]
    assert keys[:9] == [f"all_extracted_features_dict_{n}" for n in ("ndvi", "evi", "msavi", "ndwi", "mndwi", "ndbi", "bsi", "pca_result_0", "pca_result_1")]
    assert keys[19] == "all_extracted_features_dict_lbp_feature" and keys[20] == "all_extracted_features_dict_multi_scale_features_mean_scale_1"
    assert keys[-1] == "all_extracted_features_dict_filter_features_sobel_mag"
    z = np.zeros((4, 5), np.float32)
    fd = {"ndvi": z, "pca_result": [z, z], "variance_ratio": np.zeros(2), "glcm_features": {"Contrast": z}}
    assert list(oracle.flatten_features_dict(fd)) == ["all_extracted_features_dict_ndvi", "all_extracted_features_dict_pca_result_0",
                                                      "all_extracted_features_dict_pca_result_1", "all_extracted_features_dict_glcm_features_contrast"]


def test_kmeans_nan_replaced_by_zero(oracle, crop):
    planes = [crop["idx_" + n] for n in KEYS]
    planes[0] = crop["kmeans_idx7_nan_input"]
    labels, _ = oracle.kmeans_fit_planes(planes, 6)
    assert np.array_equal(labels, crop["kmeans_idx7_nan_k6"].reshape(-1))


def test_kmeans_full_scene_vs_sklearn_near_ties_only(oracle, scene, golden_dir):
    """On the 600x600 scene sklearn's own float32 accumulation order moves ~1e-4 of the labels
    (SURVEY.md §7); every disagreement must be a near-tie between the two centres involved."""
    ref = np.load(os.path.join(golden_dir, "scene_aa_ref_outputs.npz"))
    bands = oracle.stage1_preprocess(scene["dn"])
    norm = [oracle.robust_normalize(b) for b in bands]
    idx = _indices(oracle, norm)
    planes = [idx[n] for n in KEYS]
    report = json.load(open(os.path.join(golden_dir, "PIN_REPORT.json")))
    for k in (6, 8):
        labels, info = oracle.kmeans_fit_planes(planes, k)
        refl = ref[f"kmeans_idx7_k{k}"].reshape(-1).astype(np.int32)
        bad = np.nonzero(labels != refl)[0]
        assert bad.size <= report[f"kmeans_scene_idx7_k{k}"]["mismatch"]   # 47 (k=6), 66 (k=8) of 360 000 when the goldens were made
        X = np.stack([p.reshape(-1) for p in planes], 1).astype(np.float64)
        Xs = X * info["scale"] + info["min"] - info["mean"]  # centres are in the centred space
        C = info["centers"]
        d = ((Xs[bad, None, :] - C[None, :, :]) ** 2).sum(-1)
        gap = np.abs(d[np.arange(bad.size), labels[bad]] - d[np.arange(bad.size), refl[bad]])
        assert gap.max() < 2e-3, gap.max()


def test_percentiles_match_reference(oracle, scene, golden_dir):
    ref = np.load(os.path.join(golden_dir, "scene_aa_ref_outputs.npz"))
    bands = oracle.stage1_preprocess(scene["dn"])
    got = np.array([[np.percentile(b, 2), np.percentile(b, 98)] for b in bands], np.float32)
    assert np.array_equal(got, ref["percentiles_2_98"])


def test_rf_walk_matches_sklearn(oracle, crop, golden_dir):
    f = dict(np.load(os.path.join(golden_dir, "rf_samples_model_flat.npz")))
    X = crop["rf_X"]
    out = oracle.rf_predict_planes(f, [X[:, i] for i in range(19)])
    assert np.array_equal(out, crop["rf_pred_image"].reshape(-1))
    Xn = crop["rf_X_nan"]
    out = oracle.rf_predict_planes(f, [Xn[:, i] for i in range(19)])
    assert np.array_equal(out, crop["rf_pred_nan_native"].reshape(-1))
    out0 = oracle.rf_predict_planes(f, [np.nan_to_num(Xn[:, i], nan=0.0) for i in range(19)])
    assert np.array_equal(out0, crop["rf_pred_nan_zeroed"].reshape(-1))


def test_glcm_pair_form_equals_literal_form(oracle):
    rng = np.random.default_rng(7)
    q = rng.integers(0, 32, (40, 45)).astype(np.uint8)
    q[:10, :10] = 5  # constant windows: correlation = 1 branch
    for win, step in ((7, 1), (7, 7), (21, 21), (5, 3)):
        a = oracle.glcm_small_maps(q, 32, win, step, mode=0)
        b = oracle.glcm_small_maps(q, 32, win, step, mode=1)
        for k in a:
            assert np.allclose(a[k], b[k], rtol=1e-6, atol=1e-7), (k, win, step)


# The example of scikit-image's graycomatrix / graycoprops docstrings (skimage/feature/texture.py; the library itself is
# not installed): co-occurrence counts for distance 1 at 0, 45, 90, 135 degrees of a 4 x 4 image with 4 levels, and the
# contrast of the symmetric, normalised matrices at distance 1 for 0 / 90 degrees (0.58333333, 1.0).
SKIMAGE_DOC_IMAGE = np.array([[0, 0, 1, 1], [0, 0, 1, 1], [0, 2, 2, 2], [2, 2, 3, 3]], np.uint8)
SKIMAGE_DOC_COUNTS = [
    [[2, 2, 1, 0], [0, 2, 0, 0], [0, 0, 3, 1], [0, 0, 0, 1]],
    [[1, 1, 3, 0], [0, 1, 1, 0], [0, 0, 0, 2], [0, 0, 0, 0]],
    [[3, 0, 2, 0], [0, 2, 2, 0], [0, 0, 1, 2], [0, 0, 0, 0]],
    [[2, 0, 0, 0], [1, 1, 2, 0], [0, 0, 2, 1], [0, 0, 0, 0]],
]


SKIMAGE_TEST_PROPS = [("homogeneity", 0.80833333, 5e-9), ("energy", 0.38188131, 5e-9), ("correlation", 0.71953255, 5e-9),
                      ("dissimilarity", 0.418, 2e-3)]


def test_glcm_matches_skimage_docstring_vectors(oracle):
    """Pins the angle convention (offsets (0,1), (1,1), (1,0), (1,-1)), the pair loop and the contrast formula of the
    GLCM restatement to the vectors scikit-image publishes; then the windowed mode 0 / mode 1 means over the four
    angles must equal the mean of these per-angle values."""
    per_angle = []
    for a in range(4):
        counts, props = oracle.glcm_angle(SKIMAGE_DOC_IMAGE, 4, a, symmetric=False)
        assert np.array_equal(counts, np.array(SKIMAGE_DOC_COUNTS[a], np.uint32)), a
        sym, _ = oracle.glcm_angle(SKIMAGE_DOC_IMAGE, 4, a, symmetric=True)
        assert np.array_equal(sym, counts + counts.T)
        per_angle.append(props)
    assert abs(per_angle[0]["contrast"] - 0.58333333) < 5e-9 and per_angle[2]["contrast"] == 1.0
    # the known answers scikit-image's own test-suite asserts for this image at distance 1, angle 0, symmetric + normed
    # (skimage/feature/tests/test_texture.py: test_homogeneity 0.80833333, test_energy 0.38188131, test_correlation
    # 0.71953255, test_dissimilarity 0.418 to three decimals): the other properties of the restatement, pinned the same way
    for key, want, tol in SKIMAGE_TEST_PROPS:
        assert abs(per_angle[0][key] - want) < tol, (key, per_angle[0][key])
    # hand values of the other two angles from the published matrices: sum G (i-j)^2 / sum G = 16/9 and 4/9
    assert abs(per_angle[1]["contrast"] - 16 / 9) < 1e-15 and abs(per_angle[3]["contrast"] - 4 / 9) < 1e-15
    for mode in (0, 1):
        maps = oracle.glcm_small_maps(SKIMAGE_DOC_IMAGE, 4, 4, 1, mode=mode)
        for k in maps:
            want = np.float32(sum(p[k] for p in per_angle) / 4.0)
            assert maps[k].shape == (1, 1) and abs(float(maps[k][0, 0]) - float(want)) <= 1e-6 * max(1.0, abs(float(want))), (mode, k)


def test_class_map_end_to_end(oracle, scene, golden_dir):
    """The reference's committed output/class_map.npy = bundled forest applied to the 19-feature stack
    of the bundled scene: pins every feature stage (cv2 / skimage restatements included) and the walk."""
    f = dict(np.load(os.path.join(golden_dir, "rf_samples_model_flat.npz")))
    bands = oracle.stage1_preprocess(scene["dn"])
    _, hier = oracle.run_feature_extraction_stage(bands)
    assert hier["all"].shape == (600, 600, 19) and hier["all"].dtype == np.float64
    cm = oracle.predict_image(f, hier["all"])
    assert cm.dtype == np.int64
    agree = float(np.mean(cm == scene["class_map"]))
    assert agree >= 0.999, agree
    for (x, y), lab in zip(scene["sample_coords"], scene["sample_labels"]):
        assert cm[y, x] == lab
        assert scene["roi_mask"][y, x] == lab


def test_morphology_and_laplacian_restatements_vs_scipy(oracle):
    """cv2 is absent here, so the uint8 window operators of calculate_morphological_features / calculate_filter_responses
    are pinned against an independent implementation: scipy.ndimage with the equivalent border rules
    (constant border that never wins for erode / dilate; mode='mirror' = BORDER_REFLECT_101 for the Laplacian)."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (41, 57)).astype(np.uint8)
    u8[5:15, 5:15] = 200
    for k in (3, 5, 7):
        er = ndi.minimum_filter(u8, size=k, mode="constant", cval=255)
        di = ndi.maximum_filter(u8, size=k, mode="constant", cval=0)
        assert np.array_equal(oracle.morph_u8(u8, k, "erosion"), er)
        assert np.array_equal(oracle.morph_u8(u8, k, "dilation"), di)
        assert np.array_equal(oracle.morph_u8(u8, k, "opening"), ndi.maximum_filter(er, size=k, mode="constant", cval=0))
        assert np.array_equal(oracle.morph_u8(u8, k, "closing"), ndi.minimum_filter(di, size=k, mode="constant", cval=255))
        assert np.array_equal(oracle.morph_u8(u8, k, "gradient"), di - er)
    band = rng.random((41, 57)).astype(np.float32)
    q = oracle.to_u8(oracle.robust_normalize(band)).astype(np.float32)
    lap = ndi.correlate(q, np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]], np.float32), mode="mirror") / np.float32(255.0)
    want = (lap - lap.min()) / (lap.max() - lap.min() + 1e-10)
    got = oracle.laplacian_feature(band)
    assert got.dtype == np.float32 and want.dtype == np.float32
    assert np.array_equal(got, want)
    feats = oracle.calculate_morphological_features(band)
    assert len(feats) == 15 and all(v.dtype == np.float64 for v in feats.values())
    assert np.all(feats["opening_5"] <= feats["closing_5"])


def test_cv2_skimage_restatements_known_answers(oracle):
    """OpenCV and scikit-image are not installed and the reference holds no per-stage vectors, so the restatements of
    cv2.resize / boxFilter / GaussianBlur / Sobel and skimage local_binary_pattern / rank.entropy are pinned to values
    that follow BY HAND from the published definitions (conventions: half-pixel centres with edge clamping; BORDER_REFLECT_101;
    the fixed 8-bit kernel [1 4 6 4 1] / 16 of a 5-tap GaussianBlur with sigma 0; LBP 'uniform' = number of ones of a
    pattern with <= 2 transitions, neighbours sampled bilinearly, >= the centre, zero outside the image; Shannon entropy
    in bits over the disk footprint)."""
    # cv2.resize, INTER_LINEAR: fx = (dx + 0.5) * scale - 0.5, clamped at the edges
    assert np.array_equal(oracle.resize_bilinear(np.array([[0, 1]], np.float32), 1, 4), np.array([[0, 0.25, 0.75, 1]], np.float32))
    want = np.add.outer(np.array([0, 0.5, 1.5, 2], np.float32), np.array([0, 0.25, 0.75, 1], np.float32))
    assert np.array_equal(oracle.resize_bilinear(np.array([[0, 1], [2, 3]], np.float32), 4, 4), want)
    assert np.array_equal(oracle.resize_bilinear(want, 4, 4), want)                      # identity size: the identity
    # cv2.boxFilter / blur 3 x 3 on a ramp, BORDER_REFLECT_101: column 0 averages (x1, x0, x1)
    ramp = np.tile(np.arange(12, dtype=np.float32), (12, 1))
    bm = oracle.box_mean(ramp, 3, "reflect101")
    assert abs(bm[5, 0] - 2 / 3) < 1e-6 and np.allclose(bm[5, 1:11], np.arange(1, 11)) and abs(bm[5, 11] - (10 + 11 + 10) / 3) < 1e-5
    # cv2.GaussianBlur(u8, (5, 5), 0): impulse response = round(255 * outer([1 4 6 4 1], [1 4 6 4 1]) / 256)
    imp = np.zeros((9, 9), np.uint8)
    imp[4, 4] = 255
    k = np.array([1, 4, 6, 4, 1])
    assert np.array_equal(oracle.gaussian_blur_u8(imp, 5)[2:7, 2:7], np.round(255 * np.outer(k, k) / 256).astype(np.uint8))
    assert oracle.gaussian_blur_u8(np.full((9, 9), 77, np.uint8), 15).min() == 77 == oracle.gaussian_blur_u8(np.full((9, 9), 77, np.uint8), 15).max()
    # cv2.Sobel 3 x 3 on a unit ramp in x: gx = (1 + 2 + 1) * 2 = 8 everywhere inside, gy = 0; REFLECT_101 makes the border columns 0
    r8 = np.tile(np.arange(0, 40, dtype=np.float32), (8, 1))
    p = np.pad(r8, 1, mode="reflect")
    gx = (p[0:8, 2:] - p[0:8, :-2]) + 2 * (p[1:9, 2:] - p[1:9, :-2]) + (p[2:10, 2:] - p[2:10, :-2])
    assert np.all(gx[:, 1:-1] == 8) and np.all(gx[:, 0] == 0) and np.all(gx[:, -1] == 0)
    sm = oracle.sobel_mag_feature(r8)        # the feature: magnitude / max after the stage's own normalisation -> 1 inside, 0 at the border columns
    assert np.all(sm[:, 0] == 0) and np.all(sm[:, -1] == 0) and sm.max() == pytest.approx(1.0, abs=1e-9)
    # skimage.feature.local_binary_pattern(P = 24, R = 3, 'uniform'): a flat interior has all 24 neighbours >= the centre
    lb = oracle.lbp_uniform(np.full((16, 16), 7, np.uint8))
    assert np.all(lb[3:13, 3:13] == 24)
    # skimage.filters.rank.entropy(disk(1)): five pixels; four of one value and one of another -> H(0.8, 0.2) bits
    half = np.zeros((11, 11), np.uint8)
    half[:, 6:] = 9
    en = oracle.rank_entropy(half, 1)
    h = -(0.8 * np.log2(0.8) + 0.2 * np.log2(0.2))
    assert np.all(en[5, :5] == 0) and np.all(en[5, 7:] == 0) and abs(en[5, 5] - h) < 1e-6 and abs(en[5, 6] - h) < 1e-6
    assert oracle.disk(1).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]] and int(oracle.disk(3).sum()) == 29 and int(oracle.disk(5).sum()) == 81


def test_otsu_fill_holes_and_ellipse_restatements_known_answers(oracle):
    """threshold_segmentation(otsu=True) and the even-kernel branch of advanced_post_processing (extract.py:358-371, 314-316).
    cv2 is absent: the Otsu level is pinned to (a) the textbook definition evaluated independently — the level maximising the
    between-class variance w0 * w1 * (mu0 - mu1)^2, the FIRST such level as OpenCV scans upwards — and (b) the known answer for a
    two-valued image (every level between the two values separates the same classes, so the lower value is returned).
    binary_fill_holes is scipy itself; the elliptical element is pinned to the two matrices cv2's documentation shows for
    MORPH_ELLIPSE (3 x 3: the cross; 5 x 5: the square without its corner pairs) and to its symmetry at 7 / 9 / 11."""
    rng = np.random.default_rng(3)
    two = np.where(rng.random((40, 50)) < 0.3, 200, 10).astype(np.uint8)
    assert oracle.otsu_level_u8(two) == 10
    for trial in range(6):
        a = np.clip(np.where(rng.random((64, 64)) < 0.4, rng.normal(60, 12, (64, 64)), rng.normal(170, 25, (64, 64))), 0, 255).astype(np.uint8)
        h = np.bincount(a.ravel(), minlength=256).astype(np.float64) / a.size
        lv = np.arange(256.0)
        w0 = np.cumsum(h)
        m0 = np.cumsum(h * lv)
        w1 = 1 - w0
        with np.errstate(divide="ignore", invalid="ignore"):
            sig = np.where((w0 > 1e-6) & (w1 > 1e-6), w0 * w1 * (m0 / w0 - (m0[-1] - m0) / w1) ** 2, 0)
        best = int(np.argmax(sig))
        got = oracle.otsu_level_u8(a)
        assert sig[got] >= sig[best] * (1 - 1e-12) and abs(got - best) <= 1, (trial, got, best)
        assert 75 < got < 150                                            # between the two modes
    img = rng.normal(0.2, 0.05, (30, 30)).astype(np.float32)
    img[:, 15:] += 0.5
    m = oracle.threshold_segmentation(img, None, otsu=True)
    assert m.dtype == np.uint8 and m[:, :15].sum() == 0 and m[:, 15:].all()
    assert np.array_equal(oracle.threshold_segmentation(img, None, above=False, otsu=True), 1 - m)
    flat = np.full((5, 5), 0.3, np.float32)
    assert not oracle.threshold_segmentation(flat, None, otsu=True).any() and oracle.threshold_segmentation(flat, None, above=False, otsu=True).all()
    assert oracle.ellipse_element(3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert oracle.ellipse_element(5).tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    for k in (7, 9, 11):
        se = oracle.ellipse_element(k)
        assert np.array_equal(se, se[::-1]) and np.array_equal(se, se[:, ::-1]) and se[k // 2].all() and se[0].sum() == 1
    ring = np.zeros((20, 20), np.uint8)
    ring[5:15, 5:15] = 1
    ring[8:12, 8:12] = 0                    # a hole
    ring[0:3, 0:3] = 1
    ring[1, 1] = 0                          # a one-pixel hole in a block touching the border
    ring[17:20, 10:13] = 1
    ring[19, 11] = 0                        # a notch open to the border: not a hole
    out = oracle.advanced_post_processing(ring, min_area=0, smooth_kernel_size=4)
    assert out[8:12, 8:12].all() and out[1, 1] == 1 and out[19, 11] == 0
    assert np.array_equal(oracle.advanced_post_processing(ring, min_area=0, smooth_kernel_size=0), out)
    assert np.array_equal(oracle.advanced_post_processing(ring, min_area=0, smooth_kernel_size=4, fill_holes=False), ring)
    assert oracle.advanced_post_processing(ring, min_area=10, smooth_kernel_size=2).sum() == 100      # the two small blocks go


def test_pin_report_is_committed(golden_dir):
    rep = json.load(open(os.path.join(golden_dir, "PIN_REPORT.json")))
    assert rep["class_map_agreement"] >= 0.999
    assert rep["rf_mismatch"] == 0 and rep["robust_normalize_bitexact"]
    assert all(rep["indices_bitexact"].values())


def test_forest_walk_with_nan_threshold_splits_equals_sklearn(oracle):
    """A forest fitted on data with missing values can hold splits whose threshold is NaN (scikit-learn >= 1.4: the split only
    separates missing from non-missing values; `x <= NaN` is false, so non-missing values go right).  The oracle's walk follows
    Tree._apply_dense literally and must agree with model.predict on rows with and without NaNs."""
    from sklearn.ensemble import RandomForestClassifier
    found = 0
    for seed, ncls, F in ((0, 3, 5), (1, 13, 19), (2, 8, 55), (3, 9, 19), (4, 4, 3), (5, 20, 7)):
        rng = np.random.default_rng(500 + seed)
        Xtr = rng.random((30, F)).astype(np.float32)
        Xtr[rng.random((30, F)) < 0.08] = np.nan
        ytr = rng.integers(0, ncls, 30) * 7 - 20
        model = RandomForestClassifier(n_estimators=25, max_depth=6, random_state=seed, n_jobs=1).fit(Xtr, ytr)
        found += sum(int(np.isnan(e.tree_.threshold[e.tree_.children_left >= 0]).sum()) for e in model.estimators_)
        f = oracle.flatten_forest(model)
        X = rng.random((2000, F)).astype(np.float32)
        X[rng.random((2000, F)) < 0.05] = np.nan
        X[:500] = np.nan_to_num(X[:500], nan=0.5)
        assert np.array_equal(oracle.rf_predict_planes(f, [np.ascontiguousarray(X[:, i]) for i in range(F)]), model.predict(X)), seed
    assert found > 0
