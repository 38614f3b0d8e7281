"""GPU suite: the NumPy-in / NumPy-out mirror of the reference's modules (same names, signatures, dtypes,
error behaviour) against the golden vectors produced by the reference functions themselves."""
import inspect
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = ["ndvi", "evi", "msavi", "ndwi", "mndwi", "ndbi", "bsi"]


@pytest.fixture(scope="module")
def crop(golden_dir):
    return np.load(os.path.join(golden_dir, "crop96.npz"))


def test_indices_module_matches_reference_functions(ctx, crop):
    from modules.features import indices as I
    for i in range(7):
        out = I.robust_normalize(crop["bands"][i])
        assert out.dtype == np.float32 and np.array_equal(out, crop["norm"][i])
    b, g, r, n, s = crop["norm"][:5]
    got = {"ndvi": I.calculate_ndvi(n, r), "evi": I.calculate_evi(n, r, b), "msavi": I.calculate_msavi(n, r),
           "ndwi": I.calculate_ndwi(g, n), "mndwi": I.calculate_mndwi(g, s), "ndbi": I.calculate_ndbi(s, n),
           "bsi": I.calculate_bsi(b, r, n, s)}
    for k in KEYS:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], crop["idx_" + k]), k
    pcs, ratio, model = I.perform_pca(list(crop["norm"]), n_components=3)
    assert len(pcs) == 3 and pcs[0].shape == (96, 96) and pcs[0].dtype == np.float32
    assert np.allclose(np.stack(pcs), crop["pca3"], rtol=0, atol=1e-5)
    assert np.allclose(model.transform(np.zeros((2, 7), np.float32)).shape, (2, 3))
    fd = dict(got)
    fd["pca_result"] = pcs
    l1 = I.prepare_level_1_features(fd)
    assert l1.shape == (96, 96, 7) and l1.dtype == np.float32
    assert I.calculate_evi(n, r, b, L=2).shape == n.shape   # any coefficients (test_evi_coefficients_... checks the values)


def test_float64_bands_are_refused_by_name(ctx, crop):
    """NumPy would carry a float64 band through the reference's band-level functions in float64; the kernels behind the mirror
    compute in float32 (the dtype scripts/2:156 gives every band).  Narrowing silently would return different values under the
    reference's name: the mirror refuses the dtype instead, and says what to do."""
    from modules.features import indices as I
    from rsseg.runtime import RssegUnsupported
    b64 = crop["bands"][3].astype(np.float64)
    for call in (lambda: I.robust_normalize(b64), lambda: I.calculate_ndvi(b64, b64), lambda: I.perform_pca([b64, b64]),
                 lambda: I.calculate_glcm_features(b64), lambda: I.calculate_filter_responses(b64)):
        with pytest.raises(RssegUnsupported, match="astype"):
            call()
    assert I.robust_normalize(b64.astype(np.float32)).dtype == np.float32
    with pytest.raises(RssegUnsupported, match="astype"):
        I.add_spatial_context(np.stack([b64, b64], axis=-1))
    from modules.features import extract as E
    f64 = dict(ndvi=b64, ndbi=b64, ndwi=b64, height=96, width=96)
    with pytest.raises(RssegUnsupported, match="threshold_segmentation"):
        E.rule_based_classification(f64)
    assert E.threshold_segmentation(b64, 0.2).dtype == np.uint8            # the float64 comparison exists where the reference has it


def test_texture_feature_dicts_match_oracle(ctx, crop, oracle):
    """calculate_morphological_features / calculate_multi_scale_features / calculate_filter_responses of the mirror:
    member names, dtypes and values of the members that are produced (reference indices.py:401-482, 519-562)."""
    from modules.features import indices as I
    nir = crop["norm"][3]
    mf = I.calculate_morphological_features(nir)
    want = oracle.calculate_morphological_features(nir)
    assert sorted(mf) == sorted(want) and len(mf) == 15
    for k in want:
        assert mf[k].dtype == np.float64 and np.array_equal(mf[k], want[k]), k
    ms = I.calculate_multi_scale_features(nir)
    assert ({f"{m}_scale_{k}" for m in ("mean", "variance", "std_dev") for k in (1, 3, 5, 7)} | {f"entropy_scale_{k}" for k in (1, 3, 5)}) == set(ms)
    assert np.array_equal(ms["std_dev_scale_5"], oracle.std_dev_feature(nir, 5))
    assert np.array_equal(ms["variance_scale_3"], oracle.variance_feature(nir, 3))
    assert not ms["variance_scale_1"].any() and ms["variance_scale_1"].dtype == np.float32
    fr = I.calculate_filter_responses(nir)
    assert np.array_equal(fr["sobel_mag"], oracle.sobel_mag_feature(nir))
    assert np.array_equal(fr["laplacian"], oracle.laplacian_feature(nir)) and fr["laplacian"].dtype == np.float32


@pytest.mark.parametrize("shape", [(1, 1), (2, 2), (1, 9), (3, 3), (9, 1), (4, 17)])
def test_window_operators_on_rasters_smaller_than_their_windows(ctx, oracle, shape):
    """Rasters smaller than the 3 / 5 / 7-pixel windows (the border rule then reflects more than once — cv2's
    borderInterpolate loops until the index is inside, like NumPy's 'reflect'): every member against the oracle."""
    from modules.features import indices as I
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    band = rng.random(shape).astype(np.float32)
    mf = I.calculate_morphological_features(band)
    want = oracle.calculate_morphological_features(band)
    for k in want:
        assert np.array_equal(mf[k], want[k]), (shape, k)
    ms = I.calculate_multi_scale_features(band)
    for sc in (3, 5, 7):
        assert np.array_equal(ms[f"std_dev_scale_{sc}"], oracle.std_dev_feature(band, sc)), (shape, sc)
        assert np.array_equal(ms[f"variance_scale_{sc}"], oracle.variance_feature(band, sc)), (shape, sc)
    fr = I.calculate_filter_responses(band)
    assert np.array_equal(fr["sobel_mag"], oracle.sobel_mag_feature(band), equal_nan=True), shape
    assert np.array_equal(fr["laplacian"], oracle.laplacian_feature(band), equal_nan=True), shape
    ctxm = I.add_spatial_context(np.stack([band] * 2, axis=-1), 7)
    assert ctxm.dtype == np.float64 and np.array_equal(ctxm, oracle.add_spatial_context(np.stack([band] * 2, axis=-1), 7)), shape


def test_stage_takes_uint8_bands_like_float32_bands(ctx, crop):
    """8-bit rasters cross PCIe as one byte per pixel and are widened on the device (Context.upload_f32): the stage's
    outputs equal those of the same bands handed over as float32, bit for bit."""
    import torch
    from rsseg import stages
    b32 = [np.asarray(b, np.float32) for b in crop["bands"]]
    assert all(np.array_equal(b, np.round(b)) and b.min() >= 0 and b.max() <= 255 for b in b32)
    b8 = [b.astype(np.uint8) for b in b32]
    assert torch.equal(ctx.upload_f32(b8[0]), ctx.to_device(b32[0].reshape(-1)))
    fd32, h32 = stages.run_feature_extraction_stage(b32)
    fd8, h8 = stages.run_feature_extraction_stage(b8)
    assert np.array_equal(h32["all"], h8["all"], equal_nan=True)
    assert np.array_equal(fd32["ndvi"], fd8["ndvi"]) and np.array_equal(fd32["lbp_feature"], fd8["lbp_feature"])


def test_stage_function_layout_and_files(ctx, crop, tmp_path, oracle):
    from rsseg import stages
    fd, hier = stages.run_feature_extraction_stage(list(crop["bands"]))
    assert set(hier) == {"level_1", "level_2", "all"}
    assert hier["all"].shape == (96, 96, 19) and hier["all"].dtype == np.float64
    assert hier["level_1"].shape == (96, 96, 14) and hier["level_2"].shape == (96, 96, 5)
    _, want = oracle.run_feature_extraction_stage(list(crop["bands"]))
    for c in range(19):
        tol = 2e-4 if c in (6, 13) else 1e-5
        assert np.allclose(hier["all"][:, :, c], want["all"][:, :, c], rtol=0, atol=tol), c
    # the dict members around the stack (indices.py:401-482, 519-562)
    assert len(fd["morphological_features"]) == 15 and len(fd["multi_scale_features"]) == 15   # + entropy_scale_1 / 3 / 5
    assert np.array_equal(fd["morphological_features"]["gradient_5"], hier["all"][:, :, 16])
    assert np.array_equal(fd["multi_scale_features"]["std_dev_scale_5"].astype(np.float64), hier["all"][:, :, 17])
    assert np.array_equal(fd["morphological_features"]["closing_7"], oracle.calculate_morphological_features(crop["norm"][3])["closing_7"])
    assert set(fd["filter_features"]) == {"gaussian_5", "gaussian_15", "dog", "laplacian", "sobel_mag"}
    # every key scripts/2:62-106 puts into features_dict
    assert set(fd) == {"ndvi", "evi", "msavi", "ndwi", "mndwi", "ndbi", "bsi", "pca_result", "variance_ratio", "glcm_features", "lbp_feature",
                       "multi_scale_features", "morphological_features", "filter_features"}
    assert fd["lbp_feature"].dtype == np.float64 and fd["lbp_feature"].max() == 1.0
    paths = stages.save_feature_outputs(str(tmp_path), fd, hier, 96, 96)
    assert np.array_equal(np.load(paths["all"]), hier["all"])
    import pickle
    d = pickle.load(open(paths["pkl"], "rb"))
    assert set(d) == {"hierarchical_features", "all_extracted_features_dict", "dimensions", "geo_transform", "crs"}
    lab = stages.run_kmeans_stage(hier["all"], 7)
    assert lab.dtype == np.uint8 and lab.min() == 1 and lab.max() == 7


def test_kmeans_entry_point(ctx, crop):
    from modules.features import extract as E
    fd = {k: crop["idx_" + k] for k in KEYS}
    fd["height"], fd["width"] = 96, 96
    for k in (6, 8):
        out = E.unsupervised_kmeans_classification(fd, k, KEYS)
        assert out.dtype == np.int32 and out.shape == (96, 96)
        assert np.array_equal(out, crop[f"kmeans_idx7_k{k}"])
    out = E.unsupervised_kmeans_classification({"hierarchical_all": crop["stack19"], "height": 96, "width": 96}, 6, ["hierarchical_all"])
    assert np.array_equal(out, crop["kmeans_stack19_k6"])
    fdn = dict(fd)
    fdn["ndvi"] = crop["kmeans_idx7_nan_input"]
    assert np.array_equal(E.unsupervised_kmeans_classification(fdn, 6, KEYS), crop["kmeans_idx7_nan_k6"])
    with pytest.raises(ValueError):
        E.unsupervised_kmeans_classification({}, 5)
    with pytest.raises(ValueError):
        E.unsupervised_kmeans_classification(fd, 5, [])  # the reference raises on an empty key list too
    auto = E.unsupervised_kmeans_classification(fd, 6)  # feature_keys_to_use=None -> every 2-D plane
    assert auto.shape == (96, 96)
    sig = inspect.signature(E.unsupervised_kmeans_classification)
    assert list(sig.parameters) == ["features_dict", "n_clusters", "feature_keys_to_use"]
    assert sig.parameters["n_clusters"].default == 5


def test_forest_entry_points(ctx, crop, golden_dir):
    from modules import supervised_classifiers as S
    from modules.features import extract as E
    f = dict(np.load(os.path.join(golden_dir, "rf_samples_model_flat.npz")))
    X = crop["rf_X"].reshape(64, 64, 19).astype(np.float64)
    out = S.predict_image(f, X)
    assert out.dtype == np.int64 and np.array_equal(out, crop["rf_pred_image"])
    Xn = crop["rf_X_nan"].reshape(64, 64, 19).astype(np.float64)
    assert np.array_equal(S.predict_image(f, Xn), crop["rf_pred_nan_native"])
    assert np.array_equal(E.supervised_classification_predict(Xn, f), crop["rf_pred_nan_zeroed"])
    bad = S.predict_image(f, X[:, :, :5])  # wrong feature count: reported, zeros returned (reference behaviour)
    assert bad.shape == (64, 64) and not bad.any()
    with pytest.raises(ValueError):
        E.supervised_classification_predict(X[0], f)


def test_classification_stage_driver(ctx, crop, tmp_path, oracle):
    """run_classification_stage with the reference's signature (scripts/3_classification.py:267), called the way
    scripts/3:616-621 calls it.  On a stage-2 pickle the listed KMeans keys do not exist, so the selection is the automatic
    one the reference announces: every 2-D plane (55), 7 clusters, labels + 1."""
    from modules.features import extract as E
    from rsseg import stages
    sig = inspect.signature(stages.run_classification_stage)
    pos = [p for p in sig.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert [p.name for p in pos] == ["feature_file_path", "method", "output_dir", "use_hierarchical_all"]
    assert [p.default for p in pos[1:]] == ["rule_based", "segmentation_outputs", True]
    assert all(p.kind == p.KEYWORD_ONLY for n, p in sig.parameters.items() if n not in [q.name for q in pos])
    fd, hier = stages.run_feature_extraction_stage(list(crop["bands"]))
    # features_dict and its nested members in the reference's insertion order (scripts/2:62-106, indices.py:310-316, 421-440,
    # 463-480, 535-560): the order is the column order of the default KMeans selection
    want_order = oracle.full_features_dict(list(crop["bands"]), pca_result=fd["pca_result"])
    assert list(fd) == list(want_order)
    for k in ("glcm_features", "multi_scale_features", "morphological_features", "filter_features"):
        assert list(fd[k]) == list(want_order[k]), k
    paths = stages.save_feature_outputs(str(tmp_path), fd, hier, 96, 96)
    out = stages.run_classification_stage(paths["pkl"], method='kmeans', output_dir=str(tmp_path / "cls"), use_hierarchical_all=True)
    assert out.shape == (96, 96) and out.dtype == np.uint8 and out.min() == 1 and out.max() == 7
    assert np.array_equal(np.load(tmp_path / "cls" / "classification_kmeans.npy"), out)
    nf = E.normalize_features_structure(E.load_features(paths["pkl"]))
    keys2d = [k for k, v in nf.items() if isinstance(v, np.ndarray) and v.ndim == 2]
    assert len(keys2d) == 55 and keys2d == list(oracle.flatten_features_dict(want_order))
    assert np.array_equal(out, E.unsupervised_kmeans_classification(nf, 7, None) + 1)
    want, _ = oracle.unsupervised_kmeans_classification(nf, 7, None)          # the CPU restatement on the same 55 planes
    assert np.array_equal(out, want + 1)
    # explicit keys (keyword-only addition): the 19-feature stack
    out6 = stages.run_classification_stage(paths["pkl"], "kmeans", str(tmp_path / "cls6"), n_clusters=6, feature_keys=["hierarchical_features_all"])
    assert np.array_equal(out6, stages.run_kmeans_stage(hier["all"], 6))
    # default method is 'rule_based' (scripts/3:267)
    rb = stages.run_classification_stage(paths["pkl"], output_dir=str(tmp_path / "rb"))
    assert rb.dtype == np.uint8 and (tmp_path / "rb" / "classification_rule_based.npy").exists()


def test_classification_stage_forest_branch_loads_the_joblib_model(ctx, crop, tmp_path):
    """scripts/3:401-488, inference part: <output_dir>/random_forest_model.joblib is loaded when its n_features_in_ matches
    (:459-475); `use_hierarchical_all` picks the 19-feature stack, False stacks every 2-D plane (55, :425-437)."""
    import joblib
    from sklearn.ensemble import RandomForestClassifier
    from modules.features import extract as E
    from rsseg import stages
    fd, hier = stages.run_feature_extraction_stage(list(crop["bands"]))
    paths = stages.save_feature_outputs(str(tmp_path), fd, hier, 96, 96)
    rng = np.random.default_rng(5)
    idx = rng.choice(96 * 96, 600, replace=False)
    y = (fd["ndvi"].reshape(-1)[idx] > np.median(fd["ndvi"])).astype(np.int64) + 2 * (fd["ndwi"].reshape(-1)[idx] > 0) + 1
    X19 = hier["all"].reshape(-1, 19)
    m19 = RandomForestClassifier(n_estimators=20, random_state=0).fit(X19[idx], y)
    out_dir = tmp_path / "rf19"
    out_dir.mkdir()
    assert stages.run_classification_stage(paths["pkl"], "random_forest", str(out_dir)) is None     # no model yet: reported, None
    joblib.dump(m19, out_dir / stages.RF_MODEL_FILE)
    got = stages.run_classification_stage(paths["pkl"], "random_forest", str(out_dir), True)
    assert got.dtype == np.int64 and np.array_equal(got, m19.predict(X19).reshape(96, 96))
    # every 2-D plane: 55 features
    nf = E.normalize_features_structure(E.load_features(paths["pkl"]))
    X55 = np.stack([v for v in nf.values() if isinstance(v, np.ndarray) and v.ndim == 2], -1).reshape(-1, 55)
    m55 = RandomForestClassifier(n_estimators=20, random_state=1).fit(X55[idx], y)
    out55 = tmp_path / "rf55"
    out55.mkdir()
    joblib.dump(m55, out55 / stages.RF_MODEL_FILE)
    got55 = stages.run_classification_stage(paths["pkl"], "random_forest", str(out55), False)
    assert np.array_equal(got55, m55.predict(X55).reshape(96, 96))
    assert stages.run_classification_stage(paths["pkl"], "random_forest", str(out_dir), False) is None   # 19-feature model, 55 planes
    assert np.array_equal(stages.run_classification_stage(paths["pkl"], "random_forest", str(tmp_path / "kw"), classifier=m19), got)


def test_classification_stage_forest_branch_trains_from_a_label_raster(ctx, crop, tmp_path):
    """scripts/3:450-475: without a cached model the forest is fitted from the label raster (prepare_training_samples +
    train_random_forest_classifier, host scikit-learn as in the reference), cached as random_forest_model.joblib, and
    the map is the fitted model's prediction — computed by the forest kernel."""
    import joblib
    from modules.features import extract as E
    from rsseg import stages
    from rsseg.tiff import write_tiff
    fd, hier = stages.run_feature_extraction_stage(list(crop["bands"]))
    paths = stages.save_feature_outputs(str(tmp_path), fd, hier, 96, 96)
    roi = np.zeros((96, 96), np.uint8)
    ndvi = fd["ndvi"]
    roi[8:40, 8:40] = np.where(ndvi[8:40, 8:40] > np.median(ndvi), 1, 2)
    roi[60:90, 50:90] = 3
    write_tiff(str(tmp_path / "roi.tif"), roi)
    X, y = E.prepare_training_samples(hier["all"], str(tmp_path / "roi.tif"))
    keep = roi.reshape(-1) != 0
    assert np.array_equal(X, hier["all"].reshape(-1, 19)[keep]) and np.array_equal(y, roi.reshape(-1)[keep])
    with pytest.raises(FileNotFoundError):
        E.prepare_training_samples(hier["all"], str(tmp_path / "nope.tif"))
    with pytest.raises(ValueError):
        E.prepare_training_samples(hier["all"][:50], str(tmp_path / "roi.tif"))
    out_dir = tmp_path / "rf_train"
    got = stages.run_classification_stage(paths["pkl"], "random_forest", str(out_dir), labeled_roi_file=str(tmp_path / "roi.tif"))
    model = joblib.load(out_dir / stages.RF_MODEL_FILE)
    assert model.n_features_in_ == 19 and set(model.classes_) == {1, 2, 3}
    assert got.shape == (96, 96) and np.array_equal(got, model.predict(hier["all"].reshape(-1, 19)).reshape(96, 96))
    again = stages.run_classification_stage(paths["pkl"], "random_forest", str(out_dir))      # the cached model, no label raster needed
    assert np.array_equal(again, got)
    # the reference's own KMeans call on a stage-2 pickle hands over the empty key list and raises (scripts/3:391, extract.py:533)
    with pytest.raises(ValueError):
        stages.run_classification_stage(paths["pkl"], "kmeans", str(tmp_path / "strict"), strict_reference=True)
    # the context lent to the stage is handed back afterwards
    from rsseg import runtime as rt
    before = rt._default_ctx
    stages.run_classification_stage(paths["pkl"], "rule_based", str(tmp_path / "lend"), ctx=ctx)
    assert rt._default_ctx is before


def test_kmeans_default_key_selection_equals_the_reference_on_55_planes(ctx, golden_dir, oracle):
    """The DEFAULT call of unsupervised_kmeans_classification (feature_keys_to_use=None, extract.py:516-522): every 2-D
    plane of a stage-2-shaped dictionary, 55 float32 / float64 planes -> a float64 matrix.  The labels were produced by the
    reference function itself (oracle/gen_golden.py, tests/golden/crop96_stage2.npz)."""
    from modules.features import extract as E
    g = np.load(os.path.join(golden_dir, "crop96_stage2.npz"))
    d = {str(k): g[f"plane_{i:02d}"] for i, k in enumerate(g["keys"])}
    assert len(d) == 55 and {v.dtype for v in d.values()} == {np.dtype(np.float32), np.dtype(np.float64)}
    d["height"], d["width"] = int(g["height"]), int(g["width"])
    d["transform"], d["crs"] = None, None
    for k in (5, 8):
        got = E.unsupervised_kmeans_classification(d, k) if k == 5 else E.unsupervised_kmeans_classification(d, k, None)
        assert got.dtype == np.int32 and np.array_equal(got, g[f"kmeans_auto_k{k}"]), k
    planes = [ctx.to_device(np.ascontiguousarray(v, np.float64).reshape(-1)) for v in list(d.values())[:55]]
    labels, meta = ctx.kmeans_fit_predict(planes, 8)
    want, info = oracle.kmeans_fit_planes([np.asarray(v, np.float64) for v in list(d.values())[:55]], 8)
    assert meta["n_iter"] == info["n_iter"] and np.array_equal(meta["init_indices"], info["init_indices"])
    assert np.array_equal(labels.cpu().numpy(), want)


def test_scripts_2_3_driver_on_the_bundled_scene(ctx, golden_dir, tmp_path):
    """python -m rsseg.stages <image.tif> <outdir> --classify kmeans: BASELINE configs[0] as one command — GeoTIFF in,
    the files of scripts/2:193-258 and scripts/3:491-498 out."""
    from rsseg import stages
    from rsseg.tiff import read_tiff, write_tiff
    dn = np.load(os.path.join(golden_dir, "scene_aa.npz"))["dn"][:, 100:356, 200:456]
    tr = (30.0, 0.0, 440000.0, 0.0, -30.0, 3300000.0)
    write_tiff(str(tmp_path / "in.tif"), dn, transform=tr, epsg=32649)
    assert stages.main([str(tmp_path / "in.tif"), str(tmp_path / "out"), "--classify", "kmeans", "--n-clusters", "6"]) == 0
    fo = tmp_path / "out" / "feature_outputs"
    for name in ("level1_features.npy", "level2_features.npy", "all_hierarchical_features.npy", "all_features_and_metadata.pkl",
                 "all_hierarchical_features.tif"):
        assert (fo / name).exists(), name
    allf = np.load(fo / "all_hierarchical_features.npy")
    assert allf.shape == (256, 256, 19) and allf.dtype == np.float64
    cm = read_tiff(str(tmp_path / "out" / "segmentation_results" / "kmeans_classification_map.tif"))[0]
    assert cm.dtype == np.uint8 and cm.shape == (256, 256) and cm.min() == 1 and cm.max() == 6
    assert np.array_equal(cm, np.load(tmp_path / "out" / "segmentation_results" / "classification_kmeans.npy"))


def _pca_f64(bands, scaled_fn):
    """float64 truth of perform_pca for a given column scaling."""
    X = np.stack([np.asarray(b, np.float64).reshape(-1) for b in bands], 1)
    X = scaled_fn(X)
    mu = X.mean(0)
    w, V = np.linalg.eigh(np.cov(X.T))
    order = np.argsort(w)[::-1]
    Vt = V[:, order].T
    Vt *= np.sign(Vt[np.arange(len(Vt)), np.abs(Vt).argmax(1)])[:, None]
    return (X - mu) @ Vt.T, w[order] / w.sum()


def test_perform_pca_branches_the_reference_accepts(ctx, crop):
    """perform_pca on inputs other than robust-normalised bands (the reference accepts any bands, indices.py:205-246):
    raw DN 0-255 (the fixed-point accumulation sizes itself from the measured range), the min-max branch
    use_robust_scaling=False (indices.py:232-234), and NaN input (scikit-learn's PCA raises ValueError)."""
    from modules.features import indices as I
    dn = [np.asarray(b, np.float32) for b in crop["bands"]]          # integer-valued DN 0..255
    h, w = dn[0].shape

    def robust(X):
        med = np.median(X, 0)
        q = np.percentile(X, [25, 75], axis=0)
        s = q[1] - q[0]
        s[s == 0] = 1
        return (X - med) / s

    got, ratio, model = I.perform_pca(dn, n_components=3)
    T, r = _pca_f64(dn, robust)
    assert np.allclose(ratio, r[:3], atol=2e-6)
    for c in range(3):
        scale = np.abs(T[:, c]).max()
        assert np.abs(got[c].reshape(-1) - T[:, c]).max() <= 2e-5 * max(scale, 1.0), c
    got, ratio, _ = I.perform_pca(dn, use_robust_scaling=False)
    T, r = _pca_f64(dn, lambda X: (X - X.min(0)) / (X.max(0) - X.min(0) + 1e-10))
    assert len(got) == 7 and np.allclose(ratio, r, atol=2e-6)
    for c in range(3):
        assert np.abs(got[c].reshape(-1) - T[:, c]).max() <= 1e-5, c
    huge = [b * 1.0e4 + 3.0e6 for b in dn]                          # values far outside [0, 1]: still exact accumulation
    got, ratio2, _ = I.perform_pca(huge, n_components=2, use_robust_scaling=False)
    assert np.allclose(ratio2, r[:2], atol=2e-5)
    bad = [b.copy() for b in dn]
    bad[2][5, 7] = np.nan
    with pytest.raises(ValueError):
        I.perform_pca(bad, n_components=2)
    with pytest.raises(ValueError):
        I.perform_pca(bad, n_components=2, use_robust_scaling=False)


def test_evi_coefficients_and_unpreprocessed_stage(ctx, crop, oracle):
    """calculate_evi with non-default coefficients (indices.py:73) bit for bit against the NumPy formula, and
    run_feature_extraction_stage(preprocessing=False) (scripts/2:39-47) against the oracle's stage."""
    from modules.features import indices as I
    from rsseg import stages
    nb = [oracle.robust_normalize(b) for b in crop["bands"][:5]]
    blue, _, red, nir, _ = nb
    for (Lc, C1, C2, G) in [(1, 6, 7.5, 2.5), (0.5, 5.0, 7.0, 2.0), (1.0, 2.4, 0.0, 2.5)]:
        den = nir + C1 * red - C2 * blue + Lc
        want = np.zeros_like(nir, dtype=np.float32)
        m = den > 0.001
        want[m] = G * (nir[m] - red[m]) / den[m]
        want = np.clip(want, -1.0, 1.0)
        assert np.array_equal(I.calculate_evi(nir, red, blue, Lc, C1, C2, G), want), (Lc, C1, C2, G)
    unit = [oracle.robust_normalize(b) for b in crop["bands"]]       # any bands will do; these keep the indices meaningful
    fd, hier = stages.run_feature_extraction_stage(unit, preprocessing=False)
    ofd, ohier = oracle.run_feature_extraction_stage(unit, preprocessing=False)
    assert np.array_equal(fd["ndvi"], ofd["ndvi"]) and np.array_equal(fd["bsi"], ofd["bsi"])
    for c in list(range(6)) + list(range(7, 13)) + list(range(14, 19)):     # every column but PC0 and its context mean
        assert np.array_equal(hier["all"][:, :, c], ohier["all"][:, :, c]), c
    assert np.abs(hier["all"][:, :, 6] - ohier["all"][:, :, 6]).max() < 2e-5


def test_classification_stage_writes_the_geotiff(ctx, crop, tmp_path):
    """scripts/3:491-498: with transform / crs in the feature file the stage writes <method>_classification_map.tif
    (uint8, nodata 0, georeferenced); without them only the .npy."""
    from rsseg import stages
    from rsseg.tiff import read_tiff, read_tiff_georef
    fd, hier = stages.run_feature_extraction_stage(list(crop["bands"]))
    tr = (30.0, 0.0, 440000.0, 0.0, -30.0, 3300000.0)
    paths = stages.save_feature_outputs(str(tmp_path), fd, hier, 96, 96, transform=tr, crs="EPSG:32649")
    assert np.array_equal(np.moveaxis(read_tiff(paths["tif"]), 0, -1), hier["all"])       # LZW tiles, float64, 19 bands
    out = stages.run_classification_stage(paths["pkl"], "kmeans", str(tmp_path / "cls"), n_clusters=5, feature_keys=["hierarchical_features_all"])
    tif = tmp_path / "cls" / "kmeans_classification_map.tif"
    assert tif.exists()
    back = read_tiff(str(tif))
    assert back.dtype == np.uint8 and np.array_equal(back[0], out)
    assert read_tiff_georef(str(tif)) == {"transform": tr, "epsg": 32649, "nodata": 0.0}
    paths2 = stages.save_feature_outputs(str(tmp_path / "nogeo"), fd, hier, 96, 96)
    stages.run_classification_stage(paths2["pkl"], "kmeans", str(tmp_path / "cls2"), n_clusters=5)
    assert not (tmp_path / "cls2" / "kmeans_classification_map.tif").exists()
    assert stages.run_classification_stage(paths["pkl"], "no_such_method", str(tmp_path / "cls3")) is None


@pytest.fixture(scope="module")
def scene(golden_dir):
    return np.load(os.path.join(golden_dir, "scene_aa.npz"))


def test_rule_based_classification_vs_oracle(ctx, scene, oracle, tmp_path):
    """thresholds + elliptical close / open + 8-connected area filter (scipy.ndimage.label in the
    oracle, the reference's own dependency) + priority merge + bare land, on the bundled 600 x 600 scene and on random
    masks: bit for bit.  Then the stage driver with method='rule_based'."""
    from modules.features import extract as E
    from rsseg import _lib as L
    from rsseg import stages
    rng = np.random.default_rng(12)
    # the component filter and the elliptical morphology on their own
    for H, W, p in ((97, 131, 0.45), (64, 300, 0.6), (33, 33, 0.3)):
        m = (rng.random((H, W)) < p).astype(np.uint8)
        m[10:25, 5:20] = 1
        d = ctx.to_device(m.reshape(-1))
        for k in (3, 5):
            for name, op in (("erosion", L.MORPH_ERODE), ("dilation", L.MORPH_DILATE), ("opening", L.MORPH_OPEN), ("closing", L.MORPH_CLOSE)):
                assert np.array_equal(ctx.morph_ellipse(d, H, W, k, op).cpu().numpy().reshape(H, W), oracle.morph_ellipse(m, k, name)), (k, name)
        for min_area in (1, 2, 9, 40, 100000):
            from scipy import ndimage
            lab, _ = ndimage.label(m, structure=np.ones((3, 3)))
            area = np.bincount(lab.ravel())
            want = m.copy()
            want[np.isin(lab, np.where((area < min_area) & (area > 0))[0])] = 0
            got = ctx.remove_small_components(d, H, W, min_area).cpu().numpy().reshape(H, W)
            assert np.array_equal(got, want), (H, W, min_area)
        assert np.array_equal(E.advanced_post_processing(m, 7, 5), oracle.advanced_post_processing(m, 7, 5))
    # spirals / long thin components: many union steps across rows
    sp = np.zeros((80, 80), np.uint8)
    for r in range(0, 40, 2):
        sp[r, r:80 - r] = 1
        sp[r:80 - r, 79 - r] = 1
        sp[79 - r, r:80 - r] = 1
        sp[r + 2:80 - r, r] = 1
    from scipy import ndimage
    lab, nf = ndimage.label(sp, structure=np.ones((3, 3)))
    big = int(np.bincount(lab.ravel())[1:].max())
    got = ctx.remove_small_components(ctx.to_device(sp.reshape(-1)), 80, 80, big).cpu().numpy().reshape(80, 80)
    area = np.bincount(lab.ravel())
    want = sp.copy()
    want[np.isin(lab, np.where((area < big) & (area > 0))[0])] = 0
    assert np.array_equal(got, want)
    # the whole rule set on the scene
    bands = oracle.stage1_preprocess(scene["dn"])
    norm = [oracle.robust_normalize(b) for b in bands]
    b, g, r, n, s = norm[:5]
    feats = dict(ndvi=oracle.calculate_ndvi(n, r), ndwi=oracle.calculate_ndwi(g, n), mndwi=oracle.calculate_mndwi(g, s),
                 ndbi=oracle.calculate_ndbi(s, n), height=600, width=600)
    want = oracle.rule_based_classification(feats)
    got = E.rule_based_classification(feats)
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    assert set(np.unique(want)) >= {0, 1} and (want > 0).mean() > 0.2        # a non-trivial map
    no_mndwi = {k: v for k, v in feats.items() if k != "mndwi"}
    assert np.array_equal(E.rule_based_classification(no_mndwi), oracle.rule_based_classification(no_mndwi))
    # NaN pixels: threshold_segmentation counts them as 0 (extract.py:354-356), the bare-land band tests do not
    # (extract.py:486-497: a NaN fails both comparisons, although 0 lies inside (-0.1, 0.2) and (-0.2, 0.2))
    holes = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in feats.items()}
    for key in ("ndvi", "ndbi", "mndwi"):
        holes[key][300:340, 100:180] = np.nan
    holes["ndvi"][::37, ::11] = np.nan
    want_h = oracle.rule_based_classification(holes)
    assert np.array_equal(E.rule_based_classification(holes), want_h)
    assert not (want_h[300:340, 100:180] == 4).any()
    zeros = {k: (np.nan_to_num(v, nan=0.0) if isinstance(v, np.ndarray) else v) for k, v in holes.items()}
    assert (oracle.rule_based_classification(zeros)[300:340, 100:180] == 4).any()      # with 0 instead of NaN the block IS bare land
    bl = E.extract_bareland_by_rule(holes, None, None, None)
    assert not bl[300:340, 100:180].any()
    for fn in ("extract_vegetation_by_threshold", "extract_water_by_threshold", "extract_builtup_by_threshold"):
        m = getattr(E, fn)(feats)
        assert m.shape == (600, 600) and m.dtype == np.uint8
    assert np.array_equal(E.threshold_segmentation(feats["ndvi"], 0.2), (feats["ndvi"] > 0.2).astype(np.uint8))
    assert np.array_equal(E.threshold_segmentation(feats["ndvi"], 0.2, above=False), (feats["ndvi"] < 0.2).astype(np.uint8))
    # stage driver
    fd, hier = stages.run_feature_extraction_stage(bands)
    paths = stages.save_feature_outputs(str(tmp_path), fd, hier, 600, 600)
    out = stages.run_classification_stage(paths["pkl"], "rule_based", str(tmp_path / "cls"))
    assert np.array_equal(out, want)


def test_rule_based_branch_through_star_import_names(ctx, scene, oracle):
    """The rule-based branch as scripts/3_classification.py:338-375 writes it, with every name taken from
    `from modules.features.extract import *` (scripts/3:25): the three extractors with the script's thresholds and minimum
    areas, the priority merge, extract_bareland_by_rule on the merged map's classes — equal to the oracle's restatement of
    the same branch and to the fused device form rule_based_classification."""
    ns = {}
    exec("from modules.features.extract import *", ns)                    # noqa: S102
    bands = oracle.stage1_preprocess(scene["dn"])
    b, g, r, n, s = [oracle.robust_normalize(x) for x in bands[:5]]
    features = dict(ndvi=oracle.calculate_ndvi(n, r), ndwi=oracle.calculate_ndwi(g, n), mndwi=oracle.calculate_mndwi(g, s),
                    ndbi=oracle.calculate_ndbi(s, n), height=600, width=600)
    img_shape = (600, 600)
    px = img_shape[0] * img_shape[1]
    np_ = ns["np"]
    veg = ns["extract_vegetation_by_threshold"](features, ndvi_threshold=0.25, min_area=int(px * 0.0005))
    water = ns["extract_water_by_threshold"](features, ndwi_threshold=0.05, min_area=int(px * 0.0002))
    built = ns["extract_builtup_by_threshold"](features, ndbi_threshold=0.0, ndvi_threshold_for_builtup=0.2, min_area=int(px * 0.001))
    final = np_.zeros(img_shape, dtype=np_.uint8)
    for mask, cid in ((built, 3), (veg, 1), (water, 2)):
        assert mask.shape == img_shape and mask.dtype == np.uint8
        final[mask == 1] = cid
    bare = ns["extract_bareland_by_rule"](features, vegetation_mask=(final == 1), water_mask=(final == 2), builtup_mask=(final == 3),
                                          min_area=int(px * 0.0005))
    assert bare.shape == img_shape and bare.dtype == np.uint8
    final[(bare == 1) & (final == 0)] = 4
    want = oracle.rule_based_classification(features)
    assert np.array_equal(final, want)
    assert np.array_equal(ns["rule_based_classification"](features), want)
    assert set(np.unique(final)) >= {0, 1, 2, 3}
    # a scene with bare land (the bundled one has none after the area filter): lower the built-up evidence in a block
    f2 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in features.items()}
    f2["ndvi"][200:300, 200:330] = 0.05
    f2["ndbi"][200:300, 200:330] = -0.1
    f2["mndwi"][200:300, 200:330] = -0.5
    want2 = oracle.rule_based_classification(f2)
    assert (want2 == 4).sum() > 5000 and np.array_equal(ns["rule_based_classification"](f2), want2)
    bare2 = ns["extract_bareland_by_rule"](f2, vegetation_mask=(want2 == 1), water_mask=(want2 == 2), builtup_mask=(want2 == 3), min_area=int(px * 0.0005))
    assert np.array_equal(bare2 == 1, want2 == 4)


def test_otsu_hole_fill_and_large_ellipses_vs_oracle(ctx, scene, oracle):
    """threshold_segmentation(otsu=True) (extract.py:358-371), advanced_post_processing with even / zero kernel sizes
    (binary_fill_holes, :314-316) and elliptical elements beyond 5 x 5: bit for bit against the oracle (scipy for the
    hole fill), float32 and float64 planes, NaN pixels, planes without contrast."""
    from modules.features import extract as E
    from rsseg import _lib as L
    from scipy import ndimage
    rng = np.random.default_rng(21)
    bands = oracle.stage1_preprocess(scene["dn"])
    b, g, r, n, s = [oracle.robust_normalize(x) for x in bands[:5]]
    ndvi = oracle.calculate_ndvi(n, r)
    planes = [ndvi, ndvi.astype(np.float64) * 1.000001, rng.normal(0, 1, (257, 131)).astype(np.float32),
              np.where(rng.random((90, 70)) < 0.5, rng.normal(3, 0.2, (90, 70)), rng.normal(9, 0.5, (90, 70))),
              (rng.random((50, 50)) * 1e-12).astype(np.float32), rng.integers(0, 7, (40, 40)).astype(np.float32)]
    withnan = ndvi.copy()
    withnan[::7, ::5] = np.nan
    planes.append(withnan)
    for i, a in enumerate(planes):
        for above in (True, False):
            want = oracle.threshold_segmentation(a, None, above=above, otsu=True)
            got = E.threshold_segmentation(a, 123.0, above=above, otsu=True)
            assert got.dtype == np.uint8 and np.array_equal(got, want), (i, above, int((got != want).sum()))
    d = ctx.to_device(np.ascontiguousarray(ndvi).reshape(-1))
    _, level, mn, mx = ctx.otsu_mask(d)
    stretched = np.clip((ndvi - ndvi.min()) / (ndvi.max() - ndvi.min() + 1e-10) * 255, 0, 255).astype(np.uint8)
    assert level == oracle.otsu_level_u8(stretched) and mn == float(ndvi.min()) and mx == float(ndvi.max())
    for a in (np.full((9, 11), 0.25, np.float32), np.full((3, 3), np.nan, np.float32), np.zeros((4, 4))):
        assert not E.threshold_segmentation(a, 0, otsu=True).any() and E.threshold_segmentation(a, 0, above=False, otsu=True).all()
    # float64 planes compare in float64 (a value one ulp above a threshold that float32 would round onto it)
    t = 0.2
    x64 = np.array([[t, np.nextafter(t, 1), np.nextafter(t, 0), np.nan]])
    assert E.threshold_segmentation(x64, t).tolist() == (np.nan_to_num(x64) > t).astype(np.uint8).tolist() == [[0, 1, 0, 0]]
    assert E.threshold_segmentation(x64, t, above=False).tolist() == [[0, 0, 1, 1]]
    # hole fill
    for H, W, p in ((64, 64, 0.55), (97, 131, 0.7), (33, 200, 0.62), (1, 9, 0.5), (7, 1, 0.5), (2, 2, 0.5)):
        m = (rng.random((H, W)) < p).astype(np.uint8)
        got = ctx.fill_holes(ctx.to_device(m.reshape(-1)), H, W).cpu().numpy().reshape(H, W)
        assert np.array_equal(got, ndimage.binary_fill_holes(m).astype(np.uint8)), (H, W)
    ring = np.zeros((300, 300), np.uint8)
    yy, xx = np.mgrid[:300, :300]
    rr = np.hypot(yy - 150, xx - 150)
    ring[(rr < 140) & (rr > 100)] = 1
    ring[(rr < 60) & (rr > 30)] = 1
    ring[150, 0:60] = 0                        # a cut through the outer ring: what lies between the rings is no longer a hole
    got = ctx.fill_holes(ctx.to_device(ring.reshape(-1)), 300, 300).cpu().numpy().reshape(300, 300)
    assert np.array_equal(got, ndimage.binary_fill_holes(ring).astype(np.uint8)) and got[150, 150] == 1
    m = (rng.random((120, 140)) < 0.6).astype(np.uint8)
    for k in (0, 2, 4, 6):
        for fill in (True, False):
            for min_area in (0, 12):
                want = oracle.advanced_post_processing(m, min_area, k, fill)
                got = E.advanced_post_processing(m, min_area, k, fill)
                assert np.array_equal(got, want), (k, fill, min_area)
    # elliptical elements beyond 5 x 5
    d = ctx.to_device(m.reshape(-1))
    for k in (7, 9, 15, 31):
        for name, op in (("erosion", L.MORPH_ERODE), ("dilation", L.MORPH_DILATE), ("opening", L.MORPH_OPEN), ("closing", L.MORPH_CLOSE)):
            assert np.array_equal(ctx.morph_ellipse(d, 120, 140, k, op).cpu().numpy().reshape(120, 140), oracle.morph_ellipse(m, k, name)), (k, name)
    assert np.array_equal(E.advanced_post_processing(m, 30, 7), oracle.advanced_post_processing(m, 30, 7))
    assert np.array_equal(E.advanced_post_processing(m, 30, 1), oracle.advanced_post_processing(m, 30, 1))
    with pytest.raises(Exception):
        ctx.morph_ellipse(d, 120, 140, 33, L.MORPH_ERODE)


def test_lbp_entropy_gaussian_members_vs_oracle(ctx, crop, oracle):
    """SURVEY.md 8f N3, the last members of the stage-2 dictionary: uniform LBP (bit-exact codes), rank entropy over
    disk(1 / 3 / 5) (float64, 1e-12: the device log is not glibc's), OpenCV's fixed-point Gaussian 5 / 15 (bit-exact
    uint8) and the DoG — against the NumPy restatements, on an odd-sized plane and through the mirror functions."""
    from modules.features import indices as I
    from rsseg import _lib as L
    import ctypes as C
    rng = np.random.default_rng(33)
    H, W = 53, 71
    q = rng.integers(0, 256, (H, W)).astype(np.uint8)
    q[10:30, 20:50] = (np.add.outer(np.arange(20), np.arange(30)) * 3 % 256).astype(np.uint8)   # smooth ramp: many exact ties
    q[35:45, 5:25] = 200
    d = ctx.to_device(q.reshape(-1))
    for P, R in ((24, 3), (8, 1), (16, 2)):
        got = ctx.lbp_uniform(d, H, W, P, R).cpu().numpy().reshape(H, W)
        assert np.array_equal(got.astype(np.float64), oracle.lbp_uniform(q, P, R)), (P, R)
    for k in (1, 3, 5):
        got = ctx.rank_entropy(d, H, W, k).cpu().numpy().reshape(H, W)
        assert np.allclose(got, oracle.rank_entropy(q, k), rtol=0, atol=1e-12), k
    for ks in (5, 15, 3, 7, 9):
        taps = (C.c_int * ks)()
        assert L.load().rsseg_host_gaussian_kernel_fixed(ks, taps) == 0
        assert list(taps) == list(oracle.gaussian_taps_fixed(ks)) and sum(taps) == 256, ks
        assert np.array_equal(ctx.gaussian_blur_u8(d, H, W, ks).cpu().numpy().reshape(H, W), oracle.gaussian_blur_u8(q, ks)), ks
    assert list(oracle.gaussian_taps_fixed(5)) == [16, 64, 96, 64, 16]
    band = crop["bands"][3]
    f = I.calculate_filter_responses(band)
    want = oracle.filter_responses_extra(band)
    for k in ("gaussian_5", "gaussian_15", "dog"):
        assert f[k].dtype == np.float64 and np.array_equal(f[k], want[k]), k
    assert set(f) == {"gaussian_5", "gaussian_15", "dog", "laplacian", "sobel_mag"}
    u8 = oracle.to_u8(oracle.robust_normalize(band))
    lbp = I.calculate_lbp_features(band)
    o = oracle.lbp_uniform(u8, 24, 3)
    assert lbp.dtype == np.float64 and np.array_equal(lbp, o / o.max())
    ms = I.calculate_multi_scale_features(band)
    assert {"entropy_scale_1", "entropy_scale_3", "entropy_scale_5"} <= set(ms) and "entropy_scale_7" not in ms
    e3 = oracle.rank_entropy(u8, 3)
    assert np.allclose(ms["entropy_scale_3"], e3 / e3.max(), rtol=0, atol=1e-12)
