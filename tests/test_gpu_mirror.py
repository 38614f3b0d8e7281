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
    with pytest.raises(ValueError):
        I.calculate_evi(n, r, b, L=2)


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
    assert {f"{m}_scale_{k}" for m in ("mean", "variance", "std_dev") for k in (1, 3, 5, 7)} == set(ms)
    assert np.array_equal(ms["std_dev_scale_5"], oracle.std_dev_feature(nir, 5))
    assert np.array_equal(ms["variance_scale_3"], oracle.variance_feature(nir, 3))
    assert not ms["variance_scale_1"].any() and ms["variance_scale_1"].dtype == np.float32
    fr = I.calculate_filter_responses(nir)
    assert np.array_equal(fr["sobel_mag"], oracle.sobel_mag_feature(nir))
    assert np.array_equal(fr["laplacian"], oracle.laplacian_feature(nir)) and fr["laplacian"].dtype == np.float32


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
    assert len(fd["morphological_features"]) == 15 and len(fd["multi_scale_features"]) == 12
    assert np.array_equal(fd["morphological_features"]["gradient_5"], hier["all"][:, :, 16])
    assert np.array_equal(fd["multi_scale_features"]["std_dev_scale_5"].astype(np.float64), hier["all"][:, :, 17])
    assert np.array_equal(fd["morphological_features"]["closing_7"], oracle.calculate_morphological_features(crop["norm"][3])["closing_7"])
    assert set(fd["filter_features"]) == {"laplacian", "sobel_mag"}
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


def test_classification_stage_driver(ctx, crop, tmp_path):
    """stage 2 files -> load_features -> normalize_features_structure -> KMeans on 'hierarchical_features_all'."""
    from rsseg import stages
    fd, hier = stages.run_feature_extraction_stage(list(crop["bands"]))
    paths = stages.save_feature_outputs(str(tmp_path), fd, hier, 96, 96)
    out = stages.run_classification_stage(paths["pkl"], "kmeans", str(tmp_path / "cls"), n_clusters=6)
    assert out.shape == (96, 96) and out.dtype == np.uint8 and out.min() == 1 and out.max() == 6
    assert np.array_equal(np.load(tmp_path / "cls" / "classification_kmeans.npy"), out)
    direct = stages.run_kmeans_stage(hier["all"], 6)
    assert np.array_equal(out, direct)
