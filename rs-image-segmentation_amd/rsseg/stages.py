"""
Counterparts of the two stage functions that bracket the hot path:

  run_feature_extraction_stage   reference scripts/2_feature_extraction.py:27-133
  run_classification_stage       reference scripts/3_classification.py:267-505 (KMeans and forest branches)

They keep the reference's return values and on-disk layout (.npy of the (H, W, C) float64 stacks,
scripts/2:193-214; pickle with 'hierarchical_features' / 'all_extracted_features_dict' / 'dimensions',
scripts/2:222-232; label maps as .npy), but run the whole chain on the device with ONE upload of the
bands and one download per product, instead of a host round trip per function as the per-function
mirrors in modules/ do.  Plotting (scripts/2:131, 267-385; scripts/3:491) is out of scope.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import pipeline as P
from .runtime import Context, default_context


def run_feature_extraction_stage(bands_data: Sequence[np.ndarray], preprocessing: bool = True, texture_band_index: int = 3,
                                 ctx: Optional[Context] = None) -> Tuple[Dict, Dict]:
    """bands_data: list of >= 5 (H, W) arrays in TM band order.  `texture_band_index` is accepted and
    ignored exactly as in the reference (NIR = bands[3] is hard-wired, scripts/2:84)."""
    ctx = ctx or default_context()
    h, w = np.asarray(bands_data[0]).shape
    # 8-bit rasters cross PCIe as one byte per pixel and STAY one byte per pixel in HBM (Context.upload_band): the order
    # statistics, index and PCA kernels read the uint8 planes directly; anything else is float32 as in the reference
    arrs = [np.asarray(b) for b in bands_data if b is not None]
    all_u8 = all(a.dtype == np.uint8 for a in arrs)
    dev = [ctx.upload_band(a) if all_u8 else ctx.upload_f32(a) for a in arrs]
    planes, ex = P.feature_stack19(ctx, dev, h, w, preprocessing=bool(preprocessing))   # False: bands taken as given (scripts/2:43-50)

    def host(t):
        return t.cpu().numpy().reshape(h, w)

    # features_dict in the reference's insertion order (scripts/2:62-106; the members of the nested dicts in the order
    # indices.py builds them): the order is part of the interface, because unsupervised_kmeans_classification's default
    # key selection stacks the 2-D members in dict order (extract.py:516-522, 568)
    from . import _lib as L
    fd: Dict[str, object] = {k: host(v) for k, v in ex["indices"].items()}
    fd["pca_result"] = [host(p) for p in ex["pca"]]
    fd["variance_ratio"] = ex["pca_ratio"]
    fd["glcm_features"] = {k: host(v) for k, v in ex["glcm"].items()}
    nir2, q255 = ex["nir2"], ex["q255"]  # the re-normalised NIR band and its uint8 image (indices.py:412-415)
    lbp = ctx.lbp_uniform(q255, h, w, 24, 3).cpu().numpy().reshape(h, w).astype(np.float64)
    fd["lbp_feature"] = lbp / lbp.max()
    ms: Dict[str, np.ndarray] = {}
    for k in (1, 3, 5, 7):                # per scale: mean, variance, std_dev, entropy for scales <= 5 (indices.py:535-560)
        if k == 1:                        # cv2.blur with a 1 x 1 kernel is the identity: variance and std are exactly 0
            ms["mean_scale_1"] = host(nir2)
            ms["variance_scale_1"] = np.zeros((h, w), np.float32)
            ms["std_dev_scale_1"] = np.zeros((h, w), np.float32)
        else:
            ms[f"mean_scale_{k}"] = host(ctx.box_mean(nir2, h, w, k, L.BORDER_REFLECT101))
            ms[f"variance_scale_{k}"] = host(ctx.local_var(nir2, h, w, k))
            ms[f"std_dev_scale_{k}"] = host(planes[17]) if k == 5 else host(ctx.local_std(nir2, h, w, k))
        if k <= 5:
            e = ctx.rank_entropy(q255, h, w, k).cpu().numpy().reshape(h, w)
            ms[f"entropy_scale_{k}"] = e / np.max(e)
    fd["multi_scale_features"] = ms
    ops = (("erosion", L.MORPH_ERODE), ("dilation", L.MORPH_DILATE), ("opening", L.MORPH_OPEN), ("closing", L.MORPH_CLOSE),
           ("gradient", L.MORPH_GRADIENT))
    fd["morphological_features"] = {f"{name}_{k}": ctx.morph(q255, h, w, k, op).cpu().numpy().reshape(h, w) / 255.0
                                    for k in (3, 5, 7) for name, op in ops}
    g5 = ctx.gaussian_blur_u8(q255, h, w, 5).cpu().numpy().reshape(h, w) / 255.0
    g15 = ctx.gaussian_blur_u8(q255, h, w, 15).cpu().numpy().reshape(h, w) / 255.0
    dog = g5 - g15
    fd["filter_features"] = {"gaussian_5": g5, "gaussian_15": g15, "dog": (dog - dog.min()) / (dog.max() - dog.min() + 1e-10),
                             "laplacian": host(ctx.laplacian_norm(q255, h, w)), "sobel_mag": host(planes[18])}
    stack = P.stack19_to_host(planes, h, w)
    hier = {"level_1": np.ascontiguousarray(stack[:, :, :14]), "level_2": np.ascontiguousarray(stack[:, :, 14:]), "all": stack}
    return fd, hier


def save_feature_outputs(output_dir: str, features_dict: Dict, hierarchical_features: Dict, height: int, width: int,
                         transform=None, crs=None) -> Dict[str, str]:
    """File names and contents of scripts/2:193-258: the three .npy stacks, the pickle, and the full stack as
    all_hierarchical_features.tif (one band per feature, float64, georeferenced with `transform` / `crs`, LZW in
    256 x 256 tiles as the reference asks rasterio for; BigTIFF once the stack passes 4 GB).
    `crs`: an EPSG integer, 'EPSG:xxxx', or an object with to_epsg()."""
    os.makedirs(output_dir, exist_ok=True)
    paths = {"level1": os.path.join(output_dir, "level1_features.npy"),
             "level2": os.path.join(output_dir, "level2_features.npy"),
             "all": os.path.join(output_dir, "all_hierarchical_features.npy"),
             "pkl": os.path.join(output_dir, "all_features_and_metadata.pkl"),
             "tif": os.path.join(output_dir, "all_hierarchical_features.tif")}
    np.save(paths["level1"], hierarchical_features["level_1"])
    np.save(paths["level2"], hierarchical_features["level_2"])
    np.save(paths["all"], hierarchical_features["all"])
    with open(paths["pkl"], "wb") as f:
        pickle.dump({"hierarchical_features": hierarchical_features, "all_extracted_features_dict": features_dict,
                     "dimensions": (height, width), "geo_transform": transform, "crs": crs}, f)
    from .tiff import write_tiff
    write_tiff(paths["tif"], np.moveaxis(hierarchical_features["all"], -1, 0), transform=transform, epsg=_epsg_of(crs),
               geographic=_is_geographic(crs), compress="lzw", tiled=True)
    return paths


def _epsg_of(crs) -> Optional[int]:
    if crs is None:
        return None
    if hasattr(crs, "to_epsg"):
        return crs.to_epsg()
    if isinstance(crs, str) and crs.upper().startswith("EPSG:"):
        return int(crs.split(":")[1])
    return int(crs)


def _is_geographic(crs) -> Optional[bool]:
    g = getattr(crs, "is_geographic", None)   # rasterio.crs.CRS
    return bool(g) if g is not None else None


def save_class_map_tif(class_map: np.ndarray, out_tif: str, transform=None, crs=None) -> str:
    """Convenience form of modules.features.extract.save_classification_as_geotiff (reference extract.py:778-833) for callers
    that hold transform / crs instead of a feature dictionary."""
    from modules.features.extract import save_classification_as_geotiff
    a = np.asarray(class_map)
    save_classification_as_geotiff(a, {"transform": transform, "crs": crs, "height": a.shape[0], "width": a.shape[1]}, out_tif)
    return out_tif


def run_kmeans_stage(hierarchical_all: np.ndarray, n_clusters: int = 7, ctx: Optional[Context] = None) -> np.ndarray:
    """KMeans branch of run_classification_stage on the 19-feature stack: labels + 1 as uint8
    (scripts/3:390-394, 509-538: final_map = labels + 1, saved as uint8 with nodata 0)."""
    ctx = ctx or default_context()
    h, w, c = hierarchical_all.shape
    dt = np.float32 if hierarchical_all.dtype == np.float32 else np.float64
    dev = [ctx.to_device(np.ascontiguousarray(hierarchical_all[:, :, i], dtype=dt).reshape(-1)) for i in range(c)]
    labels, _ = ctx.kmeans_fit_predict(dev, n_clusters)
    return (labels.cpu().numpy().reshape(h, w) + 1).astype(np.uint8)


RF_MODEL_FILE = "random_forest_model.joblib"   # scripts/3_classification.py:459


def run_classification_stage(feature_file_path, method='rule_based', output_dir="segmentation_outputs", use_hierarchical_all=True, *,
                             n_clusters: int = 7, classifier=None, feature_keys=None, labeled_roi_file: str = "labeled_roi.tif",
                             strict_reference: bool = False, ctx: Optional[Context] = None) -> Optional[np.ndarray]:
    """run_classification_stage (scripts/3_classification.py:267-505): the reference's name, positional order and defaults
    (`method='rule_based'`, `output_dir="segmentation_outputs"`, `use_hierarchical_all=True`); what follows the `*` are
    keyword-only additions.  Loads and normalises the feature file (extract.py:32-295), dispatches on `method`:

      'rule_based'     thresholds + morphology + area filter (scripts/3:335-375).  The rules read 'ndvi', 'ndwi', 'mndwi',
                       'ndbi' (extract.py:406-505); stage 2's pickle stores them under 'all_extracted_features_dict_<index>'
                       after normalisation, and those are used when the plain keys are absent.
      'kmeans'         scripts/3:377-398: the keys ['ndvi', 'ndwi', 'ndbi', 'texture_mean', 'hierarchical_all'] that exist, 7
                       clusters (`n_clusters`), labels + 1.  On a stage-2 pickle none of them exists (the keys are prefixed,
                       SURVEY.md 3.2); the reference then announces "将使用自动选择" (automatic selection) but hands over the
                       EMPTY list, on which unsupervised_kmeans_classification raises (extract.py:533).  This driver does what
                       the message says: `feature_keys_to_use=None`, i.e. every 2-D plane of the dictionary (55 on a stage-2
                       pickle).  `feature_keys` overrides the selection; `strict_reference=True` hands over the empty list
                       as the reference does, i.e. raises its ValueError.
      'random_forest'  scripts/3:401-488, inference part: the feature array is 'hierarchical_all' (also under stage 2's key
                       'hierarchical_features_all') when `use_hierarchical_all`, else every 2-D plane of the image's shape
                       stacked (:425-437); the classifier is `classifier` when given, else <output_dir>/random_forest_model.joblib
                       when it exists and its n_features_in_ matches (:459-475); otherwise the forest is fitted on the host from
                       `labeled_roi_file` (prepare_training_samples + train_random_forest_classifier, :450-475, scikit-learn as in
                       the reference) and cached as that joblib file.  Without a model and without the label raster the stage
                       reports that and returns None (the reference insists on the label raster even when the cache exists, :405-409).

    Writes <output_dir>/classification_<method>.npy and, when the feature file carries transform / crs / width / height
    (scripts/3:495-498), <output_dir>/<method>_classification_map.tif (uint8 labels, nodata 0, LZW tiles); the PNG of
    scripts/3:491 is plotting (out of scope).  The reference returns None; this returns the label map as well (None on the
    reference's error paths, which print and return)."""
    from modules.features.extract import load_features, normalize_features_structure, unsupervised_kmeans_classification
    os.makedirs(output_dir, exist_ok=True)
    try:
        raw = load_features(feature_file_path)
        if not raw:
            print("加载原始特征失败。")
            return None
        feats = normalize_features_structure(raw)
        skip = ("transform", "crs", "width", "height", "dimensions", "geo_transform", "hierarchical_level_1", "hierarchical_level_2",
                "hierarchical_all")
        has_arrays = any(isinstance(v, np.ndarray) and v.ndim >= 2 for k, v in feats.items() if k not in skip)
        has_all = isinstance(feats.get("hierarchical_all"), np.ndarray) and feats["hierarchical_all"].ndim == 3
        has_dims = isinstance(feats.get("height"), int) and isinstance(feats.get("width"), int)
        if not (has_arrays or has_all) or not has_dims:
            print("错误：规范化后的特征不包含有效的图像数组数据（单个特征或hierarchical_all）或尺寸信息。")
            print(f"规范化后的键: {list(feats.keys())}")
            return None
    except Exception as e:  # noqa: BLE001 — scripts/3:307-311 prints and returns
        print(f"加载或规范化特征失败: {e}")
        return None
    from . import runtime as _rt
    prev_ctx = _rt._default_ctx
    if ctx is not None:   # the mirrors run on the process-wide context: lend them this one for the duration of the call
        _rt._default_ctx = ctx
    try:
        out = _classify(feats, method, output_dir, use_hierarchical_all, n_clusters, classifier, feature_keys, labeled_roi_file,
                        strict_reference)
    finally:
        _rt._default_ctx = prev_ctx
    if out is None:
        return None
    from modules.features.extract import save_classification_as_geotiff
    np.save(os.path.join(output_dir, f"classification_{method}.npy"), out)
    if all(feats.get(k) is not None for k in ("transform", "crs", "width", "height")):   # scripts/3:495-498
        save_classification_as_geotiff(out, feats, os.path.join(output_dir, f"{method}_classification_map.tif"))
    else:
        print("警告: 元数据不完整，无法将分类结果保存为带地理参考的GeoTIFF。")
    return out


def _classify(feats, method, output_dir, use_hierarchical_all, n_clusters, classifier, feature_keys, labeled_roi_file, strict_reference):
    """The dispatch of scripts/3_classification.py:335-488 on a normalised feature dictionary -> label map or None."""
    from modules.features.extract import (prepare_training_samples, rule_based_classification, supervised_classification_predict,
                                          train_random_forest_classifier, unsupervised_kmeans_classification)
    shape = (feats["height"], feats["width"])
    if method == "rule_based":
        rules = {}
        for k in ("ndvi", "ndwi", "mndwi", "ndbi"):
            v = feats.get(k)
            if v is None:
                v = feats.get(f"all_extracted_features_dict_{k}")
            if v is not None:
                rules[k] = v
        rules["height"], rules["width"] = shape
        return rule_based_classification(rules)
    if method == "kmeans":
        wanted = ["ndvi", "ndwi", "ndbi", "texture_mean", "hierarchical_all"] if feature_keys is None else list(feature_keys)
        valid = [k for k in wanted if isinstance(feats.get(k), np.ndarray) and feats[k].ndim in (2, 3)]
        if not valid:
            print(f"警告: 为KMeans指定的特征键 {wanted} 在数据中均不可用，将使用自动选择。")
        keys = valid if (valid or strict_reference) else None      # the reference passes [] and raises (scripts/3:391, extract.py:533)
        return (unsupervised_kmeans_classification(feats, n_clusters, keys) + 1).astype(np.uint8)   # scripts/3:394
    if method in ("random_forest", "rf", "supervised"):
        arr, names = None, []
        if use_hierarchical_all:
            for key in ("hierarchical_all", "hierarchical_features_all"):
                v = feats.get(key)
                if isinstance(v, np.ndarray) and v.ndim == 3 and v.shape[:2] == shape:
                    arr = v
                    names = [f"hierarchical_feature_{i + 1}" for i in range(v.shape[-1])]
                    break
        if arr is None:   # scripts/3:425-437: every 2-D plane of the image's shape, in dict order
            names = [k for k, v in feats.items() if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == shape]
            if not names:
                print("错误: 为随机森林指定的特征键在数据中均不可用或不是匹配图像形状的2D数组。")
                return None
            arr = np.stack([feats[k] for k in names], axis=-1)
        model_path = os.path.join(output_dir, RF_MODEL_FILE)
        try:
            import joblib
            if classifier is None and os.path.exists(model_path):
                print(f"加载已训练的随机森林模型: {model_path}")
                classifier = joblib.load(model_path)
            nf = getattr(classifier, "n_features_in_", None)
            if classifier is not None and nf is not None and nf != arr.shape[-1]:
                print(f"警告: 加载的分类器需要 {nf} 个特征，但准备的数据有 {arr.shape[-1]} 个。正在重新训练模型。")
                classifier = None
            if classifier is None:      # scripts/3:450-475: fit on the host from the label raster, cache the model
                if not os.path.exists(labeled_roi_file):
                    print(f"错误: 监督分类所需的标签ROI文件 '{labeled_roi_file}' 未找到 (也没有可用的模型 '{model_path}')。")
                    return None
                X, y = prepare_training_samples(arr, labeled_roi_file)
                classifier = train_random_forest_classifier(X, y, feature_names_for_training=names)
                joblib.dump(classifier, model_path)
                print(f"随机森林模型训练并保存至: {model_path}")
            return supervised_classification_predict(arr, classifier)
        except _rt_unsupported():
            raise
        except Exception as e:  # noqa: BLE001 — scripts/3:482-486 prints and returns
            print(f"随机森林分类过程中发生错误: {e}")
            return None
    print(f"错误: 不支持的分割方法 '{method}'")
    return None


def _rt_unsupported():
    from .runtime import RssegUnsupported
    return RssegUnsupported


# --------------------------------------------------------------------------------------------------
# one-command driver shaped like the two scripts' __main__ blocks (scripts/2_feature_extraction.py:137-262,
# scripts/3_classification.py:545-632):   python -m rsseg.stages <image.tif> <output_dir> [--classify kmeans]
# --------------------------------------------------------------------------------------------------
def run_scripts_2_3(image_path: str, output_dir: str, classify: Optional[str] = None, preprocessing: bool = True, n_clusters: int = 7,
                    ctx: Optional[Context] = None) -> Dict[str, object]:
    """Reads the GeoTIFF's bands as float32 with nodata -> NaN (scripts/2:154-161), runs the feature stage, writes
    <output_dir>/feature_outputs/{level1,level2,all_hierarchical}_features.npy, all_features_and_metadata.pkl and
    all_hierarchical_features.tif (scripts/2:193-258), then — `classify` in {'kmeans', 'rule_based', 'random_forest'} —
    the classification stage on that pickle into <output_dir>/segmentation_results (scripts/3:548-551)."""
    from .tiff import read_tiff, read_tiff_georef
    arr = read_tiff(image_path)
    geo = read_tiff_georef(image_path)
    bands = []
    for i in range(arr.shape[0]):
        b = arr[i] if arr.dtype == np.uint8 and geo["nodata"] is None else arr[i].astype(np.float32)
        if geo["nodata"] is not None:
            b[b == geo["nodata"]] = np.nan
        bands.append(b)
    h, w = arr.shape[1:]
    fd, hier = run_feature_extraction_stage(bands, preprocessing=preprocessing, ctx=ctx)
    crs = None if geo["epsg"] is None else f"EPSG:{geo['epsg']}"
    fdir = os.path.join(output_dir, "feature_outputs")
    paths = save_feature_outputs(fdir, fd, hier, h, w, geo["transform"], crs)
    res: Dict[str, object] = {"paths": paths, "shape": (h, w)}
    if classify:
        sdir = os.path.join(output_dir, "segmentation_results")
        res["class_map"] = run_classification_stage(paths["pkl"], classify, sdir, True, n_clusters=n_clusters, ctx=ctx)
        res["segmentation_dir"] = sdir
    return res


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(prog="python -m rsseg.stages",
                                 description="feature extraction (scripts/2) and optionally classification (scripts/3) of one GeoTIFF on the GPU")
    ap.add_argument("image")
    ap.add_argument("output_dir")
    ap.add_argument("--classify", choices=["kmeans", "rule_based", "random_forest"])
    ap.add_argument("--n-clusters", type=int, default=7)
    ap.add_argument("--no-preprocessing", action="store_true")
    a = ap.parse_args(argv)
    res = run_scripts_2_3(a.image, a.output_dir, a.classify, not a.no_preprocessing, a.n_clusters)
    for k, v in res["paths"].items():
        print(f"{k}: {v}")
    if a.classify:
        cm = res.get("class_map")
        print(f"class map: {None if cm is None else (cm.shape, np.unique(cm).tolist())} -> {res['segmentation_dir']}")
        return 0 if cm is not None else 1
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
