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
    dev = [ctx.upload_f32(np.asarray(b)) for b in bands_data if b is not None]
    planes, ex = P.feature_stack19(ctx, dev, h, w, preprocessing=bool(preprocessing))   # False: bands taken as given (scripts/2:43-50)

    def host(t):
        return t.cpu().numpy().reshape(h, w)

    fd: Dict[str, object] = {k: host(v) for k, v in ex["indices"].items()}
    fd["pca_result"] = [host(p) for p in ex["pca"]]
    fd["variance_ratio"] = ex["pca_ratio"]
    fd["glcm_features"] = {k: host(v) for k, v in ex["glcm"].items()}
    # the window-operator members of the dict (indices.py:320-344, 401-482, 519-562) beyond the three the stack consumes
    from . import _lib as L
    nir2, q255 = ex["nir2"], ex["q255"]  # the re-normalised NIR band and its uint8 image (indices.py:412-415)
    ops = (("erosion", L.MORPH_ERODE), ("dilation", L.MORPH_DILATE), ("opening", L.MORPH_OPEN), ("closing", L.MORPH_CLOSE),
           ("gradient", L.MORPH_GRADIENT))
    fd["morphological_features"] = {f"{name}_{k}": ctx.morph(q255, h, w, k, op).cpu().numpy().reshape(h, w) / 255.0
                                    for k in (3, 5, 7) for name, op in ops}
    ms = {"mean_scale_1": host(nir2), "variance_scale_1": np.zeros((h, w), np.float32), "std_dev_scale_1": np.zeros((h, w), np.float32)}
    for k in (3, 5, 7):
        ms[f"mean_scale_{k}"] = host(ctx.box_mean(nir2, h, w, k, L.BORDER_REFLECT101))
        ms[f"variance_scale_{k}"] = host(ctx.local_var(nir2, h, w, k))
        ms[f"std_dev_scale_{k}"] = host(planes[17]) if k == 5 else host(ctx.local_std(nir2, h, w, k))
    for k in (1, 3, 5):
        e = ctx.rank_entropy(q255, h, w, k).cpu().numpy().reshape(h, w)
        ms[f"entropy_scale_{k}"] = e / np.max(e)
    fd["multi_scale_features"] = ms
    lbp = ctx.lbp_uniform(q255, h, w, 24, 3).cpu().numpy().reshape(h, w).astype(np.float64)
    fd["lbp_feature"] = lbp / lbp.max()
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
    """save_classification_as_geotiff (reference modules/features/extract.py:778-833): one band, nodata 0, LZW in
    256 x 256 tiles; uint8 when the labels fit, else uint16, else int32; float labels are rounded."""
    from .tiff import write_tiff
    a = np.asarray(class_map)
    if a.size and a.max() <= 255 and a.min() >= 0:
        dt = np.uint8
    elif a.size and a.max() <= 65535 and a.min() >= 0:
        dt = np.uint16
    else:
        dt = np.int32
    a = np.round(a).astype(dt) if np.issubdtype(a.dtype, np.floating) else a.astype(dt)
    write_tiff(out_tif, a, transform=transform, epsg=_epsg_of(crs), geographic=_is_geographic(crs), nodata=0, compress="lzw", tiled=True)
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


def run_classification_stage(features_filepath: str, method: str = "kmeans", output_dir: str = "output", n_clusters: int = 7,
                             classifier=None, ctx: Optional[Context] = None) -> Optional[np.ndarray]:
    """run_classification_stage (scripts/3_classification.py:267-505) on a feature file written by stage 2:
    method 'rule_based' (thresholds + morphology + area filter, scripts/3:335-375), 'kmeans' (:377-398) or
    'random_forest' / 'rf' / 'supervised' (:401-488, with a fitted classifier passed in: training is out of scope).
    The shipped script filters the normalised dict with un-prefixed key names (scripts/3:381-383) which never match stage
    2's layout; this driver passes the keys that do exist ('hierarchical_features_all' for the
    clusterer / forest, 'all_extracted_features_dict_<index>' for the rules).
    Writes <output_dir>/classification_<method>.npy and, when the feature file carries transform / crs / width / height
    (scripts/3:495-498), <output_dir>/<method>_classification_map.tif (uint8 labels, nodata 0, LZW tiles).  Returns the
    label map (KMeans labels start at 1, scripts/3:394); None when the features are missing."""
    from modules.features.extract import load_features, normalize_features_structure, unsupervised_kmeans_classification
    feats = normalize_features_structure(load_features(features_filepath))
    key = "hierarchical_features_all"
    os.makedirs(output_dir, exist_ok=True)
    if method == "rule_based":
        from modules.features.extract import rule_based_classification
        idx = {k: feats.get(f"all_extracted_features_dict_{k}") for k in ("ndvi", "ndwi", "mndwi", "ndbi")}
        if idx["ndvi"] is None:
            print(f"特征 'ndvi' 不存在: {list(feats.keys())}")
            return None
        rules = {k: v for k, v in idx.items() if v is not None}
        rules["height"], rules["width"] = feats["height"], feats["width"]
        out = rule_based_classification(rules)
    elif key not in feats:
        print(f"特征 '{key}' 不存在: {list(feats.keys())}")
        return None
    elif method == "kmeans":
        out = (unsupervised_kmeans_classification(feats, n_clusters, [key]) + 1).astype(np.uint8)
    elif method in ("random_forest", "rf", "supervised"):
        if classifier is None:
            raise ValueError("supervised classification needs a fitted classifier")
        from modules.features.extract import supervised_classification_predict
        out = supervised_classification_predict(feats[key], classifier)
    else:
        print(f"错误: 不支持的分割方法 '{method}'")
        return None
    np.save(os.path.join(output_dir, f"classification_{method}.npy"), out)
    if all(feats.get(k) is not None for k in ("transform", "crs", "width", "height")):
        if out.shape == (feats["height"], feats["width"]):
            save_class_map_tif(out, os.path.join(output_dir, f"{method}_classification_map.tif"), feats["transform"], feats["crs"])
    else:
        print("警告: 元数据不完整，无法将分类结果保存为带地理参考的GeoTIFF。")
    return out
