"""
ORACLE — TEST INFRASTRUCTURE ONLY (parity checker; never imported by the product path).

NumPy restatement of the per-pixel feature-extraction -> classification path of
beilsme/rs-image-segmentation.  Every function cites the reference lines it follows
(paths relative to the reference repository root).  Heavy loops (KMeans, GLCM, forest walk)
are in oracle.c and reached through ctypes.

Third-party arithmetic the reference delegates to, and how it is pinned here:
  * NumPy / scikit-learn 1.7.2 are installed: the restatement is checked against the reference's
    own functions (imported with inert cv2/skimage/rasterio stubs) by oracle/gen_golden.py, and the
    resulting vectors are committed under tests/golden/.
  * OpenCV (cv2) and scikit-image are NOT installed and are unpinned in the reference
    (requirements.txt lists neither).  The window operators below restate their published
    semantics; for those stages parity is UNPINNED except end to end through the reference's
    committed output/class_map.npy (tests/test_oracle.py::test_class_map_end_to_end).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
    return _LIB


# --------------------------------------------------------------------------------------------
# stage 1 (only needed to regenerate the bundled tile the reference does not ship)
# --------------------------------------------------------------------------------------------
_GAIN = [0.671339, 1.322205, 1.043976, 0.876024, 0.120354, 0.055376, 0.065551]
_BIAS = [-2.19, -4.16, -2.21, -2.39, -0.49, 1.18, -0.22]


def stage1_preprocess(dn: np.ndarray) -> List[np.ndarray]:
    """modules/features/preprocessing.py:65-72 (gain*DN+bias, float64), :95-96 (identity warp),
    :115-118 (min-max stretch, astype(uint8) truncation), :144 (stored as Float32);
    scripts/2_feature_extraction.py:158 reads it back with .astype(float32)."""
    out = []
    for i in range(dn.shape[0]):
        rad = _GAIN[i] * dn[i] + _BIAS[i]
        mn, mx = np.min(rad), np.max(rad)
        enh = ((rad - mn) * 255.0 / (mx - mn)).astype(np.uint8)
        out.append(enh.astype(np.float32))
    return out


# --------------------------------------------------------------------------------------------
# spectral normalisation + indices   (modules/features/indices.py:25-203)
# --------------------------------------------------------------------------------------------
def robust_normalize(band: np.ndarray, lower_percentile=2, upper_percentile=98) -> np.ndarray:
    """indices.py:25-48.  np.percentile of a float32 array returns float32 scalars and, under
    NumPy-2 promotion, `+ 1e-10` is absorbed: everything stays float32."""
    lo = np.percentile(band, lower_percentile)
    hi = np.percentile(band, upper_percentile)
    clipped = np.clip(band, lo, hi)      # the reference's call: keeps a -0.0 that equals the lower bound (np.maximum would not)
    return (clipped - lo) / (hi - lo + 1e-10)


def _ratio_index(num, den):
    """indices.py:62-69 pattern: zeros where den <= 0.001 (or NaN), num/den elsewhere, clip [-1,1]."""
    with np.errstate(divide="ignore", invalid="ignore"):
        q = num / den
    out = np.where(den > 0.001, q, np.float32(0)).astype(np.float32)
    return np.clip(out, -1.0, 1.0)


def calculate_ndvi(nir, red):  # indices.py:50-71
    return _ratio_index(nir - red, nir + red)


def calculate_evi(nir, red, blue, L=1, C1=6, C2=7.5, G=2.5):  # indices.py:73-95
    den = nir + C1 * red - C2 * blue + L
    return _ratio_index(G * (nir - red), den)


def calculate_msavi(nir, red):  # indices.py:97-114
    a = 2 * nir + 1
    with np.errstate(invalid="ignore"):
        m = (a - np.sqrt(a ** 2 - 8 * (nir - red))) / 2
    return np.clip(m, -1.0, 1.0)


def calculate_ndwi(green, nir):  # indices.py:116-137
    return _ratio_index(green - nir, green + nir)


def calculate_mndwi(green, swir):  # indices.py:139-158
    return _ratio_index(green - swir, green + swir)


def calculate_ndbi(swir, nir):  # indices.py:160-179
    return _ratio_index(swir - nir, swir + nir)


def calculate_bsi(blue, red, nir, swir):  # indices.py:181-203
    a = swir + red
    b = nir + blue
    return _ratio_index(a - b, a + b)


# --------------------------------------------------------------------------------------------
# PCA   (indices.py:205-246 -> sklearn RobustScaler + PCA(covariance_eigh))
# --------------------------------------------------------------------------------------------
def perform_pca(bands: Sequence[np.ndarray], n_components=None, use_robust_scaling=True):
    """indices.py:205-246; sklearn/preprocessing/_data.py:1656-1677, 1716-1718;
    sklearn/decomposition/_pca.py:560, 600-646 (covariance_eigh), _base.py:148-155 (_transform);
    sklearn/utils/extmath.py svd_flip(u_based_decision=False)."""
    h, w = bands[0].shape
    B = len(bands)
    X = np.zeros((h * w, B), dtype=np.float32)
    for i in range(B):
        X[:, i] = bands[i].reshape(-1)
    if use_robust_scaling:
        center = np.nanmedian(X, axis=0)
        q = np.transpose([np.nanpercentile(X[:, j], (25.0, 75.0)) for j in range(B)])
        scale = q[1] - q[0]
        scale[scale < 10 * np.finfo(scale.dtype).eps] = 1.0
        X -= center
        X /= scale
    else:
        X = (X - np.min(X, axis=0)) / (np.max(X, axis=0) - np.min(X, axis=0) + 1e-10)
        center, scale = None, None
    n = X.shape[0]
    nc = min(n, B) if n_components is None else n_components
    mean = np.mean(X, axis=0)
    C = X.T @ X
    C -= n * mean.reshape(-1, 1) * mean.reshape(1, -1)
    C /= n - 1
    evals, evecs = np.linalg.eigh(C)
    evals = evals[::-1].copy()
    evecs = evecs[:, ::-1]
    evals[evals < 0.0] = 0.0
    Vt = evecs.T.copy()
    idx = np.argmax(np.abs(Vt), axis=1)
    signs = np.sign(Vt[np.arange(Vt.shape[0]), idx])
    Vt *= signs[:, None]
    ratio = evals / np.sum(evals)
    comps = Vt[:nc]
    Xt = X @ comps.T
    Xt -= mean.reshape(1, -1) @ comps.T
    result = [Xt[:, i].reshape(h, w) for i in range(nc)]
    model = dict(center=center, scale=scale, mean=mean, components=comps,
                 explained_variance=evals[:nc], cov=C)
    return result, ratio[:nc], model


# --------------------------------------------------------------------------------------------
# window operators — published OpenCV semantics (cv2 absent: parity unpinned, see module header)
# --------------------------------------------------------------------------------------------
def _pad(img, r, mode):
    # cv2.BORDER_REFLECT  = fedcba|abcdefgh|hgfedcb -> numpy 'symmetric'
    # cv2.BORDER_REFLECT_101 (BORDER_DEFAULT) = gfedcb|abcdefgh|gfedcba -> numpy 'reflect'
    return np.pad(img, r, mode={"reflect": "symmetric", "reflect101": "reflect"}[mode])


def box_mean(img: np.ndarray, k: int, border: str) -> np.ndarray:
    """cv2.boxFilter(img32f, -1, (k,k), normalize=True, borderType) / cv2.blur(img32f,(k,k)).
    OpenCV sums float32 sources in float64 (sumType CV_64F) and multiplies by 1/(k*k) before the
    cast back to float32.  The summation ORDER fixed here (the HIP kernel uses the same): row sums
    left-to-right in float64, then the k row sums top-to-bottom in float64."""
    r = k // 2
    p = _pad(img.astype(np.float32), r, border).astype(np.float64)
    H, W = img.shape
    row = np.zeros((H + 2 * r, W), dtype=np.float64)
    for dx in range(k):
        row = p[:, dx:dx + W] if dx == 0 else row + p[:, dx:dx + W]
    acc = None
    for dy in range(k):
        acc = row[dy:dy + H] if dy == 0 else acc + row[dy:dy + H]
    return (acc * (1.0 / (k * k))).astype(np.float32)


def add_spatial_context(features_array: np.ndarray, window_size=7) -> np.ndarray:
    """indices.py:760-776: per channel 7x7 normalised box mean, BORDER_REFLECT, into a float64
    buffer, concatenated after the originals -> (H,W,2C) float64."""
    h, w, c = features_array.shape
    ctx = np.zeros((h, w, c))
    for i in range(c):
        ctx[:, :, i] = box_mean(features_array[:, :, i], window_size, "reflect")
    return np.concatenate([features_array, ctx], axis=-1)


def to_u8(band01: np.ndarray, mult: int = 255) -> np.ndarray:
    """(band * 255).astype(np.uint8) — indices.py:415, 458; (band*(levels-1)).astype(uint8) :268.
    C truncation toward zero; inputs are in [0,1] after robust_normalize."""
    return (band01 * mult).astype(np.uint8)


def morph_gradient_u8(u8: np.ndarray, k: int) -> np.ndarray:
    """cv2.morphologyEx(u8, MORPH_GRADIENT, ones(k,k)) — indices.py:422, 433.  dilate - erode with
    the default constant border that never wins (out-of-image taps ignored)."""
    r = k // 2
    H, W = u8.shape
    lo = np.pad(u8, r, mode="constant", constant_values=255)
    hi = np.pad(u8, r, mode="constant", constant_values=0)
    er = np.full((H, W), 255, np.uint8)
    di = np.zeros((H, W), np.uint8)
    for dy in range(k):
        for dx in range(k):
            er = np.minimum(er, lo[dy:dy + H, dx:dx + W])
            di = np.maximum(di, hi[dy:dy + H, dx:dx + W])
    return di - er


def morph_u8(u8: np.ndarray, k: int, op: str) -> np.ndarray:
    """cv2.erode / cv2.dilate / cv2.morphologyEx(MORPH_OPEN | MORPH_CLOSE) with ones(k,k) — indices.py:421-431.
    Default border: out-of-image taps never win; opening = dilate(erode(x)), closing = erode(dilate(x))."""
    r = k // 2
    H, W = u8.shape

    def erode(x):
        p = np.pad(x, r, mode="constant", constant_values=255)
        o = np.full((H, W), 255, np.uint8)
        for dy in range(k):
            for dx in range(k):
                o = np.minimum(o, p[dy:dy + H, dx:dx + W])
        return o

    def dilate(x):
        p = np.pad(x, r, mode="constant", constant_values=0)
        o = np.zeros((H, W), np.uint8)
        for dy in range(k):
            for dx in range(k):
                o = np.maximum(o, p[dy:dy + H, dx:dx + W])
        return o

    if op == "erosion":
        return erode(u8)
    if op == "dilation":
        return dilate(u8)
    if op == "opening":
        return dilate(erode(u8))
    if op == "closing":
        return erode(dilate(u8))
    if op == "gradient":
        return morph_gradient_u8(u8, k)
    raise ValueError(op)


def calculate_morphological_features(band: np.ndarray) -> dict:
    """indices.py:401-442: 15 float64 members."""
    u8 = to_u8(robust_normalize(band))
    return {f"{name}_{k}": morph_u8(u8, k, name) / 255.0 for k in (3, 5, 7)
            for name in ("erosion", "dilation", "opening", "closing", "gradient")}


def variance_feature(band: np.ndarray, scale: int) -> np.ndarray:
    """calculate_multi_scale_features -> 'variance_scale_k' (indices.py:541-544)."""
    b = robust_normalize(band)
    mean = box_mean(b, scale, "reflect101")
    mean_sq = box_mean(b * b, scale, "reflect101")
    var = mean_sq - mean * mean
    var[var < 0] = 0
    return var


def laplacian_feature(band: np.ndarray) -> np.ndarray:
    """calculate_filter_responses -> 'laplacian' (indices.py:472-474): cv2.Laplacian(u8, CV_32F) (aperture 1: the cross
    stencil [0 1 0; 1 -4 1; 0 1 0], BORDER_REFLECT_101) / 255.0, min-max normalised, float32 throughout."""
    u8 = to_u8(robust_normalize(band)).astype(np.int32)
    p = np.pad(u8, 1, mode="reflect")
    lap = (p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4 * u8).astype(np.float32) / np.float32(255.0)
    mn, mx = lap.min(), lap.max()
    den = np.float32(np.float32(mx - mn) + np.float32(1e-10))
    return ((lap - mn) / den).astype(np.float32)


def gradient_feature(band: np.ndarray, k: int = 5) -> np.ndarray:
    """calculate_morphological_features -> 'gradient_5' (indices.py:411-415, 433, 440): float64."""
    b = robust_normalize(band)
    return morph_gradient_u8(to_u8(b), k) / 255.0


def std_dev_feature(band: np.ndarray, scale: int = 5) -> np.ndarray:
    """calculate_multi_scale_features -> 'std_dev_scale_5' (indices.py:531, 537, 541-548)."""
    b = robust_normalize(band)
    mean = box_mean(b, scale, "reflect101")
    mean_sq = box_mean(b * b, scale, "reflect101")
    var = mean_sq - mean * mean
    var[var < 0] = 0
    return np.sqrt(var)


def sobel_mag_feature(band: np.ndarray) -> np.ndarray:
    """calculate_filter_responses -> 'sobel_mag' (indices.py:455-458, 477-480).  cv2.Sobel(u8, CV_32F,
    1, 0) / (0, 1): 3x3 kernels [-1 0 1]x[1 2 1]^T, BORDER_REFLECT_101; integer-valued, exact."""
    b = robust_normalize(band)
    u = to_u8(b).astype(np.float32)
    p = np.pad(u, 1, mode="reflect")
    H, W = u.shape
    def at(dy, dx):
        return p[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    gx = (at(-1, 1) - at(-1, -1)) + 2 * (at(0, 1) - at(0, -1)) + (at(1, 1) - at(1, -1))
    gy = (at(1, -1) - at(-1, -1)) + 2 * (at(1, 0) - at(-1, 0)) + (at(1, 1) - at(-1, 1))
    sx = gx / 255.0
    sy = gy / 255.0
    mag = np.sqrt(sx ** 2 + sy ** 2)
    return mag / (mag.max() + 1e-10)


def resize_bilinear(src: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(src32f, (out_w, out_h), interpolation=INTER_LINEAR) — indices.py:308.
    Pixel-centre aligned: fx = (float)((dx + 0.5) * scale - 0.5), taps clamped to the image,
    horizontal pass then vertical pass in float32 (plain multiply-add, no FMA)."""
    sh, sw = src.shape
    src = src.astype(np.float32)

    def taps(dn, sn):
        scale = 1.0 / (dn / sn)
        i0 = np.zeros(dn, np.int64)
        i1 = np.zeros(dn, np.int64)
        a0 = np.zeros(dn, np.float32)
        a1 = np.zeros(dn, np.float32)
        for d in range(dn):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(math.floor(f))
            f = np.float32(f - np.float32(s))
            yield_f = f
            if s < 0:
                s, yield_f = 0, np.float32(0)
            if s >= sn - 1:
                s, yield_f = sn - 1, np.float32(0)
            i0[d] = s
            i1[d] = min(s + 1, sn - 1)
            a0[d] = np.float32(1.0) - yield_f
            a1[d] = yield_f
        return i0, i1, a0, a1

    def taps_v(dn, sn):
        # vertical taps keep the fractional weight when clamped (both rows clamp to the edge row)
        scale = 1.0 / (dn / sn)
        i0 = np.zeros(dn, np.int64)
        i1 = np.zeros(dn, np.int64)
        b0 = np.zeros(dn, np.float32)
        b1 = np.zeros(dn, np.float32)
        for d in range(dn):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(math.floor(f))
            f = np.float32(f - np.float32(s))
            i0[d] = min(max(s, 0), sn - 1)
            i1[d] = min(max(s + 1, 0), sn - 1)
            b0[d] = np.float32(1.0) - f
            b1[d] = f
        return i0, i1, b0, b1

    x0, x1, a0, a1 = taps(out_w, sw)
    y0, y1, b0, b1 = taps_v(out_h, sh)
    hrow = src[:, x0] * a0[None, :] + src[:, x1] * a1[None, :]          # (sh, out_w) float32
    out = hrow[y0, :] * b0[:, None] + hrow[y1, :] * b1[:, None]
    return out.astype(np.float32)


def glcm_small_maps(q: np.ndarray, levels: int, window_size: int, step_size: int, mode: int = 1):
    """Windowed GLCM properties on the quantised image (oracle.c::oracle_glcm)."""
    q = np.ascontiguousarray(q, dtype=np.uint8)
    H, W = q.shape
    oh = (H - window_size) // step_size + 1
    ow = (W - window_size) // step_size + 1
    outs = [np.zeros((oh, ow), np.float32) for _ in range(5)]
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib().oracle_glcm(q.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), H, W, levels, window_size,
                           step_size, mode, *[o.ctypes.data_as(fp) for o in outs])
    if rc != 0:
        raise ValueError(f"oracle_glcm rc={rc}")
    return dict(zip(["contrast", "dissimilarity", "homogeneity", "energy", "correlation"], outs))


def glcm_angle(q: np.ndarray, levels: int, angle: int, symmetric: bool = True):
    """One angle (0: 0 deg, 1: 45, 2: 90, 3: 135; distance 1) of the whole image as one window: graycomatrix counts and
    graycoprops' five properties of the symmetric normalised matrix (oracle.c::oracle_glcm_angle)."""
    q = np.ascontiguousarray(q, dtype=np.uint8)
    H, W = q.shape
    counts = np.zeros((levels, levels), np.uint32)
    props = np.zeros(5, np.float64)
    rc = lib().oracle_glcm_angle(q.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), H, W, levels, angle, int(symmetric),
                                 counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), props.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    if rc != 0:
        raise ValueError(f"oracle_glcm_angle rc={rc}")
    return counts, dict(zip(["contrast", "dissimilarity", "homogeneity", "energy", "correlation"], props))


def calculate_glcm_features(band, levels=32, window_size=21, step_size=21, mode: int = 1):
    """indices.py:248-318 (distances=[1], angles=[0, pi/4, pi/2, 3pi/4] are the only values the
    reference ever passes)."""
    b = robust_normalize(band)
    q = to_u8(b, levels - 1)
    h, w = band.shape
    small = glcm_small_maps(q, levels, window_size, step_size, mode)
    return {k: resize_bilinear(v, h, w) for k, v in small.items()}, small


# --------------------------------------------------------------------------------------------
# stage function   (scripts/2_feature_extraction.py:27-133; only members that reach the stack)
# --------------------------------------------------------------------------------------------
def run_feature_extraction_stage(bands_data: Sequence[np.ndarray], preprocessing=True,
                                 glcm_window=21, glcm_step=21, glcm_levels=32, pca_fn=None):
    """scripts/2_feature_extraction.py:27-133.  pca_fn (optional): a perform_pca implementation to use instead of the
    restatement above (oracle/cpu_harness.py passes the scikit-learn objects for the CPU baseline)."""
    if preprocessing:
        bands_data = [robust_normalize(b) for b in bands_data]
    blue, green, red, nir, swir1 = bands_data[:5]
    fd: Dict[str, object] = {}
    fd["ndvi"] = calculate_ndvi(nir, red)
    fd["evi"] = calculate_evi(nir, red, blue)
    fd["msavi"] = calculate_msavi(nir, red)
    fd["ndwi"] = calculate_ndwi(green, nir)
    fd["mndwi"] = calculate_mndwi(green, swir1)
    fd["ndbi"] = calculate_ndbi(swir1, nir)
    fd["bsi"] = calculate_bsi(blue, red, nir, swir1)
    pca_result, ratio, model = (pca_fn or perform_pca)([b for b in bands_data if b is not None])
    fd["pca_result"] = pca_result
    fd["variance_ratio"] = ratio
    fd["glcm_features"], _ = calculate_glcm_features(nir, glcm_levels, glcm_window, glcm_step)
    fd["multi_scale_features"] = {"std_dev_scale_5": std_dev_feature(nir, 5)}
    fd["morphological_features"] = {"gradient_5": gradient_feature(nir, 5)}
    fd["filter_features"] = {"sobel_mag": sobel_mag_feature(nir)}
    # indices.py:808-835
    level1 = np.stack([fd["ndwi"], fd["mndwi"], fd["ndvi"], fd["evi"], fd["ndbi"], fd["bsi"],
                       fd["pca_result"][0]], axis=-1)
    # indices.py:837-865
    level2 = np.stack([fd["glcm_features"]["contrast"], fd["glcm_features"]["homogeneity"],
                       fd["morphological_features"]["gradient_5"],
                       fd["multi_scale_features"]["std_dev_scale_5"],
                       fd["filter_features"]["sobel_mag"]], axis=-1)
    level1_ctx = add_spatial_context(level1)
    hier = {"level_1": level1_ctx, "level_2": level2,
            "all": np.concatenate([level1_ctx, level2], axis=-1)}
    return fd, hier


def full_features_dict(bands_data: Sequence[np.ndarray], preprocessing=True, glcm_window=21, glcm_step=21, glcm_levels=32,
                       pca_result=None) -> Dict[str, object]:
    """EVERY member of features_dict as scripts/2_feature_extraction.py:62-106 fills it, in the reference's insertion
    order (the order decides the column order of unsupervised_kmeans_classification's default key selection,
    extract.py:516-522): 7 indices, pca_result, variance_ratio, glcm_features (indices.py:310-316), lbp_feature,
    multi_scale_features (per scale 1/3/5/7: mean, variance, std_dev, entropy for scales <= 5, indices.py:535-560),
    morphological_features (per size 3/5/7: erosion .. gradient, :421-440), filter_features (:463-480).
    pca_result (optional): the component planes to use instead of the restatement's (fixtures take the reference's)."""
    if preprocessing:
        bands_data = [robust_normalize(b) for b in bands_data]
    blue, green, red, nir, swir1 = bands_data[:5]
    fd: Dict[str, object] = {}
    fd["ndvi"] = calculate_ndvi(nir, red)
    fd["evi"] = calculate_evi(nir, red, blue)
    fd["msavi"] = calculate_msavi(nir, red)
    fd["ndwi"] = calculate_ndwi(green, nir)
    fd["mndwi"] = calculate_mndwi(green, swir1)
    fd["ndbi"] = calculate_ndbi(swir1, nir)
    fd["bsi"] = calculate_bsi(blue, red, nir, swir1)
    if pca_result is None:
        pca_result, ratio, _ = perform_pca([b for b in bands_data if b is not None])
    else:
        ratio = np.zeros(len(pca_result), np.float32)
    fd["pca_result"] = list(pca_result)
    fd["variance_ratio"] = ratio
    fd["glcm_features"], _ = calculate_glcm_features(nir, glcm_levels, glcm_window, glcm_step)
    b = robust_normalize(nir)            # every texture function re-normalises the band it receives
    u8 = to_u8(b)
    lbp = lbp_uniform(u8, 24, 3)
    fd["lbp_feature"] = lbp / lbp.max()
    ms: Dict[str, np.ndarray] = {}
    for scale in (1, 3, 5, 7):
        mean = b.copy() if scale == 1 else box_mean(b, scale, "reflect101")       # cv2.blur with a 1x1 kernel is the identity
        mean_sq = b * b if scale == 1 else box_mean(b * b, scale, "reflect101")
        var = mean_sq - mean * mean
        var[var < 0] = 0
        ms[f"mean_scale_{scale}"] = mean
        ms[f"variance_scale_{scale}"] = var
        ms[f"std_dev_scale_{scale}"] = np.sqrt(var)
        if scale <= 5:
            e = rank_entropy(u8, scale)
            ms[f"entropy_scale_{scale}"] = e / np.max(e)
    fd["multi_scale_features"] = ms
    fd["morphological_features"] = calculate_morphological_features(nir)
    ff = filter_responses_extra(nir)
    ff["laplacian"] = laplacian_feature(nir)
    ff["sobel_mag"] = sobel_mag_feature(nir)
    fd["filter_features"] = ff
    return fd


def flatten_features_dict(fd: Dict[str, object], prefix: str = "all_extracted_features_dict") -> Dict[str, np.ndarray]:
    """The 2-D members of a stage-2 features_dict under the keys normalize_features_structure gives them
    (extract.py:139-174: '<outer>_<inner>' lower-case, list members by index), in the same order."""
    out: Dict[str, np.ndarray] = {}

    def walk(v, key):
        if isinstance(v, np.ndarray) and v.ndim >= 2:
            out.setdefault(key.lower(), v)
        elif isinstance(v, dict):
            for k2, v2 in v.items():
                walk(v2, f"{key}_{k2}")
        elif isinstance(v, list):
            for i, v2 in enumerate(v):
                walk(v2, f"{key}_{i}")

    for k, v in fd.items():
        walk(v, f"{prefix}_{k}" if prefix else k)
    return out


# --------------------------------------------------------------------------------------------
# LBP, rank entropy, fixed-point Gaussian (indices.py:320-344, 551-560, 463-470) — scikit-image / OpenCV semantics restated
# from their published algorithms (libraries absent, unpinned in the reference: parity unpinned)
# --------------------------------------------------------------------------------------------
def lbp_uniform(u8: np.ndarray, P: int = 24, R: float = 3) -> np.ndarray:
    """skimage.feature.local_binary_pattern(u8, P, R, method='uniform') (texture.py + _texture.pyx):
    float64 image, offsets np.round(+-R sin/cos(2 pi i / P), 5), bilinear interpolation with constant 0 outside,
    s_i = sample - centre >= 0, changes over the P - 1 consecutive pairs, code = sum s_i if changes <= 2 else P + 1."""
    img = u8.astype(np.float64)
    H, W = img.shape
    rp = np.round(-R * np.sin(2 * np.pi * np.arange(P, dtype=np.float64) / P), 5)
    cp = np.round(R * np.cos(2 * np.pi * np.arange(P, dtype=np.float64) / P), 5)
    rr, cc = np.mgrid[0:H, 0:W].astype(np.float64)

    def px(r, c):
        ok = (r >= 0) & (r < H) & (c >= 0) & (c < W)
        out = np.zeros(r.shape, np.float64)
        out[ok] = img[r[ok], c[ok]]
        return out

    s = np.zeros((P, H, W), np.int8)
    for i in range(P):
        r, c = rr + rp[i], cc + cp[i]
        minr, minc = np.floor(r).astype(np.int64), np.floor(c).astype(np.int64)
        maxr, maxc = np.ceil(r).astype(np.int64), np.ceil(c).astype(np.int64)
        dr, dc = r - minr, c - minc
        top = (1 - dc) * px(minr, minc) + dc * px(minr, maxc)
        bottom = (1 - dc) * px(maxr, minc) + dc * px(maxr, maxc)
        s[i] = ((1 - dr) * top + dr * bottom) - img >= 0
    changes = (s[:-1] != s[1:]).sum(0)
    return np.where(changes <= 2, s.sum(0), P + 1).astype(np.float64)


def disk(radius: int) -> np.ndarray:
    """skimage.morphology.disk"""
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return (X ** 2 + Y ** 2 <= radius ** 2).astype(np.uint8)


def rank_entropy(u8: np.ndarray, radius: int) -> np.ndarray:
    """skimage.filters.rank.entropy(u8, disk(radius)) (rank/generic_cy.pyx::_kernel_entropy): local histogram over the
    in-image part of the footprint, e = - sum_i p_i log(p_i) / ln 2 over the bins in ascending order, float64."""
    H, W = u8.shape
    fp = disk(radius)
    out = np.zeros((H, W), np.float64)
    offs = [(dy - radius, dx - radius) for dy in range(2 * radius + 1) for dx in range(2 * radius + 1) if fp[dy, dx]]
    for r in range(H):
        for c in range(W):
            vals = [u8[r + dy, c + dx] for dy, dx in offs if 0 <= r + dy < H and 0 <= c + dx < W]
            cnt = np.bincount(vals, minlength=256)
            e = 0.0
            pop = float(len(vals))
            for i in np.nonzero(cnt)[0]:
                p = cnt[i] / pop
                e -= p * math.log(p) / 0.6931471805599453
            out[r, c] = e
    return out


def gaussian_taps_fixed(ksize: int) -> np.ndarray:
    """OpenCV's 8-bit fixed-point Gaussian taps for sigma = 0 (imgproc/src/smooth.dispatch.cpp: getGaussianKernel +
    getGaussianKernelFixedPoint_ED): the small fixed tables up to 7 taps, else exp(-x^2 / 2 sigma^2) with
    sigma = 0.3 ((k - 1) / 2 - 1) + 0.8, normalised; rounded half-to-even with error diffusion, centre = 256 - rest."""
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if ksize in small:
        k = np.array(small[ksize], np.float64)
    else:
        sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
        x = np.arange(ksize) - (ksize - 1) * 0.5
        k = np.exp(-0.5 / (sigma * sigma) * x * x)
        k = k * (1.0 / k.sum())
    taps = np.zeros(ksize, np.int64)
    err = 0.0
    for i in range(ksize // 2):
        adj = k[i] * 256.0 + err
        v0 = int(np.rint(adj))
        err = adj - v0
        taps[i] = taps[ksize - 1 - i] = v0
    taps[ksize // 2] = 256 - 2 * taps[:ksize // 2].sum()
    return taps


def gaussian_blur_u8(u8: np.ndarray, ksize: int) -> np.ndarray:
    """cv2.GaussianBlur(u8, (k, k), 0): rows in 8.8 fixed point, columns in 16.16, (acc + 2^15) >> 16, BORDER_REFLECT_101."""
    taps = gaussian_taps_fixed(ksize)
    r = ksize // 2
    p = np.pad(u8.astype(np.int64), r, mode="reflect")
    H, W = u8.shape
    hor = sum(p[:, t:t + W] * taps[t] for t in range(ksize))             # (H + 2r, W), at most 255 * 256
    acc = sum(hor[t:t + H] * taps[t] for t in range(ksize))
    return np.minimum((acc + 32768) >> 16, 255).astype(np.uint8)


def filter_responses_extra(band: np.ndarray) -> dict:
    """calculate_filter_responses' gaussian_5, gaussian_15, dog (indices.py:463-470)."""
    u8 = to_u8(robust_normalize(band))
    g5 = gaussian_blur_u8(u8, 5) / 255.0
    g15 = gaussian_blur_u8(u8, 15) / 255.0
    dog = g5 - g15
    return {"gaussian_5": g5, "gaussian_15": g15, "dog": (dog - dog.min()) / (dog.max() - dog.min() + 1e-10)}


# --------------------------------------------------------------------------------------------
# rule-based classification   (modules/features/extract.py:299-505; scripts/3_classification.py:335-375)
# scipy.ndimage.label is the reference's own dependency and is installed: the component filter below CALLS it, so
# that part of the oracle is the real library.  cv2.morphologyEx with cv2.getStructuringElement(MORPH_ELLIPSE) is
# restated (cv2 absent): (3,3) = cross, (5,5) = 5x5 without the corner pairs; out-of-image taps never win.
# --------------------------------------------------------------------------------------------
def ellipse_element(k: int) -> np.ndarray:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)) restated from OpenCV's imgproc/morph.cpp: r = c = k // 2,
    row i (dy = i - r) is 1 for |j - c| <= cvRound(c * sqrt((r*r - dy*dy) / (r*r))); cvRound rounds half to even.
    (3,3) is the cross; (5,5) the square without its corner pairs."""
    r = c = k // 2
    se = np.zeros((k, k), np.uint8)
    inv_r2 = 1.0 / (r * r) if r else 0.0
    for i in range(k):
        dy = i - r
        dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
        se[i, max(c - dx, 0):min(c + dx + 1, k)] = 1
    return se


def morph_ellipse(mask: np.ndarray, k: int, op: str) -> np.ndarray:
    se = ellipse_element(k)
    r = k // 2
    H, W = mask.shape

    def erode(x):
        p = np.pad(x, r, mode="constant", constant_values=255)
        o = np.full((H, W), 255, np.uint8)
        for dy in range(k):
            for dx in range(k):
                if se[dy, dx]:
                    o = np.minimum(o, p[dy:dy + H, dx:dx + W])
        return o

    def dilate(x):
        p = np.pad(x, r, mode="constant", constant_values=0)
        o = np.zeros((H, W), np.uint8)
        for dy in range(k):
            for dx in range(k):
                if se[dy, dx]:
                    o = np.maximum(o, p[dy:dy + H, dx:dx + W])
        return o

    x = mask.astype(np.uint8)
    return {"erosion": lambda: erode(x), "dilation": lambda: dilate(x), "opening": lambda: dilate(erode(x)),
            "closing": lambda: erode(dilate(x))}[op]()


def advanced_post_processing(binary_mask: np.ndarray, min_area=100, smooth_kernel_size=3, fill_holes=True) -> np.ndarray:
    """extract.py:299-341.  The hole fill of the even / zero kernel branch CALLS scipy.ndimage.binary_fill_holes and the
    component filter scipy.ndimage.label — the reference's own dependency, installed."""
    from scipy import ndimage
    k = smooth_kernel_size
    odd = k > 0 and k % 2 == 1
    m = binary_mask.astype(np.uint8)
    if fill_holes and odd:
        m = morph_ellipse(m, k, "closing")
    elif fill_holes:
        m = ndimage.binary_fill_holes(m).astype(np.uint8)
    if min_area > 0:
        lab, nf = ndimage.label(m, structure=np.ones((3, 3)))
        if nf > 0:
            area = np.bincount(lab.ravel())
            rm = np.where((area < min_area) & (area > 0))[0]
            if rm.size > 0:
                m[np.isin(lab, rm)] = 0
    return morph_ellipse(m, k, "opening") if odd else m


def otsu_level_u8(u8: np.ndarray) -> int:
    """cv2.threshold(u8, 0, 255, THRESH_BINARY + THRESH_OTSU)'s level, restated from OpenCV's imgproc/thresh.cpp
    (getThreshVal_Otsu_8u; cv2 is absent: RESTATEMENT ONLY, pinned by hand-derived cases in tests/test_oracle.py): scan
    the 256 levels upwards, keep the first level with the largest between-class variance q1*q2*(mu1-mu2)^2, skipping
    levels whose class weights are within FLT_EPSILON of 0 or 1; float64."""
    h = np.bincount(u8.ravel(), minlength=256).astype(np.float64)
    scale = 1.0 / u8.size
    mu = 0.0
    for i in range(256):            # the same left-to-right float64 sum as the C loop
        mu += i * h[i]
    mu *= scale
    eps = float(np.finfo(np.float32).eps)
    mu1 = q1 = max_sigma = 0.0
    best = 0
    for i in range(256):
        p_i = h[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > max_sigma:
            max_sigma, best = sigma, i
    return best


def threshold_segmentation(feature_image: np.ndarray, threshold_value, above=True, otsu=False) -> np.ndarray:
    """extract.py:344-404 (without the cv2.error fallback, which valid input never takes)."""
    if np.isnan(feature_image).any():
        feature_image = np.nan_to_num(feature_image, nan=0.0)
    if otsu:
        mn, mx = np.min(feature_image), np.max(feature_image)
        if mx == mn:
            return np.zeros_like(feature_image, dtype=np.uint8) if above else np.ones_like(feature_image, dtype=np.uint8)
        norm = np.clip(((feature_image - mn) / (mx - mn + 1e-10) * 255), 0, 255).astype(np.uint8)
        mask = (norm > otsu_level_u8(norm)).astype(np.uint8)
        return mask if above else (1 - mask).astype(np.uint8)
    return (feature_image > threshold_value).astype(np.uint8) if above else (feature_image < threshold_value).astype(np.uint8)


def rule_based_classification(features: dict) -> np.ndarray:
    """scripts/3_classification.py:335-375 with the extract_* functions of extract.py:406-505 inlined."""
    h, w = features["height"], features["width"]
    n = h * w

    def thr(x, t, above=True):
        x = np.nan_to_num(x, nan=0.0)
        return (x > t).astype(np.uint8) if above else (x < t).astype(np.uint8)

    ndvi, ndbi = features.get("ndvi"), features.get("ndbi")
    final = np.zeros((h, w), np.uint8)
    veg = advanced_post_processing(thr(ndvi, 0.25), int(n * 0.0005), 3)
    if features.get("mndwi") is not None:
        water = advanced_post_processing(thr(features["mndwi"], 0.1), int(n * 0.0002), 3)
    else:
        water = advanced_post_processing(thr(features["ndwi"], 0.05), int(n * 0.0002), 3)
    built = advanced_post_processing(np.logical_and(thr(ndbi, 0.0), thr(ndvi, 0.2, above=False)).astype(np.uint8), int(n * 0.001), 5)
    final[built == 1] = 3
    final[veg == 1] = 1
    final[water == 1] = 2
    excl = (final == 1) | (final == 2) | (final == 3)
    bare = np.logical_not(excl).astype(np.uint8)
    bare = np.logical_and(bare, np.logical_and(ndvi > -0.1, ndvi < 0.2)).astype(np.uint8)
    bare = np.logical_and(bare, np.logical_and(ndbi > -0.2, ndbi < 0.2)).astype(np.uint8)
    bare = advanced_post_processing(bare, int(n * 0.0005), 3)
    final[(bare == 1) & (final == 0)] = 4
    return final


# --------------------------------------------------------------------------------------------
# KMeans   (modules/features/extract.py:508-581 -> oracle.c)
# --------------------------------------------------------------------------------------------
def kmeans_random_draws(n: int, k: int, dtype) -> Tuple[int, np.ndarray]:
    """The two things sklearn asks RandomState(42) for (sklearn/cluster/_kmeans.py:225, 243)."""
    rs = np.random.RandomState(42)
    w = np.ones(n, dtype=dtype)
    center_id = int(rs.choice(n, p=w / w.sum()))
    L = 2 + int(np.log(k))
    u = np.concatenate([rs.uniform(size=L) for _ in range(k - 1)]) if k > 1 else np.zeros(0)
    return center_id, u


def kmeans_fit_planes(planes: Sequence[np.ndarray], n_clusters: int, max_iter=300, tol=1e-4):
    """planes: list of equally shaped arrays, all float32 or all float64 (one per feature column)."""
    dt = np.result_type(*[p.dtype for p in planes])
    if dt not in (np.float32, np.float64):
        dt = np.dtype(np.float64)
    P = [np.ascontiguousarray(p.reshape(-1), dtype=dt) for p in planes]
    n, F = P[0].size, len(P)
    cid, u = kmeans_random_draws(n, n_clusters, dt)
    ctype = ctypes.c_float if dt == np.float32 else ctypes.c_double
    arr = (ctypes.POINTER(ctype) * F)(*[p.ctypes.data_as(ctypes.POINTER(ctype)) for p in P])
    labels = np.zeros(n, np.int32)
    centers = np.zeros((n_clusters, F), np.float64)
    n_iter = ctypes.c_int32(0)
    reloc = ctypes.c_int32(0)
    scale = np.zeros(F); minv = np.zeros(F); mean = np.zeros(F)
    tol_out = ctypes.c_double(0)
    init_idx = np.zeros(n_clusters, np.int64)
    fn = lib().oracle_kmeans_fit_f32 if dt == np.float32 else lib().oracle_kmeans_fit_f64
    dp = ctypes.POINTER(ctypes.c_double)
    rc = fn(arr, ctypes.c_int64(n), F, n_clusters, ctypes.c_int64(cid), u.ctypes.data_as(dp), max_iter,
            ctypes.c_double(tol), labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
            centers.ctypes.data_as(dp), ctypes.byref(n_iter), scale.ctypes.data_as(dp),
            minv.ctypes.data_as(dp), mean.ctypes.data_as(dp), ctypes.byref(tol_out),
            init_idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.byref(reloc))
    if rc != 0:
        raise ValueError(f"oracle_kmeans_fit rc={rc}")
    info = dict(n_iter=n_iter.value, centers=centers, scale=scale, min=minv, mean=mean,
                tol=tol_out.value, init_indices=init_idx, relocated=reloc.value)
    return labels, info


def select_feature_planes(features_dict: dict, feature_keys_to_use=None) -> List[np.ndarray]:
    """extract.py:510-566: key selection and per-channel flattening, NaN -> 0 handled downstream."""
    if not features_dict or "height" not in features_dict or "width" not in features_dict:
        raise ValueError("feature dict empty or missing height/width")
    shape = (features_dict["height"], features_dict["width"])
    if feature_keys_to_use is None:
        meta = ["transform", "crs", "width", "height", "dimensions", "geo_transform"]
        keys = [k for k, v in features_dict.items()
                if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == shape and k not in meta]
        if not keys:
            cands = ["ndvi", "ndwi", "ndbi", "texture_mean", "evi", "savi",
                     "hierarchical_level_1", "hierarchical_level_2", "hierarchical_all"]
            keys = [k for k in cands if k in features_dict and isinstance(features_dict[k], np.ndarray)
                    and ((features_dict[k].ndim == 2 and features_dict[k].shape == shape)
                         or (features_dict[k].ndim == 3 and features_dict[k].shape[:2] == shape))]
        feature_keys_to_use = keys
    if not feature_keys_to_use:
        raise ValueError("no features available for K-Means")
    planes = []
    for key in feature_keys_to_use:
        v = features_dict.get(key)
        if isinstance(v, np.ndarray) and v.ndim == 3 and v.shape[:2] == shape:
            planes.extend(v[:, :, i] for i in range(v.shape[2]))
        elif isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == shape:
            planes.append(v)
    if not planes:
        raise ValueError("no feature data prepared for K-Means")
    return planes


def unsupervised_kmeans_classification(features_dict, n_clusters=5, feature_keys_to_use=None):
    planes = select_feature_planes(features_dict, feature_keys_to_use)
    labels, info = kmeans_fit_planes(planes, n_clusters)
    return labels.reshape(features_dict["height"], features_dict["width"]), info


# --------------------------------------------------------------------------------------------
# random-forest inference   (modules/supervised_classifiers.py:99-115 -> oracle.c)
# --------------------------------------------------------------------------------------------
def flatten_forest(model) -> dict:
    """sklearn RandomForestClassifier -> flat arrays (tree_ node records)."""
    offs, left, right, feat, thr, miss, val = [0], [], [], [], [], [], []
    for est in model.estimators_:
        t = est.tree_
        left.append(t.children_left.astype(np.int32))
        right.append(t.children_right.astype(np.int32))
        feat.append(t.feature.astype(np.int32))
        thr.append(t.threshold.astype(np.float64))
        mg = getattr(t, "missing_go_to_left", None)
        miss.append(np.zeros(t.node_count, np.uint8) if mg is None else np.asarray(mg, np.uint8))
        val.append(t.value[:, 0, :len(model.classes_)].astype(np.float64))
        offs.append(offs[-1] + t.node_count)
    return dict(tree_off=np.asarray(offs, np.int64), left=np.concatenate(left), right=np.concatenate(right),
                feature=np.concatenate(feat), threshold=np.concatenate(thr),
                missing_left=np.concatenate(miss), value=np.ascontiguousarray(np.concatenate(val)),
                classes=np.asarray(model.classes_, np.int64), n_features=int(model.n_features_in_))


def rf_predict_planes(forest: dict, planes: Sequence[np.ndarray]) -> np.ndarray:
    P = [np.ascontiguousarray(np.asarray(p).reshape(-1), dtype=np.float32) for p in planes]
    n, F = P[0].size, len(P)
    fp = ctypes.POINTER(ctypes.c_float)
    arr = (fp * F)(*[p.ctypes.data_as(fp) for p in P])
    out = np.zeros(n, np.int64)
    ip = ctypes.POINTER(ctypes.c_int32)
    lp = ctypes.POINTER(ctypes.c_int64)
    dp = ctypes.POINTER(ctypes.c_double)
    f = forest
    feat = np.where(f["feature"] < 0, 0, f["feature"]).astype(np.int32)
    rc = lib().oracle_rf_predict(arr, ctypes.c_int64(n), F, len(f["tree_off"]) - 1,
                                 f["tree_off"].ctypes.data_as(lp), f["left"].ctypes.data_as(ip),
                                 f["right"].ctypes.data_as(ip), feat.ctypes.data_as(ip),
                                 f["threshold"].ctypes.data_as(dp),
                                 f["missing_left"].ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                 f["value"].ctypes.data_as(dp), f["value"].shape[1],
                                 f["classes"].ctypes.data_as(lp), out.ctypes.data_as(lp))
    if rc != 0:
        raise ValueError(f"oracle_rf_predict rc={rc}")
    return out


def predict_image(forest: dict, features: np.ndarray) -> np.ndarray:
    """supervised_classifiers.py:99-115: (H,W,D) -> (H,W) int64."""
    h, w, d = features.shape
    planes = [features[:, :, i].astype(np.float32) for i in range(d)]
    return rf_predict_planes(forest, planes).reshape(h, w)


# --------------------------------------------------------------------------------------------
# synthetic raster of SURVEY.md §8(d) — shared by tests and bench (inputs only, no reference code)
# --------------------------------------------------------------------------------------------
def synthetic_stripe(stripe: int, rows: int, width: int, row0: int, bands: int = 7) -> np.ndarray:
    proto = np.random.default_rng(355).integers(20, 230, (8, bands))
    rng = np.random.default_rng([355, stripe])
    y = (np.arange(rows) + row0)[:, None]
    x = np.arange(width)[None, :]
    lab = ((y // 64) * 7 + (x // 64) * 3) % 8
    out = np.empty((bands, rows, width), np.float32)
    for b in range(bands):
        v = proto[lab, b] + rng.normal(0.0, 6.0, (rows, width))
        out[b] = np.clip(v, 0, 255).astype(np.uint8).astype(np.float32)
    return out


def synthetic_raster(height: int, width: int, bands: int = 7, stripe_rows: int = 1024) -> np.ndarray:
    parts = []
    for s, r0 in enumerate(range(0, height, stripe_rows)):
        parts.append(synthetic_stripe(s, min(stripe_rows, height - r0), width, r0, bands))
    return np.concatenate(parts, axis=1)
