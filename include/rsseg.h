/*
 * rsseg.h — C ABI of librsseg_hip.so: MI355X (gfx950) kernels for the per-pixel
 * feature-extraction -> clustering / classification path of beilsme/rs-image-segmentation.
 *
 * The reference has no FFI: its boundary for this path is a set of NumPy-in / NumPy-out Python
 * functions.  Each entry point below names the reference function (file:line,
 * relative to the reference repository) whose arithmetic it replaces; the Python mirror of those
 * functions (rs-image-segmentation_amd/modules/...) binds this header through ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - Pointers named d_* are DEVICE addresses (e.g. torch tensor .data_ptr()); all others are host.
 *   - Rasters are band/feature-PLANAR: one contiguous (H*W) plane per band, row-major.
 *   - Every call is synchronous with respect to its results: it returns after the work it
 *     enqueued on the context's stream has completed, unless stated otherwise.
 *   - Return value: 0 = RSSEG_OK, negative = error (rsseg_last_error gives the text).  Never aborts.
 *   - One caller thread per context (the reference is single-threaded).
 *   - Multi-GPU: one context per process/GPU; a context given world > 1 calls the registered
 *     all-reduce hook on small device buffers (RCCL through torch.distributed on the host side).
 */
#ifndef RSSEG_H
#define RSSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSSEG_OK 0
#define RSSEG_ERR_INVALID (-1)   /* bad argument (ValueError on the Python side) */
#define RSSEG_ERR_HIP (-2)       /* HIP runtime error */
#define RSSEG_ERR_NOMEM (-3)
#define RSSEG_ERR_COMM (-4)      /* all-reduce hook failed */
#define RSSEG_ERR_UNSUPPORTED (-5)

#define RSSEG_F32 0
#define RSSEG_F64 1
#define RSSEG_I64 2

#define RSSEG_SUM 0
#define RSSEG_MIN 1
#define RSSEG_MAX 2

#define RSSEG_BORDER_REFLECT 0     /* cv2.BORDER_REFLECT      fedcba|abcdefgh|hgfedcb */
#define RSSEG_BORDER_REFLECT101 1  /* cv2.BORDER_REFLECT_101  gfedcb|abcdefgh|gfedcba */

#define RSSEG_MAX_FEATURES 64
#define RSSEG_MAX_CLUSTERS 64
#define RSSEG_MAX_RANKS 16

typedef struct rsseg_ctx rsseg_ctx;

/* All-reduce hook: reduce `count` elements of `dtype` (RSSEG_F32/F64/I64) with `op`
 * (RSSEG_SUM/MIN/MAX) IN PLACE at byte offset `offset` of the communication buffer registered with
 * rsseg_ctx_set_comm, across all ranks, ordered after the work already enqueued on the context's
 * stream AND before the work the library enqueues on that stream after the hook has returned: inside
 * rsseg_kmeans_fit_predict a kernel leaves its partials in the buffer, the hook is called, and the next kernel reads
 * the reduced values, without the host waiting in between (ncclAllReduce on the context's stream, or
 * torch.distributed.all_reduce with that stream current, give exactly this; a hook that synchronises is also
 * correct).  Every rank calls the hook the same number of times with the same arguments.  Returns 0 on success. */
typedef int (*rsseg_allreduce_fn)(void *user, int64_t offset, int64_t count, int dtype, int op);

/* ---- context ---------------------------------------------------------------------------- */
/* stream: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream), or NULL for a non-blocking
 * stream owned by the context.  The library orders its work on THAT stream only: an input plane still being written
 * by work on another stream must be complete before the call.  A host whose producers run on the legacy default
 * stream (torch's default) passes hipStreamLegacy ((hipStream_t)1), as rsseg/runtime.py does. */
int rsseg_ctx_create(int device, void *stream, rsseg_ctx **out);
void rsseg_ctx_destroy(rsseg_ctx *ctx);
const char *rsseg_last_error(const rsseg_ctx *ctx);
const char *rsseg_version(void);
/* While on, the spectral-index call (7 index planes), the two bilinear-upsample calls (1 plane) and the PCA call
 * (n_components planes) also reduce the minimum and maximum of every plane they write, NaN counted
 * as 0 (what KMeans' MinMaxScaler sees); rsseg_ctx_last_minmax returns them, by output index (at most 16), until the next such call. */
int rsseg_ctx_collect_minmax(rsseg_ctx *ctx, int on);
int rsseg_ctx_last_minmax(rsseg_ctx *ctx, int plane, double *mn, double *mx);
/* Asynchronous mode: entry points whose results stay on the device (normalize, indices, quantize, glcm, resize,
 * window operators, forest_predict) return right after enqueueing on the context's stream; rsseg_ctx_sync waits.
 * Used to run the VALU-bound GLCM on a second stream beside the HBM-bound passes (rsseg/pipeline.py). */
int rsseg_ctx_set_async(rsseg_ctx *ctx, int on);
int rsseg_ctx_sync(rsseg_ctx *ctx);
/* d_comm: device buffer of comm_bytes (>= 1 MiB) owned by the caller, visible to the hook.  The hook is called whenever one is
 * installed together with a buffer — also with world == 1, where the reduction is the identity: that is how a one-GPU box
 * runs every collective of a step through RCCL.  fn == NULL with world == 1 removes it. */
int rsseg_ctx_set_comm(rsseg_ctx *ctx, int rank, int world, rsseg_allreduce_fn fn, void *user,
                       void *d_comm, size_t comm_bytes);

/* The same with RCCL driven by the library itself (no callback, no Python in the loop): every reduction becomes an
 * ncclAllReduce, in place in the communication buffer, enqueued on the context's stream.  Rank 0 obtains the 128-byte
 * ncclUniqueId with rsseg_rccl_unique_id and the host hands it to every rank by whatever means it has (torch.distributed's
 * store, MPI, a file); then EVERY rank calls rsseg_ctx_set_comm_rccl (collective: ncclCommInitRank).  librccl_path names
 * the librccl.so the process uses (torch ships one in torch/lib; NULL: "librccl.so.1" from the loader path); it is
 * dlopen()ed, not linked.  d_comm may be NULL: the library then owns a 4 MiB buffer.  world == 1 is allowed (identity
 * reductions through RCCL).  rsseg_ctx_set_comm or rsseg_ctx_destroy release the communicator.
 * (This is the form SURVEY.md 8b sketched as rsseg_ctx_create(device, rank, world, ncclUniqueId*).) */
#define RSSEG_RCCL_ID_BYTES 128
int rsseg_rccl_unique_id(const char *librccl_path, void *id_out);
int rsseg_ctx_set_comm_rccl(rsseg_ctx *ctx, int rank, int world, const void *unique_id, const char *librccl_path, void *d_comm,
                            size_t comm_bytes);

/* One reduction over whatever communication path is installed (callback or RCCL), as the library's own kernels use it:
 * `count` elements of `dtype` at byte `offset` of the communication buffer, in place, enqueued on the context's stream
 * (no host wait).  Without a communication path (one rank) it is the identity.  For hosts that keep cross-rank values of
 * their own in the buffer, and for testing a path's stream ordering. */
int rsseg_ctx_allreduce(rsseg_ctx *ctx, int64_t offset, int64_t count, int dtype, int op);

/* Per-kernel timing (HIP events on the context's stream, around each launch of the named
 * kernel family).  Used by bench.py for the roofline object.  name: "glcm", "lloyd", "indices", "normalize", "quantize", "range",
 * "select", "kpp", "moment" (KMeans' column means), "labels" (uint8 -> int32 label plane), "box" (one plane), "ctxmean" (several planes per launch), "morph", "filt_max" / "filt_write" (the two passes
 * of Sobel / Laplacian), "project", "indices_project" (the fused pass of rsseg_indices_pca_*), "gram", "forest", "resize", and "allreduce" (host wall time of the hook calls). */
/* Number of times the library has made the host wait for the context's stream since the last reset (every result
 * read-back, every all-reduce staged through the host): what a step costs in launch-pipeline bubbles. */
int rsseg_ctx_host_syncs(rsseg_ctx *ctx, int reset, int64_t *count);
int rsseg_prof_enable(rsseg_ctx *ctx, int on);
int rsseg_prof_reset(rsseg_ctx *ctx);
int rsseg_prof_get(rsseg_ctx *ctx, const char *name, double *total_ms, int64_t *launches);

/* ---- K1: exact order statistics ------------------------------------------------------- */
/* Replaces the partition inside np.percentile (modules/features/indices.py:38-39) and
 * np.nanmedian / np.nanpercentile of RobustScaler (indices.py:230-231).
 * For each r in ranks[0..nranks) (0-based positions in the ascending order of ALL n_global
 * values, NaNs last) writes the value at that position to out_values.  n_nan_out receives the
 * global NaN count.  The interpolation between neighbouring order statistics is done by the
 * caller exactly as NumPy does (host arithmetic on two scalars). */
int rsseg_order_stats_f32(rsseg_ctx *ctx, const float *d_x, int64_t n_local, const int64_t *ranks,
                          int nranks, float *out_values, int64_t *n_nan_out);
/* The same for up to 8 planes of equal length at once (ranks: [nplanes][nranks], out_values: [nplanes][nranks],
 * n_nan_out: [nplanes]): the planes advance pass by pass together, so a group costs three host synchronisations
 * and three all-reduces instead of three per plane.  With world > 1 the communication buffer must hold
 * nplanes * 262 208 bytes. */
int rsseg_order_stats_multi_f32(rsseg_ctx *ctx, const float *const *d_planes, int nplanes, int64_t n_local,
                                const int64_t *ranks, int nranks, float *out_values, int64_t *n_nan_out);

/* The same for 8-bit planes (one byte per pixel; the TM tiles the reference reads are uint8 digital numbers,
 * modules/features/preprocessing.py:117-118, which scripts/2_feature_extraction.py:156 widens with .astype(np.float32)):
 * one 256-bin pass, the values come back as float32 like np.percentile of the widened band would give them. */
int rsseg_order_stats_multi_u8(rsseg_ctx *ctx, const uint8_t *const *d_planes, int nplanes, int64_t n_local,
                               const int64_t *ranks, int nranks, float *out_values, int64_t *n_nan_out);

/* ---- K2: percentile normalisation + spectral indices ------------------------------------ */
/* robust_normalize (indices.py:25-48), elementwise part: clip to [lo,hi], (x-lo)/(hi-lo+1e-10)
 * in float32.  d_out may alias d_x. */
int rsseg_normalize_f32(rsseg_ctx *ctx, const float *d_x, int64_t n, float lo, float hi, float *d_out);

/* Fused robust_normalize of the five bands the indices read + calculate_ndvi / evi / msavi /
 * ndwi / mndwi / ndbi / bsi (indices.py:50-203; call order scripts/2_feature_extraction.py:63-73).
 * d_bands[5] = raw blue, green, red, nir, swir1 planes; lohi[10] = (lo,hi) per band.
 * d_out[7] = ndvi, evi, msavi, ndwi, mndwi, ndbi, bsi planes (any may be NULL = not wanted).
 * d_norm[5] (optional, entries may be NULL) receive the normalised bands.
 * If lohi == NULL the bands are taken as already normalised. */
int rsseg_spectral_indices_f32(rsseg_ctx *ctx, const float *const *d_bands, int64_t n, const float *lohi,
                               float *const *d_out, float *const *d_norm);
/* The same with calculate_evi's coefficients evi_coef[4] = {L, C1, C2, G} (indices.py:73; NULL = the defaults 1, 6, 7.5, 2.5). */
int rsseg_spectral_indices_evi_f32(rsseg_ctx *ctx, const float *const *d_bands, int64_t n, const float *lohi,
                                   float *const *d_out, float *const *d_norm, const float *evi_coef);

/* The same on 8-bit band planes: bit for bit the result of the float32 entry point on the widened planes (the
 * normalisation of a byte is a 256-entry table filled with the float path's own operations); 5 B/px read instead of 20. */
int rsseg_spectral_indices_evi_u8(rsseg_ctx *ctx, const uint8_t *const *d_bands, int64_t n, const float *lohi,
                                  float *const *d_out, float *const *d_norm, const float *evi_coef);

/* ---- K3: PCA ------------------------------------------------------------------------------ */
/* perform_pca (indices.py:205-246): RobustScaler transform x' = (float)((double)(x - center) /
 * scale) applied on the fly, then mean / Gram accumulation (exact), covariance, symmetric
 * eigen-decomposition, sign convention of sklearn's svd_flip, projection to n_components planes.
 * d_bands[nb]: normalised band planes; center[nb] (float32 medians) and scale[nb] (float64 IQRs)
 * come from K1 + host interpolation; pass center = NULL for no scaling.
 * Outputs (host): components (n_components x nb, row-major, float32), explained_variance_ratio
 * (n_components), mean (nb), explained_variance (n_components). d_out[n_components]: projected planes. */
int rsseg_pca_fit_transform_f32(rsseg_ctx *ctx, const float *const *d_bands, int nb, int64_t n_local,
                                const float *center, const double *scale, int n_components,
                                float *const *d_out, float *components, float *explained_variance_ratio,
                                float *mean, float *explained_variance);
/* The same on RAW bands: each value first goes through robust_normalize with lohi[2*b], lohi[2*b+1] (np.percentile 2 /
 * 98 of band b) — the arithmetic of rsseg_normalize_f32 — so the normalised planes need not exist in memory. */
int rsseg_pca_fit_transform_raw_f32(rsseg_ctx *ctx, const float *const *d_bands, int nb, int64_t n_local, const float *lohi,
                                    const float *center, const double *scale, int n_components, float *const *d_out,
                                    float *components, float *explained_variance_ratio, float *mean,
                                    float *explained_variance);

/* The same with the FIT restricted to the pixels [fit_off, fit_off + fit_n) of the planes while all n_local pixels are
 * projected: a rank of a row-sharded raster passes its stripe plus halo rows (the 7x7 context mean of the first
 * component needs 3 rows either side, indices.py:770) and fits on the rows it owns.  lohi may be NULL (bands already
 * normalised), center / scale may be NULL (no RobustScaler). */
int rsseg_pca_fit_transform_ext_f32(rsseg_ctx *ctx, const float *const *d_bands, int nb, int64_t n_local, int64_t fit_off,
                                    int64_t fit_n, const float *lohi, const float *center, const double *scale, int n_components,
                                    float *const *d_out, float *components, float *explained_variance_ratio, float *mean,
                                    float *explained_variance);
/* rsseg_pca_fit_transform_ext_f32 on 8-bit band planes (bit for bit the float32 result on the widened planes). */
int rsseg_pca_fit_transform_ext_u8(rsseg_ctx *ctx, const uint8_t *const *d_bands, int nb, int64_t n_local, int64_t fit_off,
                                   int64_t fit_n, const float *lohi, const float *center, const double *scale, int n_components,
                                   float *const *d_out, float *components, float *explained_variance_ratio, float *mean,
                                   float *explained_variance);
/* The spectral indices AND the PCA of the same raw bands in two passes instead of three: the fit (Gram pass) as in
 * rsseg_pca_fit_transform_ext_*, then ONE pass that writes the seven index planes d_idx[7] (ndvi, evi, msavi, ndwi, mndwi,
 * ndbi, bsi; entries may be NULL), optionally the normalised bands d_norm[5] (may be NULL, entries may be NULL) and the
 * n_components projected planes d_pc — the values rsseg_spectral_indices_evi_* and rsseg_pca_fit_transform_ext_* return,
 * bit for bit (calculate_* of indices.py:50-203 and perform_pca of :205-246 on the same robust-normalised bands, as
 * scripts/2_feature_extraction.py:63-78 calls them).  d_bands[0..4] = blue, green, red, nir, swir1; lohi[2 * nb] is required.
 * d_q (optional): the texture chain's input in the same pass — rsseg_normalize_quantize_u8 of the normalised NIR band
 * (band 3) with the percentiles q_lo, q_hi of THAT normalised band and the multiplier q_mult (levels - 1 or 255):
 * calculate_glcm_features re-normalises the band it receives and truncates it to uint8 (indices.py:265-268).
 * With rsseg_ctx_collect_minmax on, rsseg_ctx_last_minmax(0..6) are the indices' extrema, (7 + c) the components'. */
int rsseg_indices_pca_f32(rsseg_ctx *ctx, const float *const *d_bands, int nb, int64_t n_local, int64_t fit_off, int64_t fit_n,
                          const float *lohi, const float *center, const double *scale, int n_components, const float *evi_coef,
                          float *const *d_idx, float *const *d_norm, float *const *d_pc, uint8_t *d_q, float q_lo, float q_hi,
                          float q_mult, float *components, float *explained_variance_ratio, float *mean, float *explained_variance);
int rsseg_indices_pca_u8(rsseg_ctx *ctx, const uint8_t *const *d_bands, int nb, int64_t n_local, int64_t fit_off, int64_t fit_n,
                         const float *lohi, const float *center, const double *scale, int n_components, const float *evi_coef,
                         float *const *d_idx, float *const *d_norm, float *const *d_pc, uint8_t *d_q, float q_lo, float q_hi,
                         float q_mult, float *components, float *explained_variance_ratio, float *mean, float *explained_variance);
/* Errors of the PCA entry points: RSSEG_ERR_INVALID "Input X contains NaN." / "... infinity" (sklearn's PCA raises
 * ValueError on such input, sklearn/utils/validation.py); bands that are not robust-normalised are range-checked
 * with one extra pass so that the exact fixed-point accumulation fits any finite input (raw DN, reflectances). */

/* ---- K4/K5: GLCM texture + bilinear upsample ------------------------------------------- */
/* calculate_glcm_features (indices.py:248-318), window loop: d_q is the quantised uint8 plane
 * ((band*(levels-1)).astype(uint8), :268), H x W.  Writes the five property maps
 * (contrast, dissimilarity, homogeneity, energy, correlation), each ((H-win)/step+1) x
 * ((W-win)/step+1) float32, mean over the 4 angles (0, 45, 90, 135 degrees, distance 1),
 * symmetric + normalised co-occurrence. Any of d_props[i] may be NULL. */
int rsseg_glcm_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int levels, int win, int step,
                  float *const *d_props);
/* float32 plane in [0,1] -> uint8 by truncation of x*mult (indices.py:268, 415, 458). */
int rsseg_quantize_u8(rsseg_ctx *ctx, const float *d_x, int64_t n, float mult, uint8_t *d_q);
/* robust_normalize(x) with the given percentiles, then the truncation of rsseg_quantize_u8, in one pass (the texture
 * functions re-normalise the band they receive before quantising it, indices.py:265-268, 412-415). */
int rsseg_normalize_quantize_u8(rsseg_ctx *ctx, const float *d_x, int64_t n, float lo, float hi, float mult, uint8_t *d_q);
/* uint8 plane -> float32 of (uint8 / 255.0 in float64): the value sklearn sees for the morphological members
 * (features[...] = gradient / 255.0, indices.py:436-440; float32 cast sklearn/ensemble/_forest.py:640). */
int rsseg_u8_to_unit_f32(rsseg_ctx *ctx, const uint8_t *d_q, int64_t n, float *d_out);
/* uint8 plane -> float32 plane, exact: band.astype(np.float32) of an 8-bit raster (scripts/2_feature_extraction.py:156), for the
 * entry points that have no 8-bit form. */
int rsseg_u8_to_f32(rsseg_ctx *ctx, const uint8_t *d_q, int64_t n, float *d_out);
/* cv2.resize(src, (dw, dh), interpolation=INTER_LINEAR) for float32 (indices.py:308). */
int rsseg_resize_bilinear_f32(rsseg_ctx *ctx, const float *d_src, int sh, int sw, float *d_dst, int dh, int dw);
/* Row-striped form for sharded rasters: d_src holds rows [src_row0, src_row0 + sh_local) of a source sh rows tall,
 * d_dst receives rows [dst_row0, dst_row0 + dh_local) of the dh x dw result; identical values to the un-sharded call. */
int rsseg_resize_bilinear_rows_f32(rsseg_ctx *ctx, const float *d_src, int sh_local, int sw, int src_row0, int sh,
                                   float *d_dst, int dh_local, int dw, int dst_row0, int dh);

/* The same for nplanes (<= 8) source maps of one shape in ONE launch (the five GLCM property maps, indices.py:307-310);
 * with rsseg_ctx_collect_minmax on, rsseg_ctx_last_minmax(plane) returns the extrema of each. */
int rsseg_resize_bilinear_rows_multi_f32(rsseg_ctx *ctx, const float *const *d_src, int nplanes, int sh_local, int sw, int src_row0, int sh,
                                         float *const *d_dst, int dh_local, int dw, int dst_row0, int dh);

/* ---- K6/K7/K8: window operators ---------------------------------------------------------- */
/* Every operator exists in a ROWS form for row-sharded rasters: the plane holds Hin rows, rows
 * [y0, y1) are produced into a compact (y1 - y0) x W output.  edges bit 0 / bit 1 say that row 0 / row Hin - 1 is the
 * image's first / last row, where the operator's border rule applies; a side that is not an image edge must carry
 * R = k/2 halo rows (2R for opening / closing), otherwise RSSEG_ERR_INVALID.  The plain forms are rows (0, H), edges 3.
 * The stripes of a sharded raster get exactly the rows of the un-sharded result. */
/* cv2.boxFilter(normalize=True) / cv2.blur for float32 (indices.py:770-771, 537, 541): k x k mean, k in {3,5,7,9},
 * float64 sums in a fixed order (row sums left to right, then top to bottom), border RSSEG_BORDER_*.  If square != 0
 * the input is squared (float32) first (blur(band*band), indices.py:541). */
int rsseg_box_mean_f32(rsseg_ctx *ctx, const float *d_x, int H, int W, int k, int border, int square,
                       float *d_out);
/* The same for nplanes (<= 8) planes of equal shape in ONE launch (add_spatial_context loops over 7 channels). */
int rsseg_box_mean_rows_f32(rsseg_ctx *ctx, const float *const *d_x, int nplanes, int Hin, int W, int y0, int y1, int edges,
                            int k, int border, int square, float *const *d_out);
/* std_dev_scale_k (indices.py:537-548): sqrt(max(blur(x*x) - blur(x)^2, 0)), REFLECT_101. */
int rsseg_local_std_f32(rsseg_ctx *ctx, const float *d_x, int H, int W, int k, float *d_out);
/* variance_scale_k (indices.py:541-544): max(blur(x*x) - blur(x)^2, 0), REFLECT_101 (rsseg_local_std_f32 without the sqrt). */
int rsseg_local_var_f32(rsseg_ctx *ctx, const float *d_x, int H, int W, int k, float *d_out);
/* rows form of both: variance != 0 selects variance_scale_k */
int rsseg_local_std_rows_f32(rsseg_ctx *ctx, const float *d_x, int Hin, int W, int y0, int y1, int edges, int k, int variance,
                             float *d_out);
/* cv2.morphologyEx(MORPH_GRADIENT, ones(k,k)) on uint8 (indices.py:422, 433); output uint8. */
int rsseg_morph_gradient_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int k, uint8_t *d_out);
/* cv2.erode / cv2.dilate / cv2.morphologyEx(MORPH_OPEN | MORPH_CLOSE | MORPH_GRADIENT) with ones(k,k) on uint8
 * (calculate_morphological_features, indices.py:421-433): k in {3,5,7}, default border (out-of-image taps never win),
 * opening = dilate(erode(x)), closing = erode(dilate(x)).  Output uint8 (the caller divides by 255.0, :436-440). */
#define RSSEG_MORPH_ERODE 0
#define RSSEG_MORPH_DILATE 1
#define RSSEG_MORPH_OPEN 2
#define RSSEG_MORPH_CLOSE 3
#define RSSEG_MORPH_GRADIENT 4
int rsseg_morph_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int k, int op, uint8_t *d_out);
int rsseg_morph_rows_u8(rsseg_ctx *ctx, const uint8_t *d_q, int Hin, int W, int y0, int y1, int edges, int k, int op,
                        uint8_t *d_out);
/* 'laplacian' of calculate_filter_responses (indices.py:472-474): cv2.Laplacian(u8, CV_32F) (3x3 cross, REFLECT_101)
 * / 255.0, then (l - min) / (max - min + 1e-10) in float32.  Two passes over the uint8 plane (extrema, then the
 * normalised write); the global min / max of the produced rows go through the all-reduce hook. */
int rsseg_laplacian_norm_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, float *d_out);
int rsseg_laplacian_norm_rows_u8(rsseg_ctx *ctx, const uint8_t *d_q, int Hin, int W, int y0, int y1, int edges, float *d_out);
/* sobel_mag (indices.py:477-480): 3x3 Sobel x/y of the uint8 plane as float32 / 255, magnitude,
 * divided by (global max + 1e-10).  Two passes as above; the global max goes through the all-reduce hook. */
int rsseg_sobel_mag_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, float *d_out);
int rsseg_sobel_mag_rows_u8(rsseg_ctx *ctx, const uint8_t *d_q, int Hin, int W, int y0, int y1, int edges, float *d_out);

/* ---- K9/K10: KMeans ------------------------------------------------------------------------ */
typedef struct rsseg_kmeans_info {
    int32_t n_iter;
    int32_t relocated;            /* empty-cluster relocations performed */
    double tol;                   /* tol * mean(var) actually used */
    double scale[RSSEG_MAX_FEATURES];
    double min[RSSEG_MAX_FEATURES];
    double mean[RSSEG_MAX_FEATURES];
    int64_t init_indices[RSSEG_MAX_CLUSTERS]; /* global pixel index of each k-means++ seed */
    double ms_init;               /* wall milliseconds: scaler + k-means++ */
    double ms_lloyd;              /* wall milliseconds: Lloyd iterations */
} rsseg_kmeans_info;

/* unsupervised_kmeans_classification (modules/features/extract.py:568-579): NaN->0, MinMaxScaler,
 * KMeans(n_clusters, random_state=seed, n_init=1, init='k-means++', max_iter, tol).fit_predict.
 * d_planes[F]: feature planes, all float32 (dtype RSSEG_F32) or all float64 (RSSEG_F64), n_local
 * pixels each (this rank's stripe; ranks hold consecutive stripes in rank order).
 * d_labels: int32[n_local], cluster ids 0..k-1.  centers (host, k*F doubles, scaled-and-centred
 * space + mean added back, like cluster_centers_). */
int rsseg_kmeans_fit_predict(rsseg_ctx *ctx, const void *const *d_planes, int F, int dtype, int64_t n_local,
                             int k, uint32_t seed, int max_iter, double tol, int32_t *d_labels,
                             double *centers, rsseg_kmeans_info *info);
/* The same when the caller already knows this rank's minimum and maximum of every plane (NaN counted as 0), e.g. from
 * rsseg_ctx_last_minmax: the MinMaxScaler pass over the planes is skipped (the extrema are still all-reduced). */
int rsseg_kmeans_fit_predict_mm(rsseg_ctx *ctx, const void *const *d_planes, int F, int dtype, int64_t n_local, int k,
                                uint32_t seed, int max_iter, double tol, int32_t *d_labels, double *centers,
                                rsseg_kmeans_info *info, const double *local_min, const double *local_max);

/* ---- K11: random-forest inference ---------------------------------------------------------- */
/* Flattened sklearn forest (tree_ arrays concatenated; children are tree-local, -1 = leaf).
 * value: (n_nodes_total x n_classes) float64 class fractions.  The forest is copied to the device
 * and stays loaded in the context until the next load or destroy. */
int rsseg_forest_load(rsseg_ctx *ctx, int n_trees, const int64_t *tree_off, const int32_t *left,
                      const int32_t *right, const int32_t *feature, const double *threshold,
                      const uint8_t *missing_go_left, const double *value, int n_classes,
                      const int64_t *classes, int n_features);
/* predict_image (modules/supervised_classifiers.py:99-115) / supervised_classification_predict
 * (modules/features/extract.py:690-719): d_planes[F] float32 feature planes -> int64 class per pixel. */
int rsseg_forest_predict(rsseg_ctx *ctx, const float *const *d_planes, int F, int64_t n, int64_t *d_out);

/* ---- K12: rule-based classification --------------------------------------- */
/* threshold_segmentation (modules/features/extract.py:344-404, otsu=False): NaN counts as 0, then d_out = 1 where
 * lo < x < hi, else 0 (pass -INFINITY / INFINITY for `x > t` / `x < t`). */
int rsseg_threshold_band_f32(rsseg_ctx *ctx, const float *d_x, int64_t n, float lo, float hi, uint8_t *d_out);
/* The band conditions of extract_bareland_by_rule (extract.py:486-497: np.logical_and(x > lo, x < hi) on the raw plane):
 * nan_as_zero = 0 leaves a NaN pixel outside every interval (both comparisons are false); nan_as_zero = 1 is
 * rsseg_threshold_band_f32. */
int rsseg_band_interval_f32(rsseg_ctx *ctx, const float *d_x, int64_t n, float lo, float hi, int nan_as_zero, uint8_t *d_out);
/* The same on a float64 plane (the comparisons of a float64 feature with a Python-float threshold are float64 comparisons). */
int rsseg_band_interval_f64(rsseg_ctx *ctx, const double *d_x, int64_t n, double lo, double hi, int nan_as_zero, uint8_t *d_out);
/* threshold_segmentation(..., otsu=True) (extract.py:358-371): NaN -> 0; *vmin / *vmax = the plane's extrema; when they are
 * equal d_out is all 0 (above) / all 1 and *level = -1; otherwise the plane is stretched to uint8 in its own dtype
 * (np.clip((x - min) / (max - min + 1e-10) * 255, 0, 255).astype(uint8)), *level = cv2.threshold(THRESH_OTSU)'s level of its
 * 256-bin histogram (OpenCV's getThreshVal_Otsu_8u restated), and d_out = q > level (above) or its complement.
 * dtype: RSSEG_F32 or RSSEG_F64.  Whole raster on one GPU. */
int rsseg_otsu_mask(rsseg_ctx *ctx, const void *d_x, int dtype, int64_t n, int above, uint8_t *d_out, int *level, double *vmin,
                    double *vmax);
/* Mask algebra of extract_builtup_by_threshold / extract_bareland_by_rule (extract.py:447-505) on 0 / 1 planes:
 * op 0: a & b, 1: a | b, 2: a & !b, 3: !a (d_b may be NULL).  d_out may alias an input. */
int rsseg_mask_op_u8(rsseg_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, int64_t n, int op, uint8_t *d_out);
/* final_map[mask == 1] = value, or only where final_map == 0 (scripts/3_classification.py:361-363, 373). */
int rsseg_mask_paint_u8(rsseg_ctx *ctx, uint8_t *d_map, const uint8_t *d_mask, int64_t n, int value, int only_unset);
/* cv2.morphologyEx / erode / dilate with cv2.getStructuringElement(MORPH_ELLIPSE, (k, k)), k odd in 3 ... 31, on a uint8
 * plane (advanced_post_processing, extract.py:311-313, 334-336): op = RSSEG_MORPH_ERODE / DILATE / OPEN / CLOSE. */
int rsseg_morph_ellipse_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int k, int op, uint8_t *d_out);
/* scipy.ndimage.label(mask, structure=ones((3,3))) + the np.bincount area filter (extract.py:319-329): d_out = mask
 * without its 8-connected components of fewer than min_area pixels.  Whole raster on one GPU (components cross
 * stripes, so this operator is not row-sharded). */
int rsseg_remove_small_components_u8(rsseg_ctx *ctx, const uint8_t *d_mask, int H, int W, int min_area, uint8_t *d_out);
/* scipy.ndimage.binary_fill_holes(mask) (advanced_post_processing's branch for an even or zero smooth_kernel_size,
 * extract.py:314-316): d_out = mask plus every 4-connected background component that touches no image border.
 * Whole raster on one GPU. */
int rsseg_fill_holes_u8(rsseg_ctx *ctx, const uint8_t *d_mask, int H, int W, uint8_t *d_out);

/* ---- K13: remaining texture members of the feature dictionary ---------------- */
/* calculate_lbp_features (indices.py:320-344): skimage.feature.local_binary_pattern(u8, n_points, radius, 'uniform');
 * d_out: the codes 0 .. n_points + 1 as uint8 (the caller divides by their maximum in float64, :342). */
int rsseg_lbp_uniform_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int n_points, double radius, uint8_t *d_out);
/* entropy_scale_k (indices.py:551-560): skimage.filters.rank.entropy(u8, disk(radius)), float64, radius in {1,2,3,5,7}
 * (the caller divides by the maximum, :559). */
int rsseg_rank_entropy_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int radius, double *d_out);
/* gaussian_5 / gaussian_15 (indices.py:463-464): cv2.GaussianBlur(u8, (ksize, ksize), 0), OpenCV's fixed-point path for
 * 8-bit images, BORDER_REFLECT_101; uint8 out (the caller divides by 255.0 and forms the DoG, :463-470). */
int rsseg_gaussian_blur_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int ksize, uint8_t *d_out);
/* host-only: the ksize taps of that blur in 8 fractional bits (they sum to 256). */
int rsseg_host_gaussian_kernel_fixed(int ksize, int *taps);

/* ---- host-only helper (no GPU needed) ------------------------------------------------------- */
/* The random draws of k-means++ as the library makes them (numpy RandomState(seed): MT19937, random_sample,
 * RandomState.choice over n equal weights — sklearn/cluster/_kmeans.py:213-270): *center_id = index of the first
 * centre, uniforms[(k-1) * (2 + floor(ln k))] = the uniform(size=L) draws of the later rounds.  Used by the CPU test
 * suite to pin the generator against NumPy. */
int rsseg_host_kmeans_draws(uint32_t seed, int64_t n, int dtype, int k, int64_t *center_id, double *uniforms);
/* TIFF LZW (TIFF 6.0 section 13, as libtiff / GDAL write it: MSB-first codes of 9..12 bits, early change) for the GeoTIFF
 * outputs the reference writes with rasterio compress='lzw' (scripts/2_feature_extraction.py:239-258,
 * scripts/3_classification.py:509-538); called per strip / tile by rsseg/tiff.py.  Return the number of bytes produced,
 * or a negative value (-2: output buffer too small, -3: corrupt stream). */
int64_t rsseg_host_lzw_encode(const uint8_t *in, int64_t n, uint8_t *out, int64_t cap);
int64_t rsseg_host_lzw_decode(const uint8_t *in, int64_t n, uint8_t *out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* RSSEG_H */
