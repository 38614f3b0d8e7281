// K12 — rule-based classification: index thresholds, elliptical binary morphology and the
// 8-connected component area filter of advanced_post_processing.
//
// Replaces (reference modules/features/extract.py):
//   threshold_segmentation          :344-404   NaN -> 0, then x > t / x < t; otsu=True (:358-371): min-max stretch to uint8,
//                                              256-bin histogram, OpenCV's between-class-variance scan, q > t
//   advanced_post_processing        :299-341   cv2.morphologyEx(CLOSE, ELLIPSE k) -> scipy.ndimage.label(structure=ones(3,3))
//                                              + np.bincount area filter -> cv2.morphologyEx(OPEN, ELLIPSE k); even / zero
//                                              kernel sizes: scipy.ndimage.binary_fill_holes instead of the closing (:315)
//   extract_*_by_threshold / _rule  :406-505   mask algebra (scripts/3_classification.py:335-375 merges them by priority)
// Masks are uint8 planes holding 0 / 1.  cv2.getStructuringElement(MORPH_ELLIPSE): (3,3) is the cross, (5,5) the 5x5
// square without its four corner pairs (rows 01110 are 00100: see rf_se below); cv2's default morphology border never
// wins (erode: outside = max, dilate: outside = min), i.e. out-of-image taps are skipped.
// Connected components: lock-free union-find over the image (Playne & Hawick's atomicMin union on a parent plane):
// one pass unites every foreground pixel with its W / NW / N / NE foreground neighbours (8-connectivity needs only
// these four), a second pass flattens, a third counts the pixels of every root, a fourth keeps components of at least
// min_area pixels.  The result depends on connectivity and areas only, so it equals scipy's label + bincount filter
// whatever order the atomics land in.  binary_fill_holes is the same union-find on the BACKGROUND with 4-connectivity
// (scipy's default structure): a hole is a background component that owns no pixel of the image border.
#include "common.h"

#include <float.h>
#include <math.h>

#define K12_THREADS 256

static inline unsigned k12_grid(int64_t n) { return (unsigned)std::min<int64_t>(65535 * 16, std::max<int64_t>(1, ceil_div64(n, K12_THREADS))); }

// out = 1 where lo < x < hi; use -inf / +inf for one-sided thresholds.  nan_as_zero: threshold_segmentation's rule
// (NaN -> 0 before the comparison, extract.py:354-356); otherwise a NaN pixel fails both comparisons, as the plain NumPy
// comparisons of extract_bareland_by_rule do (extract.py:486-497)
template <typename T>
__global__ __launch_bounds__(K12_THREADS) void k12_band(const T *__restrict__ x, int64_t n, T lo, T hi, int nan_as_zero,
                                                        uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        T v = x[i];
        if (nan_as_zero && v != v) v = (T)0;
        out[i] = (v > lo && v < hi) ? 1 : 0;
    }
}

// ---- Otsu (extract.py:358-371) ----
// extrema of a plane with NaN counted as 0 (np.nan_to_num first, :354-356): one {min, max} pair per workgroup, the host folds them
template <typename T>
__global__ __launch_bounds__(K12_THREADS) void k12_minmax(const T *__restrict__ x, int64_t n, double *__restrict__ part)
{
    __shared__ double s_mn[K12_THREADS / WAVE], s_mx[K12_THREADS / WAVE];
    double mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        T v = x[i];
        if (v != v) v = (T)0;
        mn = fmin(mn, (double)v);
        mx = fmax(mx, (double)v);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane_id() == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < K12_THREADS / WAVE; w++) { mn = fmin(mn, s_mn[w]); mx = fmax(mx, s_mx[w]); }
        part[2 * blockIdx.x] = mn;
        part[2 * blockIdx.x + 1] = mx;
    }
}
// np.clip((x - min) / (max - min + 1e-10) * 255, 0, 255).astype(np.uint8) in the plane's own dtype (one IEEE operation per
// NumPy operation; `den` is the host's fl(fl(max - min) + 1e-10) in that dtype), written to q, and its 256-bin histogram
template <typename T>
__global__ __launch_bounds__(K12_THREADS) void k12_otsu_quantize(const T *__restrict__ x, int64_t n, T mn, T den, uint8_t *__restrict__ q,
                                                                 unsigned long long *__restrict__ hist)
{
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;     // K12_THREADS == 256
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        T v = x[i];
        if (v != v) v = (T)0;
        T s = (v - mn) / den;
        s = s * (T)255;
        s = s < (T)0 ? (T)0 : s;
        s = s > (T)255 ? (T)255 : s;
        const int b = (int)s;
        q[i] = (uint8_t)b;
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
// cv2.threshold(THRESH_BINARY): q > t -> 1 (above) or its complement; in place on the quantised plane
__global__ __launch_bounds__(K12_THREADS) void k12_otsu_apply(uint8_t *__restrict__ q, int64_t n, int t, int above)
{
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        const int m = q[i] > t;
        q[i] = (uint8_t)(above ? m : 1 - m);
    }
}

// op 0: a & b, 1: a | b, 2: a & !b, 3: !a
__global__ __launch_bounds__(K12_THREADS) void k12_maskop(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, int64_t n, int op,
                                                          uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        const bool x = a[i] != 0, y = b ? b[i] != 0 : false;
        out[i] = (op == 0 ? (x && y) : op == 1 ? (x || y) : op == 2 ? (x && !y) : !x) ? 1 : 0;
    }
}

// map[mask == 1 (and, if only_unset, map == 0)] = value   (scripts/3:361-363, 373)
__global__ __launch_bounds__(K12_THREADS) void k12_paint(uint8_t *__restrict__ map, const uint8_t *__restrict__ mask, int64_t n, int value, int only_unset)
{
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS)
        if (mask[i] == 1 && (!only_unset || map[i] == 0)) map[i] = (uint8_t)value;
}

// binary erode (MODE 0) / dilate (MODE 1) with cv2's elliptical structuring element, K = 3 or 5
template <int K, int MODE>
__global__ __launch_bounds__(K12_THREADS) void k12_morph(const uint8_t *__restrict__ q, int H, int W, uint8_t *__restrict__ out)
{
    constexpr int R = K / 2;
    const int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        int acc = MODE == 0 ? 255 : 0;
#pragma unroll
        for (int dy = -R; dy <= R; dy++) {
            // ellipse rows: K = 3: 010 / 111 / 010; K = 5: 00100 / 11111 / 11111 / 11111 / 00100
            const int half = (dy == -R || dy == R) ? 0 : R;
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = -R; dx <= R; dx++) {
                if (dx < -half || dx > half) continue;
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const int v = q[(size_t)yy * W + xx];
                acc = MODE == 0 ? (v < acc ? v : acc) : (v > acc ? v : acc);
            }
        }
        out[i] = (uint8_t)acc;
    }
}

// any odd K up to 31: cv2.getStructuringElement(MORPH_ELLIPSE, (K, K)) row by row — r = K / 2, row dy spans
// |dx| <= cvRound(r * sqrt((r*r - dy*dy) / (r*r))); the half-widths come from the host
struct k12_se {
    int K;
    int half[31];
};
__global__ __launch_bounds__(K12_THREADS) void k12_morph_any(const uint8_t *__restrict__ q, int H, int W, uint8_t *__restrict__ out, k12_se se, int mode)
{
    const int R = se.K / 2;
    const int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        int acc = mode == 0 ? 255 : 0;
        for (int dy = -R; dy <= R; dy++) {
            const int yy = y + dy, half = se.half[dy + R];
            if (yy < 0 || yy >= H) continue;
            const int x0 = x - half < 0 ? 0 : x - half, x1 = x + half >= W ? W - 1 : x + half;
            for (int xx = x0; xx <= x1; xx++) {
                const int v = q[(size_t)yy * W + xx];
                acc = mode == 0 ? (v < acc ? v : acc) : (v > acc ? v : acc);
            }
        }
        out[i] = (uint8_t)acc;
    }
}

// ---- connected components ----
__device__ __forceinline__ int cc_find(const int *L, int i)
{
    int p = L[i];
    while (p != i) { i = p; p = L[i]; }
    return i;
}
__device__ __forceinline__ void cc_union(int *L, int a, int b)
{
    bool done;
    do {
        a = cc_find(L, a);
        b = cc_find(L, b);
        if (a < b) {
            const int old = atomicMin(&L[b], a);
            done = old == b;
            b = old;
        } else if (b < a) {
            const int old = atomicMin(&L[a], b);
            done = old == a;
            a = old;
        } else
            done = true;
    } while (!done);
}
__global__ __launch_bounds__(K12_THREADS) void k12_cc_init(const uint8_t *__restrict__ m, int64_t n, int *__restrict__ L, int *__restrict__ area)
{
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        L[i] = m[i] ? (int)i : -1;
        area[i] = 0;
    }
}
__global__ __launch_bounds__(K12_THREADS) void k12_cc_union(const uint8_t *__restrict__ m, int H, int W, int *__restrict__ L)
{
    const int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        if (!m[i]) continue;
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        if (x > 0 && m[i - 1]) cc_union(L, (int)i, (int)i - 1);
        if (y > 0) {
            if (m[i - W]) cc_union(L, (int)i, (int)(i - W));
            if (x > 0 && m[i - W - 1]) cc_union(L, (int)i, (int)(i - W - 1));
            if (x + 1 < W && m[i - W + 1]) cc_union(L, (int)i, (int)(i - W + 1));
        }
    }
}
__global__ __launch_bounds__(K12_THREADS) void k12_cc_count(int64_t n, int *__restrict__ L, int *__restrict__ area)
{
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        if (L[i] < 0) continue;
        const int r = cc_find(L, (int)i);
        L[i] = r;                      // roots never change after the union pass, so flattening here is safe
        atomicAdd(&area[r], 1);
    }
}
__global__ __launch_bounds__(K12_THREADS) void k12_cc_filter(int64_t n, const int *__restrict__ L, const int *__restrict__ area, int min_area,
                                                             uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        const int r = L[i];
        // a pixel flattened before its root received its last child still points INTO its component: walk up (read-only)
        out[i] = (r >= 0 && area[cc_find(L, r)] >= min_area) ? 1 : 0;
    }
}

// ---- binary_fill_holes: union-find over the background, 4-connectivity ----
__global__ __launch_bounds__(K12_THREADS) void k12_bg_init(const uint8_t *__restrict__ m, int64_t n, int *__restrict__ L, int *__restrict__ open_)
{
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        L[i] = m[i] ? -1 : (int)i;
        open_[i] = 0;
    }
}
__global__ __launch_bounds__(K12_THREADS) void k12_bg_union(const uint8_t *__restrict__ m, int H, int W, int *__restrict__ L)
{
    const int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS) {
        if (m[i]) continue;
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        if (x > 0 && !m[i - 1]) cc_union(L, (int)i, (int)i - 1);
        if (y > 0 && !m[i - W]) cc_union(L, (int)i, (int)(i - W));
    }
}
// every background pixel of the image border marks its component as open (connected to the outside)
__global__ __launch_bounds__(K12_THREADS) void k12_bg_border(const uint8_t *__restrict__ m, int H, int W, const int *__restrict__ L, int *__restrict__ open_)
{
    const int64_t per = 2 * ((int64_t)W + H);
    for (int64_t t = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; t < per; t += (int64_t)gridDim.x * K12_THREADS) {
        int y, x;
        if (t < W) { y = 0; x = (int)t; }
        else if (t < 2 * (int64_t)W) { y = H - 1; x = (int)(t - W); }
        else if (t < 2 * (int64_t)W + H) { y = (int)(t - 2 * (int64_t)W); x = 0; }
        else { y = (int)(t - 2 * (int64_t)W - H); x = W - 1; }
        const int64_t i = (int64_t)y * W + x;
        if (!m[i]) open_[cc_find(L, (int)i)] = 1;
    }
}
__global__ __launch_bounds__(K12_THREADS) void k12_bg_fill(const uint8_t *__restrict__ m, int64_t n, const int *__restrict__ L, const int *__restrict__ open_,
                                                           uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * K12_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K12_THREADS)
        out[i] = (m[i] || !open_[cc_find(L, (int)i)]) ? 1 : 0;
}

static int k12_check(rsseg_ctx *ctx, const char *what, const void *a, const void *out, int H, int W)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!a || !out || H < 1 || W < 1) return rs_fail(ctx, RSSEG_ERR_INVALID, "%s: bad arguments", what);
    if ((int64_t)H * W > 0x7fffffff) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "%s: more than 2^31 pixels", what);
    return RSSEG_OK;
}

extern "C" int rsseg_band_interval_f32(rsseg_ctx *ctx, const float *d_x, int64_t n, float lo, float hi, int nan_as_zero, uint8_t *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || !d_out || n < 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "threshold: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n) hipLaunchKernelGGL(k12_band<float>, dim3(k12_grid(n)), dim3(K12_THREADS), 0, ctx->stream, d_x, n, lo, hi, nan_as_zero, d_out);
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_band_interval_f64(rsseg_ctx *ctx, const double *d_x, int64_t n, double lo, double hi, int nan_as_zero, uint8_t *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || !d_out || n < 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "threshold: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n) hipLaunchKernelGGL(k12_band<double>, dim3(k12_grid(n)), dim3(K12_THREADS), 0, ctx->stream, d_x, n, lo, hi, nan_as_zero, d_out);
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

// cv2.threshold(..., THRESH_OTSU) on an 8-bit image (OpenCV imgproc/thresh.cpp, getThreshVal_Otsu_8u): the level that
// maximises the between-class variance q1 * q2 * (mu1 - mu2)^2, scanned upwards, first maximum wins; float64 throughout
static int otsu_level(const unsigned long long *h, int64_t n)
{
    double mu = 0.0;
    const double scale = 1.0 / (double)n;
    for (int i = 0; i < 256; i++) mu += (double)i * (double)h[i];
    mu *= scale;
    double mu1 = 0.0, q1 = 0.0, max_sigma = 0.0;
    int best = 0;
    for (int i = 0; i < 256; i++) {
        const double p_i = (double)h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1.0 - q1;
        if (std::min(q1, q2) < (double)FLT_EPSILON || std::max(q1, q2) > 1.0 - (double)FLT_EPSILON) continue;
        mu1 = (mu1 + (double)i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            best = i;
        }
    }
    return best;
}

template <typename T> static int otsu_run(rsseg_ctx *ctx, const T *d_x, int64_t n, int above, uint8_t *d_out, int *level, double *vmin, double *vmax)
{
    const unsigned g = (unsigned)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div64(n, K12_THREADS)));
    RSCHK(ws_reserve(ctx, sizeof(double) * 2 * 1024 + sizeof(unsigned long long) * 256));
    RSCHK(pin_reserve(ctx, sizeof(double) * 2 * 1024 + sizeof(unsigned long long) * 256));
    double *d_part = (double *)ctx->d_ws;
    unsigned long long *d_hist = (unsigned long long *)(d_part + 2 * 1024);
    double *h_part = (double *)ctx->h_pin;
    unsigned long long *h_hist = (unsigned long long *)(h_part + 2 * 1024);
    hipLaunchKernelGGL(k12_minmax<T>, dim3(g), dim3(K12_THREADS), 0, ctx->stream, d_x, n, d_part);
    HIPCHK(ctx, hipMemcpyAsync(h_part, d_part, sizeof(double) * 2 * g, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, rs_sync(ctx));
    double mn = h_part[0], mx = h_part[1];
    for (unsigned b = 1; b < g; b++) { mn = std::min(mn, h_part[2 * b]); mx = std::max(mx, h_part[2 * b + 1]); }
    *vmin = mn;
    *vmax = mx;
    if (mx == mn) {      // extract.py:361-363: no contrast -> all 0 (above) or all 1
        *level = -1;
        HIPCHK(ctx, hipMemsetAsync(d_out, above ? 0 : 1, (size_t)n, ctx->stream));
        return stream_sync(ctx);
    }
    volatile T d0 = (T)mx - (T)mn;          // the plane's dtype: NumPy keeps float32 for float32 scalars + a Python float
    volatile T den = d0 + (T)1e-10;
    HIPCHK(ctx, hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * 256, ctx->stream));
    hipLaunchKernelGGL(k12_otsu_quantize<T>, dim3(k12_grid(n)), dim3(K12_THREADS), 0, ctx->stream, d_x, n, (T)mn, (T)den, d_out, d_hist);
    HIPCHK(ctx, hipMemcpyAsync(h_hist, d_hist, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, rs_sync(ctx));
    *level = otsu_level(h_hist, n);
    hipLaunchKernelGGL(k12_otsu_apply, dim3(k12_grid(n)), dim3(K12_THREADS), 0, ctx->stream, d_out, n, *level, above);
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_otsu_mask(rsseg_ctx *ctx, const void *d_x, int dtype, int64_t n, int above, uint8_t *d_out, int *level, double *vmin,
                               double *vmax)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || !d_out || n < 1 || !level || !vmin || !vmax) return rs_fail(ctx, RSSEG_ERR_INVALID, "otsu_mask: bad arguments");
    if (dtype != RSSEG_F32 && dtype != RSSEG_F64) return rs_fail(ctx, RSSEG_ERR_INVALID, "otsu_mask: dtype must be RSSEG_F32 or RSSEG_F64");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return dtype == RSSEG_F32 ? otsu_run<float>(ctx, (const float *)d_x, n, above, d_out, level, vmin, vmax)
                              : otsu_run<double>(ctx, (const double *)d_x, n, above, d_out, level, vmin, vmax);
}

extern "C" int rsseg_threshold_band_f32(rsseg_ctx *ctx, const float *d_x, int64_t n, float lo, float hi, uint8_t *d_out)
{
    return rsseg_band_interval_f32(ctx, d_x, n, lo, hi, 1, d_out);
}

extern "C" int rsseg_mask_op_u8(rsseg_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, int64_t n, int op, uint8_t *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_a || !d_out || n < 0 || op < 0 || op > 3 || (op < 3 && !d_b)) return rs_fail(ctx, RSSEG_ERR_INVALID, "mask_op: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n) hipLaunchKernelGGL(k12_maskop, dim3(k12_grid(n)), dim3(K12_THREADS), 0, ctx->stream, d_a, d_b, n, op, d_out);
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_mask_paint_u8(rsseg_ctx *ctx, uint8_t *d_map, const uint8_t *d_mask, int64_t n, int value, int only_unset)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_map || !d_mask || n < 0 || value < 0 || value > 255) return rs_fail(ctx, RSSEG_ERR_INVALID, "mask_paint: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n) hipLaunchKernelGGL(k12_paint, dim3(k12_grid(n)), dim3(K12_THREADS), 0, ctx->stream, d_map, d_mask, n, value, only_unset);
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

template <int MODE> static int k12_morph_launch(rsseg_ctx *ctx, const uint8_t *in, int H, int W, int k, uint8_t *out)
{
    const unsigned g = k12_grid((int64_t)H * W);
    if (k == 3) hipLaunchKernelGGL((k12_morph<3, MODE>), dim3(g), dim3(K12_THREADS), 0, ctx->stream, in, H, W, out);
    else if (k == 5) hipLaunchKernelGGL((k12_morph<5, MODE>), dim3(g), dim3(K12_THREADS), 0, ctx->stream, in, H, W, out);
    else {
        k12_se se;
        se.K = k;
        const int r = k / 2;
        for (int i = 0; i < k; i++) {
            const int dy = i - r;
            se.half[i] = (int)nearbyint((double)r * sqrt((double)(r * r - dy * dy) * (1.0 / ((double)r * r))));   // cvRound: half to even
        }
        hipLaunchKernelGGL(k12_morph_any, dim3(g), dim3(K12_THREADS), 0, ctx->stream, in, H, W, out, se, MODE);
    }
    return RSSEG_OK;
}

extern "C" int rsseg_morph_ellipse_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int k, int op, uint8_t *d_out)
{
    RSCHK(k12_check(ctx, "morph_ellipse", d_q, d_out, H, W));
    if (k < 3 || k > 31 || k % 2 == 0) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "morph_ellipse: kernel size %d is not an odd number in 3 ... 31", k);
    if (d_q == d_out) return rs_fail(ctx, RSSEG_ERR_INVALID, "morph_ellipse: in-place not supported");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (op == RSSEG_MORPH_ERODE) RSCHK(k12_morph_launch<0>(ctx, d_q, H, W, k, d_out));
    else if (op == RSSEG_MORPH_DILATE) RSCHK(k12_morph_launch<1>(ctx, d_q, H, W, k, d_out));
    else if (op == RSSEG_MORPH_OPEN || op == RSSEG_MORPH_CLOSE) {
        RSCHK(ws_reserve(ctx, (size_t)H * W + 64));
        uint8_t *tmp = (uint8_t *)ctx->d_ws;
        if (op == RSSEG_MORPH_OPEN) {
            RSCHK(k12_morph_launch<0>(ctx, d_q, H, W, k, tmp));
            RSCHK(k12_morph_launch<1>(ctx, tmp, H, W, k, d_out));
        } else {
            RSCHK(k12_morph_launch<1>(ctx, d_q, H, W, k, tmp));
            RSCHK(k12_morph_launch<0>(ctx, tmp, H, W, k, d_out));
        }
    } else
        return rs_fail(ctx, RSSEG_ERR_INVALID, "morph_ellipse: unknown operation %d", op);
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_remove_small_components_u8(rsseg_ctx *ctx, const uint8_t *d_mask, int H, int W, int min_area, uint8_t *d_out)
{
    RSCHK(k12_check(ctx, "remove_small_components", d_mask, d_out, H, W));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t n = (int64_t)H * W;
    RSCHK(ws_reserve(ctx, sizeof(int) * 2 * (size_t)n + 256));
    int *L = (int *)ctx->d_ws, *area = L + n;
    const unsigned g = k12_grid(n);
    {
        prof_scope ps(ctx, "components");
        hipLaunchKernelGGL(k12_cc_init, dim3(g), dim3(K12_THREADS), 0, ctx->stream, d_mask, n, L, area);
        hipLaunchKernelGGL(k12_cc_union, dim3(g), dim3(K12_THREADS), 0, ctx->stream, d_mask, H, W, L);
        hipLaunchKernelGGL(k12_cc_count, dim3(g), dim3(K12_THREADS), 0, ctx->stream, n, L, area);
        hipLaunchKernelGGL(k12_cc_filter, dim3(g), dim3(K12_THREADS), 0, ctx->stream, n, (const int *)L, (const int *)area, min_area, d_out);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_fill_holes_u8(rsseg_ctx *ctx, const uint8_t *d_mask, int H, int W, uint8_t *d_out)
{
    RSCHK(k12_check(ctx, "fill_holes", d_mask, d_out, H, W));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t n = (int64_t)H * W;
    RSCHK(ws_reserve(ctx, sizeof(int) * 2 * (size_t)n + 256));
    int *L = (int *)ctx->d_ws, *open_ = L + n;
    const unsigned g = k12_grid(n);
    {
        prof_scope ps(ctx, "components");
        hipLaunchKernelGGL(k12_bg_init, dim3(g), dim3(K12_THREADS), 0, ctx->stream, d_mask, n, L, open_);
        hipLaunchKernelGGL(k12_bg_union, dim3(g), dim3(K12_THREADS), 0, ctx->stream, d_mask, H, W, L);
        hipLaunchKernelGGL(k12_bg_border, dim3(k12_grid(2 * ((int64_t)W + H))), dim3(K12_THREADS), 0, ctx->stream, d_mask, H, W, (const int *)L, open_);
        hipLaunchKernelGGL(k12_bg_fill, dim3(g), dim3(K12_THREADS), 0, ctx->stream, d_mask, n, (const int *)L, (const int *)open_, d_out);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}
