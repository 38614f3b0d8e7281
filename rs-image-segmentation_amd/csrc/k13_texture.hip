// K13 — the remaining texture members of the feature dictionary: uniform local binary patterns, rank-filter
// entropy over a disk, OpenCV's fixed-point Gaussian blur of uint8 images.  None of them reaches the 19-feature stack;
// they complete features_dict['lbp_feature'], ['multi_scale_features']['entropy_scale_k'] and
// ['filter_features']['gaussian_5' / 'gaussian_15' / 'dog'] (reference modules/features/indices.py:320-344, 551-560, 463-470).
//
// Library semantics restated (scikit-image and OpenCV are not installed, the reference pins neither: parity unpinned,
// checked against the NumPy restatement in oracle/ref_np.py):
//   skimage.feature.local_binary_pattern(u8, P, R, 'uniform')   image as float64; sample i at (r - R sin(2 pi i / P),
//        c + R cos(2 pi i / P)), offsets rounded to 5 decimals; bilinear interpolation top = (1-dc)*tl + dc*tr,
//        bottom likewise, (1-dr)*top + dr*bottom, pixels outside the image = 0; s_i = (sample - centre >= 0);
//        changes counted over the P - 1 neighbouring pairs (not circular); code = sum s_i if changes <= 2 else P + 1.
//   skimage.filters.rank.entropy(u8, disk(k))   histogram of the in-image pixels of the disk x^2 + y^2 <= k^2 around the
//        pixel; e = - sum over the bins in ascending order of p * log(p) / ln 2, p = count / population, float64.
//   cv2.GaussianBlur(u8, (k, k), 0)   fixed-point path: kernel in 8 fractional bits (k = 5: [16 64 96 64 16]; other
//        sizes: exp(-x^2 / 2 sigma^2), sigma = 0.3 ((k-1)/2 - 1) + 0.8, normalised, rounded with error diffusion so
//        that the taps sum to 256 — computed on the host); rows then columns in 8.8 / 16.16 fixed point,
//        (acc + 32768) >> 16; BORDER_REFLECT_101.
#include <cmath>

#include "common.h"

#define K13_THREADS 256

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ---- uniform LBP ----
struct lbp_args {
    double rp[32], cp[32];
    int P;
};

__device__ __forceinline__ double lbp_px(const uint8_t *__restrict__ q, int H, int W, long r, long c)
{
    return (r < 0 || r >= H || c < 0 || c >= W) ? 0.0 : (double)q[(size_t)r * W + c];
}

__global__ __launch_bounds__(K13_THREADS) void k13_lbp(const uint8_t *__restrict__ q, int H, int W, lbp_args a, uint8_t *__restrict__ out)
{
    const int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * K13_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K13_THREADS) {
        const int r = (int)(i / W), c = (int)(i - (int64_t)r * W);
        const double centre = (double)q[i];
        unsigned bits = 0;
        for (int p = 0; p < a.P; p++) {
            const double rr = (double)r + a.rp[p], cc = (double)c + a.cp[p];
            const long minr = (long)floor(rr), minc = (long)floor(cc), maxr = (long)ceil(rr), maxc = (long)ceil(cc);
            const double dr = rr - (double)minr, dc = cc - (double)minc;
            const double tl = lbp_px(q, H, W, minr, minc), tr = lbp_px(q, H, W, minr, maxc);
            const double bl = lbp_px(q, H, W, maxr, minc), br = lbp_px(q, H, W, maxr, maxc);
            const double top = (1.0 - dc) * tl + dc * tr;
            const double bottom = (1.0 - dc) * bl + dc * br;
            const double v = (1.0 - dr) * top + dr * bottom;
            if (v - centre >= 0.0) bits |= 1u << p;
        }
        const unsigned mask = a.P >= 32 ? 0xffffffffu : ((1u << a.P) - 1u);
        const unsigned pairs = (bits ^ (bits >> 1)) & (mask >> 1);  // s_i != s_{i+1} for i = 0 .. P-2
        const int changes = __popc(pairs);
        out[i] = (uint8_t)(changes <= 2 ? __popc(bits & mask) : a.P + 1);
    }
}

// ---- rank entropy over a disk ----
// One 256-bin histogram per THREAD in LDS as packed 8-bit counters (a disk of radius <= 7 has at most 149 pixels):
// hist[64 dwords][256 threads], bank = thread, every access conflict-free.
template <int RAD>
__global__ __launch_bounds__(K13_THREADS) void k13_entropy(const uint8_t *__restrict__ q, int H, int W, double *__restrict__ out)
{
    __shared__ unsigned hist[64 * K13_THREADS];
    const int64_t n = (int64_t)H * W;
    const int64_t per = (int64_t)gridDim.x * K13_THREADS, nround = (n + per - 1) / per;
    for (int64_t it = 0; it < nround; it++) {
        const int64_t i = (it * gridDim.x + blockIdx.x) * K13_THREADS + threadIdx.x;
        if (i >= n) continue;  // no barrier below: every thread owns its histogram
#pragma unroll 8
        for (int d = 0; d < 64; d++) hist[d * K13_THREADS + threadIdx.x] = 0;
        const int r = (int)(i / W), c = (int)(i - (int64_t)r * W);
        int pop = 0;
        for (int dy = -RAD; dy <= RAD; dy++) {
            const int yy = r + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -RAD; dx <= RAD; dx++) {
                if (dx * dx + dy * dy > RAD * RAD) continue;
                const int xx = c + dx;
                if (xx < 0 || xx >= W) continue;
                const unsigned v = q[(size_t)yy * W + xx];
                hist[(v >> 2) * K13_THREADS + threadIdx.x] += 1u << (8 * (v & 3));
                pop++;
            }
        }
        double e = 0.0;
        const double dpop = (double)pop;
        for (int d = 0; d < 64; d++) {
            const unsigned h = hist[d * K13_THREADS + threadIdx.x];
            if (!h) continue;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned cnt = (h >> (8 * k)) & 255u;
                if (cnt) {
                    const double p = (double)cnt / dpop;
                    e -= p * log(p) / 0.6931471805599453;
                }
            }
        }
        out[i] = e;
    }
}

// ---- fixed-point Gaussian ----
struct gk_args {
    unsigned short k[32];
    int n;
};
__global__ __launch_bounds__(K13_THREADS) void k13_gauss_h(const uint8_t *__restrict__ q, int H, int W, gk_args g, unsigned short *__restrict__ tmp)
{
    const int64_t n = (int64_t)H * W;
    const int R = g.n / 2;
    for (int64_t i = (int64_t)blockIdx.x * K13_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K13_THREADS) {
        const int r = (int)(i / W), c = (int)(i - (int64_t)r * W);
        unsigned acc = 0;
        for (int t = 0; t < g.n; t++) acc += (unsigned)q[(size_t)r * W + reflect101(c + t - R, W)] * g.k[t];
        tmp[i] = (unsigned short)acc;   // 8.8 fixed point: at most 255 * 256
    }
}
__global__ __launch_bounds__(K13_THREADS) void k13_gauss_v(const unsigned short *__restrict__ tmp, int H, int W, gk_args g, uint8_t *__restrict__ out)
{
    const int64_t n = (int64_t)H * W;
    const int R = g.n / 2;
    for (int64_t i = (int64_t)blockIdx.x * K13_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K13_THREADS) {
        const int r = (int)(i / W), c = (int)(i - (int64_t)r * W);
        unsigned acc = 0;
        for (int t = 0; t < g.n; t++) acc += (unsigned)tmp[(size_t)reflect101(r + t - R, H) * W + c] * g.k[t];
        const unsigned v = (acc + 32768u) >> 16;
        out[i] = (uint8_t)(v > 255u ? 255u : v);
    }
}

static unsigned k13_grid(int64_t n) { return (unsigned)std::min<int64_t>(65535 * 16, std::max<int64_t>(1, ceil_div64(n, K13_THREADS))); }

static int k13_check(rsseg_ctx *ctx, const char *what, const void *a, const void *out, int H, int W)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!a || !out || H < 1 || W < 1) return rs_fail(ctx, RSSEG_ERR_INVALID, "%s: bad arguments", what);
    return RSSEG_OK;
}

extern "C" int rsseg_lbp_uniform_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int n_points, double radius, uint8_t *d_out)
{
    RSCHK(k13_check(ctx, "lbp", d_q, d_out, H, W));
    if (n_points < 1 || n_points > 32 || !(radius > 0)) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "lbp: n_points in [1, 32], radius > 0");
    lbp_args a;
    memset(&a, 0, sizeof(a));
    a.P = n_points;
    for (int p = 0; p < n_points; p++) {
        // rr = -R sin(2 pi i / P), cc = R cos(2 pi i / P), np.round(., 5)
        volatile double ang = 2.0 * M_PI * (double)p;
        volatile double t = ang / (double)n_points;
        volatile double rr = -radius * std::sin(t), cc = radius * std::cos(t);
        a.rp[p] = std::nearbyint(rr * 1e5) / 1e5;
        a.cp[p] = std::nearbyint(cc * 1e5) / 1e5;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "lbp");
        hipLaunchKernelGGL(k13_lbp, dim3(k13_grid((int64_t)H * W)), dim3(K13_THREADS), 0, ctx->stream, d_q, H, W, a, d_out);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_rank_entropy_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int radius, double *d_out)
{
    RSCHK(k13_check(ctx, "rank_entropy", d_q, d_out, H, W));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const unsigned g = (unsigned)std::min<int64_t>(8192, std::max<int64_t>(1, ceil_div64((int64_t)H * W, K13_THREADS)));
    {
        prof_scope ps(ctx, "entropy");
        switch (radius) {
        case 1: hipLaunchKernelGGL(k13_entropy<1>, dim3(g), dim3(K13_THREADS), 0, ctx->stream, d_q, H, W, d_out); break;
        case 2: hipLaunchKernelGGL(k13_entropy<2>, dim3(g), dim3(K13_THREADS), 0, ctx->stream, d_q, H, W, d_out); break;
        case 3: hipLaunchKernelGGL(k13_entropy<3>, dim3(g), dim3(K13_THREADS), 0, ctx->stream, d_q, H, W, d_out); break;
        case 5: hipLaunchKernelGGL(k13_entropy<5>, dim3(g), dim3(K13_THREADS), 0, ctx->stream, d_q, H, W, d_out); break;
        case 7: hipLaunchKernelGGL(k13_entropy<7>, dim3(g), dim3(K13_THREADS), 0, ctx->stream, d_q, H, W, d_out); break;
        default: return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "rank_entropy: disk radius %d not in {1,2,3,5,7}", radius);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

// round half to even of a double (cvRound)
static long long cv_round(double v) { return (long long)std::nearbyint(v); }

extern "C" int rsseg_host_gaussian_kernel_fixed(int ksize, int *taps /* ksize */)
{
    if (ksize < 1 || ksize > 31 || !(ksize & 1) || !taps) return RSSEG_ERR_INVALID;
    static const double small[4][7] = {{1.0}, {0.25, 0.5, 0.25}, {0.0625, 0.25, 0.375, 0.25, 0.0625},
                                       {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125}};
    std::vector<double> k((size_t)ksize);
    if (ksize <= 7) {
        for (int i = 0; i < ksize; i++) k[i] = small[ksize >> 1][i];
    } else {
        const double sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8, scale2 = -0.5 / (sigma * sigma);
        double sum = 0.0;
        for (int i = 0; i < ksize; i++) {
            const double x = i - (ksize - 1) * 0.5;
            k[i] = std::exp(scale2 * x * x);
            sum += k[i];
        }
        const double inv = 1.0 / sum;
        for (int i = 0; i < ksize; i++) k[i] *= inv;
    }
    // error diffusion to 8 fractional bits; the centre tap takes what is left of 256
    double err = 0.0;
    long long sum = 0;
    const int h = ksize / 2;
    for (int i = 0; i < h; i++) {
        const double adj = k[i] * 256.0 + err;
        const long long v0 = cv_round(adj);
        err = adj - (double)v0;
        taps[i] = taps[ksize - 1 - i] = (int)v0;
        sum += v0;
    }
    taps[h] = (int)(256 - 2 * sum);
    return RSSEG_OK;
}

extern "C" int rsseg_gaussian_blur_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int ksize, uint8_t *d_out)
{
    RSCHK(k13_check(ctx, "gaussian_blur", d_q, d_out, H, W));
    gk_args g;
    memset(&g, 0, sizeof(g));
    int taps[32];
    if (rsseg_host_gaussian_kernel_fixed(ksize, taps) != RSSEG_OK) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "gaussian_blur: odd kernel size <= 31 expected, got %d", ksize);
    g.n = ksize;
    for (int i = 0; i < ksize; i++) g.k[i] = (unsigned short)taps[i];
    HIPCHK(ctx, hipSetDevice(ctx->device));
    RSCHK(ws_reserve(ctx, sizeof(unsigned short) * (size_t)H * W + 64));
    {
        prof_scope ps(ctx, "gauss");
        const unsigned gr = k13_grid((int64_t)H * W);
        hipLaunchKernelGGL(k13_gauss_h, dim3(gr), dim3(K13_THREADS), 0, ctx->stream, d_q, H, W, g, (unsigned short *)ctx->d_ws);
        hipLaunchKernelGGL(k13_gauss_v, dim3(gr), dim3(K13_THREADS), 0, ctx->stream, (const unsigned short *)ctx->d_ws, H, W, g, d_out);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}
