// K5-K8 — window operators of the level-2 features and of add_spatial_context:
//   bilinear upsample (cv2.resize INTER_LINEAR, indices.py:308), k x k box mean with float64 sums
//   (cv2.boxFilter / cv2.blur, indices.py:770-771, 537, 541), local standard deviation
//   (indices.py:537-548), 5x5 morphological gradient on uint8 (indices.py:422, 433) and 3x3 Sobel
//   magnitude / global max (indices.py:477-480).  OpenCV semantics as restated in oracle/ref_np.py.
// All are streaming stencils: one read + one write of the plane (4+4 B/px), HBM-bound; the k x k
// neighbourhood is served by L1/L2 (row-major 64x4 pixel workgroups).
#include <cmath>

#include "common.h"

__device__ __forceinline__ int border_idx(int i, int n, int mode)
{
    // mode 0: BORDER_REFLECT (edge duplicated), mode 1: BORDER_REFLECT_101
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = mode == 0 ? -i - 1 : -i;
        else i = mode == 0 ? 2 * n - 1 - i : 2 * n - 2 - i;
    }
    return i;
}

#define WG_X 64
#define WG_Y 4

// ---- box mean (optionally of x*x), float64 sums: rows left-to-right, then rows top-to-bottom ----
template <int K>
__global__ __launch_bounds__(256) void k6_box(const float *__restrict__ x, int H, int W, int mode, int square, float *__restrict__ out)
{
    const int px = blockIdx.x * WG_X + (threadIdx.x & 63), py = blockIdx.y * WG_Y + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    constexpr int R = K / 2;
    int xs[K];
#pragma unroll
    for (int d = 0; d < K; d++) xs[d] = border_idx(px + d - R, W, mode);
    double acc = 0.0;
#pragma unroll
    for (int dy = 0; dy < K; dy++) {
        const float *row = x + (size_t)border_idx(py + dy - R, H, mode) * W;
        double rs = 0.0;
#pragma unroll
        for (int d = 0; d < K; d++) {
            float v = row[xs[d]];
            if (square) v = v * v;
            rs = d == 0 ? (double)v : rs + (double)v;
        }
        acc = dy == 0 ? rs : acc + rs;
    }
    out[(size_t)py * W + px] = (float)(acc * (1.0 / (double)(K * K)));
}

template <int K, bool VAR>
__global__ __launch_bounds__(256) void k6_std(const float *__restrict__ x, int H, int W, float *__restrict__ out)
{
    const int px = blockIdx.x * WG_X + (threadIdx.x & 63), py = blockIdx.y * WG_Y + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    constexpr int R = K / 2;
    int xs[K];
#pragma unroll
    for (int d = 0; d < K; d++) xs[d] = border_idx(px + d - R, W, 1);
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int dy = 0; dy < K; dy++) {
        const float *row = x + (size_t)border_idx(py + dy - R, H, 1) * W;
        double r1 = 0.0, r2 = 0.0;
#pragma unroll
        for (int d = 0; d < K; d++) {
            const float v = row[xs[d]];
            const float vv = v * v;
            r1 = d == 0 ? (double)v : r1 + (double)v;
            r2 = d == 0 ? (double)vv : r2 + (double)vv;
        }
        a1 = dy == 0 ? r1 : a1 + r1;
        a2 = dy == 0 ? r2 : a2 + r2;
    }
    const double sc = 1.0 / (double)(K * K);
    const float mean = (float)(a1 * sc), mean_sq = (float)(a2 * sc);
    const float mm = mean * mean;
    float var = mean_sq - mm;
    if (var < 0.f) var = 0.f;
    out[(size_t)py * W + px] = VAR ? var : sqrtf(var);
}

template <int K>
__global__ __launch_bounds__(256) void k7_morph_grad(const uint8_t *__restrict__ q, int H, int W, uint8_t *__restrict__ out)
{
    const int px = blockIdx.x * WG_X + (threadIdx.x & 63), py = blockIdx.y * WG_Y + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    constexpr int R = K / 2;
    int mn = 255, mx = 0;
#pragma unroll
    for (int dy = -R; dy <= R; dy++) {
        const int yy = py + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -R; dx <= R; dx++) {
            const int xx = px + dx;
            if (xx < 0 || xx >= W) continue;
            const int v = q[(size_t)yy * W + xx];
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
    }
    out[(size_t)py * W + px] = (uint8_t)(mx - mn);
}

// erode (MODE 0: min) / dilate (MODE 1: max) with a K x K rectangle; taps outside the image never win
template <int K, int MODE>
__global__ __launch_bounds__(256) void k7_morph(const uint8_t *__restrict__ q, int H, int W, uint8_t *__restrict__ out)
{
    const int px = blockIdx.x * WG_X + (threadIdx.x & 63), py = blockIdx.y * WG_Y + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    constexpr int R = K / 2;
    int acc = MODE == 0 ? 255 : 0;
#pragma unroll
    for (int dy = -R; dy <= R; dy++) {
        const int yy = py + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -R; dx <= R; dx++) {
            const int xx = px + dx;
            if (xx < 0 || xx >= W) continue;
            const int v = q[(size_t)yy * W + xx];
            acc = MODE == 0 ? (v < acc ? v : acc) : (v > acc ? v : acc);
        }
    }
    out[(size_t)py * W + px] = (uint8_t)acc;
}

// cv2.Laplacian(u8, CV_32F), aperture 1: cross stencil, BORDER_REFLECT_101; value / 255 in float32.
// blockmin/blockmax[blk] = extrema over the block (for the min-max normalisation that follows)
__global__ __launch_bounds__(256) void k8_laplace(const uint8_t *__restrict__ q, int H, int W, float *__restrict__ out,
                                                  float *__restrict__ blockmin, float *__restrict__ blockmax)
{
    const int px = blockIdx.x * WG_X + (threadIdx.x & 63), py = blockIdx.y * WG_Y + (threadIdx.x >> 6);
    float l = 0.f;
    const bool live = px < W && py < H;
    if (live) {
        const int xm = border_idx(px - 1, W, 1), xp = border_idx(px + 1, W, 1);
        const int ym = border_idx(py - 1, H, 1), yp = border_idx(py + 1, H, 1);
        const int c = q[(size_t)py * W + px];
        const int s = (int)q[(size_t)ym * W + px] + (int)q[(size_t)yp * W + px] + (int)q[(size_t)py * W + xm] + (int)q[(size_t)py * W + xp] - 4 * c;
        l = (float)s / 255.0f;
        out[(size_t)py * W + px] = l;
    }
    float mn = wave_min(live ? l : INFINITY), mx = wave_max(live ? l : -INFINITY);
    __shared__ float smn[4], smx[4];
    if (lane_id() == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        blockmin[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        blockmax[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    }
}

__global__ __launch_bounds__(256) void k8_sub_div(float *__restrict__ x, int64_t n, float sub, float den)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = x[i] - sub;
        x[i] = d / den;
    }
}

// Sobel magnitude; blockmax[blk] = max over the block
__global__ __launch_bounds__(256) void k8_sobel(const uint8_t *__restrict__ q, int H, int W, float *__restrict__ out,
                                                float *__restrict__ blockmax)
{
    const int px = blockIdx.x * WG_X + (threadIdx.x & 63), py = blockIdx.y * WG_Y + (threadIdx.x >> 6);
    float mag = 0.f;
    if (px < W && py < H) {
        const int xm = border_idx(px - 1, W, 1), xp = border_idx(px + 1, W, 1);
        const int ym = border_idx(py - 1, H, 1), yp = border_idx(py + 1, H, 1);
        const uint8_t *r0 = q + (size_t)ym * W, *r1 = q + (size_t)py * W, *r2 = q + (size_t)yp * W;
        const int a = r0[xm], b = r0[px], c = r0[xp], d = r1[xm], f = r1[xp], g = r2[xm], h = r2[px], i = r2[xp];
        const int gx = (c - a) + 2 * (f - d) + (i - g);
        const int gy = (g - a) + 2 * (h - b) + (i - c);
        const float sx = (float)gx / 255.0f, sy = (float)gy / 255.0f;
        const float s2 = sx * sx + sy * sy;
        mag = sqrtf(s2);
        out[(size_t)py * W + px] = mag;
    }
    float m = wave_max(mag);
    __shared__ float sm[4];
    if (lane_id() == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) blockmax[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

__global__ __launch_bounds__(256) void k8_div(float *__restrict__ x, int64_t n, float den)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = x[i] / den;
}

// cv2.resize INTER_LINEAR float32: horizontal taps (edge taps get weight 0), vertical taps clamp rows
// Row-striped form: the local source holds rows [src_row0, src_row0 + sh_local) of a source sh rows tall and the
// local destination rows [dst_row0, dst_row0 + dh_local) of a destination dh rows tall; taps are computed in
// GLOBAL coordinates, so a stripe gets exactly the values of the un-sharded call.
__device__ __forceinline__ float resize_px(const float *__restrict__ src, int sh, int sw, double scale_x, double scale_y, int src_row0,
                                           int px, int py)
{
    float fx = (float)(((double)px + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx = fx - (float)sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= sw - 1) { sx = sw - 1; fx = 0.f; }
    const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
    const float a0 = 1.0f - fx, a1 = fx;
    float fy = (float)(((double)py + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy = fy - (float)sy;
    const int y0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
    const int y1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
    const float b0 = 1.0f - fy, b1 = fy;
    const float *r0 = src + (size_t)(y0 - src_row0) * sw, *r1 = src + (size_t)(y1 - src_row0) * sw;
    const float t00 = r0[sx] * a0, t01 = r0[sx1] * a1, t10 = r1[sx] * a0, t11 = r1[sx1] * a1;
    const float h0 = t00 + t01, h1 = t10 + t11;
    const float u0 = h0 * b0, u1 = h1 * b1;
    return u0 + u1;
}

// A fixed number of workgroups walk the 64 x 4 tiles (measured faster than one workgroup per tile: 0.63 against 0.72 ms
// per 16384^2 plane).  MM (rsseg_ctx_collect_minmax): the plane's minimum / maximum are committed once per wave at the end
// (one global atomic per TILE would serialise on one address).
// (Sharing the right-hand tap with the next lane by shuffle, and four pixels per lane with one 16-byte store, were both
// measured slower than the plain per-pixel form.)
template <bool MM>
__global__ __launch_bounds__(256) void k5_resize(const float *__restrict__ src, int sh, int sw, float *__restrict__ dst, int dh,
                                                 int dw, double scale_x, double scale_y, int src_row0, int dst_row0, int dh_local,
                                                 int gx, int64_t ntiles, uint32_t *__restrict__ mm)
{
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int px = (int)(tile % gx) * WG_X + (threadIdx.x & 63), pyl = (int)(tile / gx) * WG_Y + (threadIdx.x >> 6);
        if (px < dw && pyl < dh_local) {
            const float v = resize_px(src, sh, sw, scale_x, scale_y, src_row0, px, pyl + dst_row0);
            dst[(size_t)pyl * dw + px] = v;
            if (MM) {
                const float z = v != v ? 0.f : v;
                mn = fminf(mn, z);
                mx = fmaxf(mx, z);
            }
        }
    }
    if (MM) mm_commit(mm, mn, mx);
}

static dim3 grid2d(int H, int W) { return dim3((W + WG_X - 1) / WG_X, (H + WG_Y - 1) / WG_Y); }

extern "C" int rsseg_box_mean_f32(rsseg_ctx *ctx, const float *d_x, int H, int W, int k, int border, int square, float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || !d_out || H < 1 || W < 1 || (border != 0 && border != 1)) return rs_fail(ctx, RSSEG_ERR_INVALID, "box_mean: bad arguments");
    if (d_x == d_out) return rs_fail(ctx, RSSEG_ERR_INVALID, "box_mean: in-place not supported");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "box");
        switch (k) {
        case 3: hipLaunchKernelGGL(k6_box<3>, grid2d(H, W), dim3(256), 0, ctx->stream, d_x, H, W, border, square, d_out); break;
        case 5: hipLaunchKernelGGL(k6_box<5>, grid2d(H, W), dim3(256), 0, ctx->stream, d_x, H, W, border, square, d_out); break;
        case 7: hipLaunchKernelGGL(k6_box<7>, grid2d(H, W), dim3(256), 0, ctx->stream, d_x, H, W, border, square, d_out); break;
        case 9: hipLaunchKernelGGL(k6_box<9>, grid2d(H, W), dim3(256), 0, ctx->stream, d_x, H, W, border, square, d_out); break;
        default: return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "box_mean: kernel size %d not in {3,5,7,9}", k);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_local_std_f32(rsseg_ctx *ctx, const float *d_x, int H, int W, int k, float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || !d_out || H < 1 || W < 1 || d_x == d_out) return rs_fail(ctx, RSSEG_ERR_INVALID, "local_std: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "box");
        switch (k) {
        case 3: hipLaunchKernelGGL((k6_std<3, false>), grid2d(H, W), dim3(256), 0, ctx->stream, d_x, H, W, d_out); break;
        case 5: hipLaunchKernelGGL((k6_std<5, false>), grid2d(H, W), dim3(256), 0, ctx->stream, d_x, H, W, d_out); break;
        case 7: hipLaunchKernelGGL((k6_std<7, false>), grid2d(H, W), dim3(256), 0, ctx->stream, d_x, H, W, d_out); break;
        default: return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "local_std: kernel size %d not in {3,5,7}", k);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_morph_gradient_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int k, uint8_t *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_q || !d_out || H < 1 || W < 1 || d_q == d_out) return rs_fail(ctx, RSSEG_ERR_INVALID, "morph_gradient: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "stencil");
        switch (k) {
        case 3: hipLaunchKernelGGL(k7_morph_grad<3>, grid2d(H, W), dim3(256), 0, ctx->stream, d_q, H, W, d_out); break;
        case 5: hipLaunchKernelGGL(k7_morph_grad<5>, grid2d(H, W), dim3(256), 0, ctx->stream, d_q, H, W, d_out); break;
        case 7: hipLaunchKernelGGL(k7_morph_grad<7>, grid2d(H, W), dim3(256), 0, ctx->stream, d_q, H, W, d_out); break;
        default: return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "morph_gradient: kernel size %d not in {3,5,7}", k);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_local_var_f32(rsseg_ctx *ctx, const float *d_x, int H, int W, int k, float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || !d_out || H < 1 || W < 1 || d_x == d_out) return rs_fail(ctx, RSSEG_ERR_INVALID, "local_var: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "box");
        switch (k) {
        case 3: hipLaunchKernelGGL((k6_std<3, true>), grid2d(H, W), dim3(256), 0, ctx->stream, d_x, H, W, d_out); break;
        case 5: hipLaunchKernelGGL((k6_std<5, true>), grid2d(H, W), dim3(256), 0, ctx->stream, d_x, H, W, d_out); break;
        case 7: hipLaunchKernelGGL((k6_std<7, true>), grid2d(H, W), dim3(256), 0, ctx->stream, d_x, H, W, d_out); break;
        default: return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "local_var: kernel size %d not in {3,5,7}", k);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

template <int MODE> static int morph_launch(rsseg_ctx *ctx, const uint8_t *in, int H, int W, int k, uint8_t *out)
{
    prof_scope ps(ctx, "stencil");
    switch (k) {
    case 3: hipLaunchKernelGGL((k7_morph<3, MODE>), grid2d(H, W), dim3(256), 0, ctx->stream, in, H, W, out); break;
    case 5: hipLaunchKernelGGL((k7_morph<5, MODE>), grid2d(H, W), dim3(256), 0, ctx->stream, in, H, W, out); break;
    case 7: hipLaunchKernelGGL((k7_morph<7, MODE>), grid2d(H, W), dim3(256), 0, ctx->stream, in, H, W, out); break;
    default: return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "morph: kernel size %d not in {3,5,7}", k);
    }
    return RSSEG_OK;
}

extern "C" int rsseg_morph_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int k, int op, uint8_t *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_q || !d_out || H < 1 || W < 1 || d_q == d_out) return rs_fail(ctx, RSSEG_ERR_INVALID, "morph: bad arguments");
    if (op == RSSEG_MORPH_GRADIENT) return rsseg_morph_gradient_u8(ctx, d_q, H, W, k, d_out);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (op == RSSEG_MORPH_ERODE) RSCHK(morph_launch<0>(ctx, d_q, H, W, k, d_out));
    else if (op == RSSEG_MORPH_DILATE) RSCHK(morph_launch<1>(ctx, d_q, H, W, k, d_out));
    else if (op == RSSEG_MORPH_OPEN || op == RSSEG_MORPH_CLOSE) {
        RSCHK(ws_reserve(ctx, (size_t)H * W));
        uint8_t *tmp = (uint8_t *)ctx->d_ws;
        if (op == RSSEG_MORPH_OPEN) {
            RSCHK(morph_launch<0>(ctx, d_q, H, W, k, tmp));
            RSCHK(morph_launch<1>(ctx, tmp, H, W, k, d_out));
        } else {
            RSCHK(morph_launch<1>(ctx, d_q, H, W, k, tmp));
            RSCHK(morph_launch<0>(ctx, tmp, H, W, k, d_out));
        }
    } else
        return rs_fail(ctx, RSSEG_ERR_INVALID, "morph: unknown operation %d", op);
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_laplacian_norm_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_q || !d_out || H < 1 || W < 1) return rs_fail(ctx, RSSEG_ERR_INVALID, "laplacian: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const dim3 g = grid2d(H, W);
    const size_t nb = (size_t)g.x * g.y;
    RSCHK(ws_reserve(ctx, 2 * nb * sizeof(float)));
    RSCHK(pin_reserve(ctx, 2 * nb * sizeof(float)));
    {
        prof_scope ps(ctx, "stencil");
        hipLaunchKernelGGL(k8_laplace, g, dim3(256), 0, ctx->stream, d_q, H, W, d_out, (float *)ctx->d_ws, (float *)ctx->d_ws + nb);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, ctx->d_ws, 2 * nb * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = 0; i < nb; i++) {
        mn = std::min(mn, ((const float *)ctx->h_pin)[i]);
        mx = std::max(mx, ((const float *)ctx->h_pin)[nb + i]);
    }
    double mm[2] = {-(double)mn, (double)mx};  // MAX-reduce of the negated minimum
    RSCHK(comm_allreduce_host(ctx, mm, 2, RSSEG_F64, RSSEG_MAX));
    volatile float fmn = (float)(-mm[0]), fmx = (float)mm[1];
    volatile float range = fmx - fmn;
    volatile float den = range + 1e-10f;  // float32 throughout (NumPy 2 weak-scalar promotion)
    {
        prof_scope ps(ctx, "stencil");
        hipLaunchKernelGGL(k8_sub_div, dim3((unsigned)std::min<int64_t>(2048, ceil_div64((int64_t)H * W, 256))), dim3(256), 0, ctx->stream,
                           d_out, (int64_t)H * W, (float)fmn, (float)den);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_sobel_mag_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_q || !d_out || H < 1 || W < 1) return rs_fail(ctx, RSSEG_ERR_INVALID, "sobel_mag: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const dim3 g = grid2d(H, W);
    const size_t nb = (size_t)g.x * g.y;
    RSCHK(ws_reserve(ctx, nb * sizeof(float)));
    RSCHK(pin_reserve(ctx, nb * sizeof(float)));
    {
        prof_scope ps(ctx, "stencil");
        hipLaunchKernelGGL(k8_sobel, g, dim3(256), 0, ctx->stream, d_q, H, W, d_out, (float *)ctx->d_ws);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, ctx->d_ws, nb * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    float mx = 0.f;
    for (size_t i = 0; i < nb; i++) mx = std::max(mx, ((const float *)ctx->h_pin)[i]);
    double mxd = mx;
    RSCHK(comm_allreduce_host(ctx, &mxd, 1, RSSEG_F64, RSSEG_MAX));
    volatile float den = (float)mxd + 1e-10f;  // sobel_mag.max() + 1e-10 in float32
    {
        prof_scope ps(ctx, "stencil");
        hipLaunchKernelGGL(k8_div, dim3((unsigned)std::min<int64_t>(2048, ceil_div64((int64_t)H * W, 256))), dim3(256), 0, ctx->stream,
                           d_out, (int64_t)H * W, (float)den);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

static int resize_rows(rsseg_ctx *ctx, const float *d_src, int sh_local, int sw, int src_row0, int sh, float *d_dst, int dh_local,
                       int dw, int dst_row0, int dh)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_src || !d_dst || sh < 1 || sw < 1 || dh < 1 || dw < 1 || sh_local < 1 || dh_local < 1 || src_row0 < 0 || dst_row0 < 0 ||
        src_row0 + sh_local > sh || dst_row0 + dh_local > dh)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "resize: bad arguments");
    const double scale_x = 1.0 / ((double)dw / (double)sw), scale_y = 1.0 / ((double)dh / (double)sh);
    // the source stripe must hold every row the destination stripe taps
    auto tap = [&](int py) {
        float fy = (float)(((double)py + 0.5) * scale_y - 0.5);
        return (int)floorf(fy);
    };
    const int need0 = std::min(std::max(tap(dst_row0), 0), sh - 1), need1 = std::min(std::max(tap(dst_row0 + dh_local - 1) + 1, 0), sh - 1);
    if (need0 < src_row0 || need1 >= src_row0 + sh_local)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "resize: source stripe rows [%d,%d) do not cover the tapped rows [%d,%d]", src_row0,
                       src_row0 + sh_local, need0, need1);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    RSCHK(mm_begin(ctx, 1));
    {
        prof_scope ps(ctx, "resize");
        const dim3 g = grid2d(dh_local, dw);
        const int64_t ntiles = (int64_t)g.x * g.y;
        const dim3 pg((unsigned)std::min<int64_t>(ntiles, 8192));
        if (ctx->mm_collect)
            hipLaunchKernelGGL(k5_resize<true>, pg, dim3(256), 0, ctx->stream, d_src, sh, sw, d_dst, dh, dw, scale_x, scale_y, src_row0, dst_row0,
                               dh_local, (int)g.x, ntiles, ctx->d_mm);
        else
            hipLaunchKernelGGL(k5_resize<false>, pg, dim3(256), 0, ctx->stream, d_src, sh, sw, d_dst, dh, dw, scale_x, scale_y, src_row0, dst_row0,
                               dh_local, (int)g.x, ntiles, (uint32_t *)nullptr);
    }
    HIPCHK(ctx, hipGetLastError());
    RSCHK(mm_end(ctx, 1));
    return stream_sync(ctx);
}

extern "C" int rsseg_resize_bilinear_f32(rsseg_ctx *ctx, const float *d_src, int sh, int sw, float *d_dst, int dh, int dw)
{
    return resize_rows(ctx, d_src, sh, sw, 0, sh, d_dst, dh, dw, 0, dh);
}

extern "C" int rsseg_resize_bilinear_rows_f32(rsseg_ctx *ctx, const float *d_src, int sh_local, int sw, int src_row0, int sh,
                                              float *d_dst, int dh_local, int dw, int dst_row0, int dh)
{
    return resize_rows(ctx, d_src, sh_local, sw, src_row0, sh, d_dst, dh_local, dw, dst_row0, dh);
}
