// K2 — percentile normalisation + the seven spectral indices, fused, float32, bit-exact with the
// NumPy expressions of the reference (modules/features/indices.py:42-46, 62-69, 86-93, 109-112,
// 128-135, 150-156, 171-177, 194-201).  Band-planar input, 16 B per lane per band, grid-stride.
// HBM-bound: reads 5 planes (20 B/px), writes up to 7 (+5 normalised) planes.
// Compiled with -ffp-contract=off: every NumPy operation is one IEEE float32 operation here.
#include "common.h"

#define K2_THREADS 256

struct k2_args {
    const void *band[5];   // float32 planes, or uint8 planes (k2_indices<.., true>)
    float *out[7];
    float *norm[5];
    float lo[5], hi[5], den[5];
    int normalise;
    evi_coef_t evi;
};

// o: ndvi, evi, msavi, ndwi, mndwi, ndbi, bsi
__device__ __forceinline__ void k2_pixel(const k2_args &a, const float nb[5], float o[7]) { indices_pixel(a.evi, nb, o); }

// MM: also reduce min / max of the 7 index planes into mm[j] (rsseg_ctx_collect_minmax)
// U8: the bands are 8-bit planes (1 byte per pixel instead of 4).  A byte has 256 values, so robust_normalize of a band is a
// 256-entry table — filled by every workgroup with the SAME float32 operations the float path applies per pixel (norm1 of
// (float)v), hence the same bits — and the 5 IEEE divisions per pixel become 5 LDS look-ups.
template <bool MM, bool U8>
__global__ __launch_bounds__(K2_THREADS) void k2_indices(k2_args a, int64_t n, uint32_t *__restrict__ mm)
{
    __shared__ float lut[U8 ? 5 * 256 : 1];
    if (U8) {
        for (int i = threadIdx.x; i < 5 * 256; i += K2_THREADS) {
            const int j = i >> 8;
            const float v = (float)(i & 255);
            lut[i] = a.normalise ? norm1(v, a.lo[j], a.hi[j], a.den[j]) : v;
        }
        __syncthreads();
    }
    const int64_t n4 = n >> 2;
    float lmn[7], lmx[7];
#pragma unroll
    for (int j = 0; j < 7; j++) { lmn[j] = INFINITY; lmx[j] = -INFINITY; }
    for (int64_t i = (int64_t)blockIdx.x * K2_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * K2_THREADS) {
        float nb[4][5], o[4][7];
        if (U8) {
            uint32_t w[5];
#pragma unroll
            for (int j = 0; j < 5; j++) w[j] = ld_stream_u32(a.band[j], i);
#pragma unroll
            for (int j = 0; j < 5; j++)
#pragma unroll
                for (int p = 0; p < 4; p++) nb[p][j] = lut[j * 256 + ((w[j] >> (8 * p)) & 255u)];
        } else {
            float4 b[5];
#pragma unroll
            for (int j = 0; j < 5; j++) b[j] = ld_stream_f4(a.band[j], i);
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const float v[4] = {b[j].x, b[j].y, b[j].z, b[j].w};
#pragma unroll
                for (int p = 0; p < 4; p++) nb[p][j] = a.normalise ? norm1(v[p], a.lo[j], a.hi[j], a.den[j]) : v[p];
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) k2_pixel(a, nb[p], o[p]);
        if (MM) {
#pragma unroll
            for (int j = 0; j < 7; j++)
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const float v = o[p][j] != o[p][j] ? 0.f : o[p][j];
                    lmn[j] = fminf(lmn[j], v);
                    lmx[j] = fmaxf(lmx[j], v);
                }
        }
#pragma unroll
        for (int j = 0; j < 7; j++)
            if (a.out[j]) reinterpret_cast<float4 *>(a.out[j])[i] = make_float4(o[0][j], o[1][j], o[2][j], o[3][j]);
#pragma unroll
        for (int j = 0; j < 5; j++)
            if (a.norm[j]) reinterpret_cast<float4 *>(a.norm[j])[i] = make_float4(nb[0][j], nb[1][j], nb[2][j], nb[3][j]);
    }
    // tail (n % 4 pixels)
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * K2_THREADS + threadIdx.x;
    if (t < n) {
        float nb[5], o[7];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            if (U8) nb[j] = lut[j * 256 + reinterpret_cast<const uint8_t *>(a.band[j])[t]];
            else {
                const float v = reinterpret_cast<const float *>(a.band[j])[t];
                nb[j] = a.normalise ? norm1(v, a.lo[j], a.hi[j], a.den[j]) : v;
            }
        }
        k2_pixel(a, nb, o);
        if (MM) {
#pragma unroll
            for (int j = 0; j < 7; j++) {
                const float v = o[j] != o[j] ? 0.f : o[j];
                lmn[j] = fminf(lmn[j], v);
                lmx[j] = fmaxf(lmx[j], v);
            }
        }
#pragma unroll
        for (int j = 0; j < 7; j++)
            if (a.out[j]) a.out[j][t] = o[j];
#pragma unroll
        for (int j = 0; j < 5; j++)
            if (a.norm[j]) a.norm[j][t] = nb[j];
    }
    if (MM) {
#pragma unroll
        for (int j = 0; j < 7; j++) mm_commit_wg(mm + 2 * j, lmn[j], lmx[j]);
    }
}

__global__ __launch_bounds__(K2_THREADS) void k2_normalize(const float *__restrict__ x, float *__restrict__ y, int64_t n,
                                                          float lo, float hi, float den)
{
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * K2_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * K2_THREADS) {
        float4 v = reinterpret_cast<const float4 *>(x)[i];
        v.x = norm1(v.x, lo, hi, den); v.y = norm1(v.y, lo, hi, den);
        v.z = norm1(v.z, lo, hi, den); v.w = norm1(v.w, lo, hi, den);
        reinterpret_cast<float4 *>(y)[i] = v;
    }
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * K2_THREADS + threadIdx.x;
    if (t < n) y[t] = norm1(x[t], lo, hi, den);
}

// NORM: robust_normalize(x) with the given percentiles first (the texture functions re-normalise the band they are
// handed, indices.py:265, 412, 455), then the truncation — one pass, no float32 intermediate plane
template <bool NORM>
__global__ __launch_bounds__(K2_THREADS) void k2_quantize(const float *__restrict__ x, uint8_t *__restrict__ q, int64_t n, float mult,
                                                         float lo, float hi, float den)
{
    // (x * mult).astype(np.uint8): truncation toward zero of a value in [0, mult]
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * K2_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * K2_THREADS) {
        float4 v = reinterpret_cast<const float4 *>(x)[i];
        if (NORM) {
            v.x = norm1(v.x, lo, hi, den); v.y = norm1(v.y, lo, hi, den);
            v.z = norm1(v.z, lo, hi, den); v.w = norm1(v.w, lo, hi, den);
        }
        uchar4 o;
        o.x = (uint8_t)(int)(v.x * mult); o.y = (uint8_t)(int)(v.y * mult);
        o.z = (uint8_t)(int)(v.z * mult); o.w = (uint8_t)(int)(v.w * mult);
        reinterpret_cast<uchar4 *>(q)[i] = o;
    }
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * K2_THREADS + threadIdx.x;
    if (t < n) q[t] = (uint8_t)(int)((NORM ? norm1(x[t], lo, hi, den) : x[t]) * mult);
}

__global__ __launch_bounds__(K2_THREADS) void k2_u8_unit(const uint8_t *__restrict__ q, float *__restrict__ y, int64_t n)
{
    // uint8 / 255.0 evaluated in float64 (indices.py:436-440), then the float32 cast sklearn applies to X
    for (int64_t i = (int64_t)blockIdx.x * K2_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * K2_THREADS)
        y[i] = (float)((double)q[i] / 255.0);
}

__global__ __launch_bounds__(K2_THREADS) void k2_u8_widen(const uint8_t *__restrict__ q, float *__restrict__ y, int64_t n)
{
    // .astype(np.float32) of an 8-bit band (scripts/2_feature_extraction.py:156): exact
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * K2_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * K2_THREADS) {
        const uint32_t w = ld_stream_u32(q, i);
        reinterpret_cast<float4 *>(y)[i] = make_float4((float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24));
    }
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * K2_THREADS + threadIdx.x;
    if (t < n) y[t] = (float)q[t];
}

static int stream_grid(int64_t n4)
{
    return (int)std::min<int64_t>(2048, std::max<int64_t>(1, ceil_div64(n4, K2_THREADS)));
}

extern "C" int rsseg_normalize_f32(rsseg_ctx *ctx, const float *d_x, int64_t n, float lo, float hi, float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || !d_out || n < 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "normalize: bad arguments");
    if ((((uintptr_t)d_x | (uintptr_t)d_out) & 15) != 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "normalize: planes must be 16-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "normalize");
        hipLaunchKernelGGL(k2_normalize, dim3(stream_grid(n >> 2)), dim3(K2_THREADS), 0, ctx->stream, d_x, d_out, n, lo, hi, norm_den(lo, hi));
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_spectral_indices_f32(rsseg_ctx *ctx, const float *const *d_bands, int64_t n, const float *lohi,
                                          float *const *d_out, float *const *d_norm)
{
    return rsseg_spectral_indices_evi_f32(ctx, d_bands, n, lohi, d_out, d_norm, nullptr);
}

static int indices_core(rsseg_ctx *ctx, const void *const *d_bands, bool u8, int64_t n, const float *lohi, float *const *d_out, float *const *d_norm,
                        const float *evi_coef);

extern "C" int rsseg_spectral_indices_evi_f32(rsseg_ctx *ctx, const float *const *d_bands, int64_t n, const float *lohi,
                                              float *const *d_out, float *const *d_norm, const float *evi_coef)
{
    return indices_core(ctx, (const void *const *)d_bands, false, n, lohi, d_out, d_norm, evi_coef);
}

extern "C" int rsseg_spectral_indices_evi_u8(rsseg_ctx *ctx, const uint8_t *const *d_bands, int64_t n, const float *lohi,
                                             float *const *d_out, float *const *d_norm, const float *evi_coef)
{
    return indices_core(ctx, (const void *const *)d_bands, true, n, lohi, d_out, d_norm, evi_coef);
}

static int indices_core(rsseg_ctx *ctx, const void *const *d_bands, bool u8, int64_t n, const float *lohi, float *const *d_out, float *const *d_norm,
                        const float *evi_coef)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_bands || !d_out || n < 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "spectral_indices: bad arguments");
    k2_args a;
    memset(&a, 0, sizeof(a));
    for (int j = 0; j < 5; j++) {
        if (!d_bands[j] || ((uintptr_t)d_bands[j] & 15)) return rs_fail(ctx, RSSEG_ERR_INVALID, "spectral_indices: band %d null or unaligned", j);
        a.band[j] = d_bands[j];
        a.norm[j] = d_norm ? d_norm[j] : nullptr;
        if (a.norm[j] && ((uintptr_t)a.norm[j] & 15)) return rs_fail(ctx, RSSEG_ERR_INVALID, "spectral_indices: norm plane unaligned");
        if (lohi) {
            a.lo[j] = lohi[2 * j];
            a.hi[j] = lohi[2 * j + 1];
            a.den[j] = norm_den(a.lo[j], a.hi[j]);
        }
    }
    a.normalise = lohi != nullptr;
    a.evi.L = evi_coef ? evi_coef[0] : 1.0f;
    a.evi.C1 = evi_coef ? evi_coef[1] : 6.0f;
    a.evi.C2 = evi_coef ? evi_coef[2] : 7.5f;
    a.evi.G = evi_coef ? evi_coef[3] : 2.5f;
    for (int j = 0; j < 7; j++) {
        a.out[j] = d_out[j];
        if (a.out[j] && ((uintptr_t)a.out[j] & 15)) return rs_fail(ctx, RSSEG_ERR_INVALID, "spectral_indices: output plane unaligned");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    RSCHK(mm_begin(ctx, 7));
    {
        prof_scope ps(ctx, "indices");
        const dim3 g(stream_grid(n >> 2));
        if (u8) {
            if (ctx->mm_collect) hipLaunchKernelGGL((k2_indices<true, true>), g, dim3(K2_THREADS), 0, ctx->stream, a, n, ctx->d_mm);
            else hipLaunchKernelGGL((k2_indices<false, true>), g, dim3(K2_THREADS), 0, ctx->stream, a, n, (uint32_t *)nullptr);
        } else {
            if (ctx->mm_collect) hipLaunchKernelGGL((k2_indices<true, false>), g, dim3(K2_THREADS), 0, ctx->stream, a, n, ctx->d_mm);
            else hipLaunchKernelGGL((k2_indices<false, false>), g, dim3(K2_THREADS), 0, ctx->stream, a, n, (uint32_t *)nullptr);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    RSCHK(mm_end(ctx, 7));
    return ctx->mm_collect ? RSSEG_OK : stream_sync(ctx);   // the extrema read-back has already waited for the stream
}

extern "C" int rsseg_quantize_u8(rsseg_ctx *ctx, const float *d_x, int64_t n, float mult, uint8_t *d_q)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || !d_q || n < 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "quantize: bad arguments");
    if (((uintptr_t)d_x & 15) || ((uintptr_t)d_q & 3)) return rs_fail(ctx, RSSEG_ERR_INVALID, "quantize: unaligned plane");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "quantize");
        hipLaunchKernelGGL(k2_quantize<false>, dim3(stream_grid(n >> 2)), dim3(K2_THREADS), 0, ctx->stream, d_x, d_q, n, mult, 0.f, 1.f, 1.f);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_normalize_quantize_u8(rsseg_ctx *ctx, const float *d_x, int64_t n, float lo, float hi, float mult, uint8_t *d_q)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || !d_q || n < 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "normalize_quantize: bad arguments");
    if (((uintptr_t)d_x & 15) || ((uintptr_t)d_q & 3)) return rs_fail(ctx, RSSEG_ERR_INVALID, "normalize_quantize: unaligned plane");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "quantize");
        hipLaunchKernelGGL(k2_quantize<true>, dim3(stream_grid(n >> 2)), dim3(K2_THREADS), 0, ctx->stream, d_x, d_q, n, mult, lo, hi,
                           norm_den(lo, hi));
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_u8_to_f32(rsseg_ctx *ctx, const uint8_t *d_q, int64_t n, float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_q || !d_out || n < 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "u8_to_f32: bad arguments");
    if (((uintptr_t)d_q & 3) || ((uintptr_t)d_out & 15)) return rs_fail(ctx, RSSEG_ERR_INVALID, "u8_to_f32: unaligned plane");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "quantize");
        hipLaunchKernelGGL(k2_u8_widen, dim3(stream_grid(n >> 2)), dim3(K2_THREADS), 0, ctx->stream, d_q, d_out, n);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_u8_to_unit_f32(rsseg_ctx *ctx, const uint8_t *d_q, int64_t n, float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_q || !d_out || n < 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "u8_to_unit: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "quantize");
        hipLaunchKernelGGL(k2_u8_unit, dim3(stream_grid(n)), dim3(K2_THREADS), 0, ctx->stream, d_q, d_out, n);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}
