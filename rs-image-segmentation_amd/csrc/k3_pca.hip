// K3 — PCA of the band stack: RobustScaler transform on the fly, exact Gram / mean accumulation,
// 7x7 symmetric eigen-solve on the host, projection kernel.
//
// Replaces perform_pca (reference modules/features/indices.py:205-246):
//   RobustScaler.transform   X -= center_ (float32);  X /= scale_ (float64 divisor, result float32)
//                            sklearn/preprocessing/_data.py:1716-1718
//   PCA._fit_full, covariance_eigh   sklearn/decomposition/_pca.py:560, 600-646
//   PCA._transform                    sklearn/decomposition/_base.py:148-155
// The scaled matrix X (N x B) is never materialised: both kernels recompute it from the band planes
// (28 B/px read each; the projection writes 4 B/px per component).  The Gram matrix is accumulated
// as exact fixed-point sums (independent of launch geometry / sharding); B = 7 makes this 3.5 flop/B,
// i.e. HBM-bound, so it runs on the vector ALUs next to the loads rather than through MFMA tiles.
#include <cmath>

#include "common.h"

#define PCA_THREADS 256
#define PCA_MAXB 8
#define PCA_NACC (PCA_MAXB + PCA_MAXB * (PCA_MAXB + 1) / 2)
#define PCA_PSTRIDE (2 * PCA_NACC + 2)  // long longs per block in the partial table of k3_gram

struct pca_args {
    const void *band[PCA_MAXB];   // float32 planes, or uint8 planes (the <.., U8 = true> kernels)
    float center[PCA_MAXB];
    double scale[PCA_MAXB];
    double rinv[PCA_MAXB];  // RN(1 / scale)
    int slow_div;           // a scale whose reciprocal trick is not provably exact: use the IEEE division
    float nlo[PCA_MAXB], nhi[PCA_MAXB], nden[PCA_MAXB];  // raw bands: robust_normalize(v) with these percentiles first
    int normalise;
    int nb;
    int scaled;
    double fx_scale;  // 2^Q used for the fixed-point accumulation of x and x*x
};

__device__ __forceinline__ float pca_x(const pca_args &a, int b, float v)
{
    if (a.normalise) v = norm1(v, a.nlo[b], a.nhi[b], a.nden[b]);
    if (!a.scaled) return v;
    const float d = v - a.center[b];
    if (a.slow_div) return (float)((double)d / a.scale[b]);
    // RN(d / s) without the ~35-instruction IEEE division sequence: q = RN(d * RN(1/s)) is a faithful quotient, the
    // fma residual r = d - q*s is exact, and RN(q + r * RN(1/s)) is then the correctly rounded quotient (Markstein's
    // theorem; it needs RN(1/s) and a significand of s that is not all ones — checked on the host).
    const double s = a.scale[b], y = a.rinv[b];
    const double q = (double)d * y;
    const double r = fma(-q, s, (double)d);
    return (float)fma(r, y, q);
}

__device__ __forceinline__ long long to_fixed_q(double x, double s)
{
    double d = fma(x, s, FX_MAGIC);
    return __double_as_longlong(d) - __double_as_longlong(FX_MAGIC);
}

// partial[blk][PCA_NACC][2] ({hi, lo} 32-bit limb sums): sums of x_b (nb entries) then x_a*x_b for a <= b (row-major upper
// triangle of a PCA_MAXB x PCA_MAXB matrix).  The accumulators take the raw bit patterns of fma(v, 2^Q, 1.5*2^52); the
// constant bits(1.5*2^52) is subtracted once per thread (count * constant, modulo 2^64) instead of once per term.
__host__ __device__ constexpr int pca_tri(int b, int c) { return PCA_MAXB + b * PCA_MAXB - b * (b - 1) / 2 + (c - b); }

// U8: the bands are 8-bit planes.  pca_x of a band is then a 256-entry table, filled by every workgroup with the operations
// the float path applies per pixel (same bits): the two divisions per band and pixel become one LDS look-up.
template <int NB, bool U8> __device__ __forceinline__ void pca_fill_lut(const pca_args &a, float *lut)
{
    if (U8) {   // (the float32 try-pass of k3_gram passes U8 = true here: the same table)
        for (int i = threadIdx.x; i < NB * 256; i += PCA_THREADS) lut[i] = pca_x(a, i >> 8, (float)(i & 255));
        __syncthreads();
    }
}

// FL (float32 planes, tried first when the select has just seen these planes hold only integers 0..255 — the reference's
// preprocessed tiles are 8-bit digital numbers stored as float32): the table of the U8 form, indexed by the value; a value
// that is not such an integer (or NaN) is counted in the partial table's spare slot and the host runs the general form.
template <int NB, bool U8, bool FL = false>
__global__ __launch_bounds__(PCA_THREADS) void k3_gram(pca_args a, int64_t n, long long *__restrict__ partial)
{
    static_assert(!(U8 && FL), "FL is a form of the float32 kernel");
    constexpr bool TAB = U8 || FL;
    __shared__ float lut[TAB ? NB * 256 : 1];
    pca_fill_lut<NB, TAB>(a, lut);
    bool miss = false;
    auto tab = [&](int b, float v) {   // FL: the table entry of an integer-valued v in [0, 255]
        const int idx = (int)fminf(fmaxf(v, -1.f), 256.f);   // NaN and anything outside [0, 255] convert to a defined, missing index
        miss = miss || !(v == (float)idx && (unsigned)idx < 256u);
        return lut[b * 256 + (idx & 255)];
    };
    // partial[blk][2 * PCA_NACC + 2]: the limb sums, then the number of threads that met a NaN (sklearn's PCA rejects
    // NaN input; a NaN's bit pattern in the integer sums would otherwise pass unnoticed), then padding
    bool bad = false;
    unsigned long long acc[PCA_NACC];
#pragma unroll
    for (int i = 0; i < PCA_NACC; i++) acc[i] = 0;
    unsigned long long cnt = 0;
    const int64_t n4 = n >> 2;
    auto pixel = [&](const float *v) {   // U8: v already holds the table values
        float x[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            x[b] = TAB ? v[b] : pca_x(a, b, v[b]);
            bad = bad || x[b] != x[b];
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
            // xs = x_b * 2^Q is exact (a power of two), and so is the product xs * x_c inside the fma (two float32 values: 48 bits):
            // fma(xs, x_c, MAGIC) rounds x_b * x_c * 2^Q to an integer ONCE, like fma(x_b * x_c, 2^Q, MAGIC) with its exact float64
            // product did — the same bits for one float64 multiplication less per term
            const double xs = (double)x[b] * a.fx_scale;
            acc[b] += (unsigned long long)__double_as_longlong(xs + FX_MAGIC);
#pragma unroll
            for (int c = b; c < NB; c++)
                acc[pca_tri(b, c)] += (unsigned long long)__double_as_longlong(fma(xs, (double)x[c], FX_MAGIC));
        }
        cnt++;
    };
    for (int64_t i = (int64_t)blockIdx.x * PCA_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * PCA_THREADS) {
        float p0[NB], p1[NB], p2[NB], p3[NB];
        if (U8) {
            uint32_t w[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) w[b] = ld_stream_u32(a.band[b], i);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                p0[b] = lut[b * 256 + (w[b] & 255u)];
                p1[b] = lut[b * 256 + ((w[b] >> 8) & 255u)];
                p2[b] = lut[b * 256 + ((w[b] >> 16) & 255u)];
                p3[b] = lut[b * 256 + (w[b] >> 24)];
            }
        } else {
            float4 v[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) v[b] = ld_stream_f4(a.band[b], i);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (FL) { p0[b] = tab(b, v[b].x); p1[b] = tab(b, v[b].y); p2[b] = tab(b, v[b].z); p3[b] = tab(b, v[b].w); }
                else { p0[b] = v[b].x; p1[b] = v[b].y; p2[b] = v[b].z; p3[b] = v[b].w; }
            }
        }
        pixel(p0); pixel(p1); pixel(p2); pixel(p3);
    }
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * PCA_THREADS + threadIdx.x;
    if (t < n) {
        float p[NB];
#pragma unroll
        for (int b = 0; b < NB; b++)
            p[b] = U8 ? lut[b * 256 + reinterpret_cast<const uint8_t *>(a.band[b])[t]]
                      : FL ? tab(b, reinterpret_cast<const float *>(a.band[b])[t]) : reinterpret_cast<const float *>(a.band[b])[t];
        pixel(p);
    }
    const unsigned long long bias = cnt * (unsigned long long)__double_as_longlong(FX_MAGIC);
    // per-thread sums stay below 2^61; split into 32-bit limbs before the cross-lane sums
    __shared__ long long sh[4][2 * PCA_NACC];
#pragma unroll
    for (int i = 0; i < PCA_NACC; i++) {
        bool used = i < NB;
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int c = b; c < NB; c++) used = used || i == pca_tri(b, c);
        const long long v = used ? (long long)(acc[i] - bias) : 0ll;
        long long hi = wave_sum(v >> 32), lo = wave_sum(v & 0xffffffffLL);
        if (lane_id() == 0) { sh[threadIdx.x >> 6][2 * i] = hi; sh[threadIdx.x >> 6][2 * i + 1] = lo; }
    }
    const int nbad = __syncthreads_count(bad);
    const int nmiss = FL ? __syncthreads_count(miss) : 0;
    if (threadIdx.x < 2 * PCA_NACC)
        partial[(size_t)blockIdx.x * PCA_PSTRIDE + threadIdx.x] =
            sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
    if (threadIdx.x == 0) {
        partial[(size_t)blockIdx.x * PCA_PSTRIDE + 2 * PCA_NACC] = nbad;
        partial[(size_t)blockIdx.x * PCA_PSTRIDE + 2 * PCA_NACC + 1] = nmiss;
    }
}

// per-band extrema of the planes as given (NaN ignored): sizes the fixed-point quantum of k3_gram for bands that are not
// robust-normalised to [0,1].  part[blk][2 * NB] = {min_b, max_b}
__global__ __launch_bounds__(PCA_THREADS) void k3_range(pca_args a, int64_t off, int64_t n, float *__restrict__ part)
{
    __shared__ float sh[4][2 * PCA_MAXB];
    for (int b = 0; b < a.nb; b++) {
        float mn = INFINITY, mx = -INFINITY;
        for (int64_t i = (int64_t)blockIdx.x * PCA_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * PCA_THREADS) {
            const float v = reinterpret_cast<const float *>(a.band[b])[off + i];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
        mn = wave_min(mn);
        mx = wave_max(mx);
        if (lane_id() == 0) { sh[threadIdx.x >> 6][2 * b] = mn; sh[threadIdx.x >> 6][2 * b + 1] = mx; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * a.nb) {
        const int t = threadIdx.x;
        const float v0 = sh[0][t], v1 = sh[1][t], v2 = sh[2][t], v3 = sh[3][t];
        part[(size_t)blockIdx.x * 2 * PCA_MAXB + t] = (t & 1) ? fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)) : fminf(fminf(v0, v1), fminf(v2, v3));
    }
}

struct proj_args {
    float comp[PCA_MAXB][PCA_MAXB];  // [component][band]
    float offs[PCA_MAXB];            // mean @ components.T
    float *out[PCA_MAXB];
    int nc;
};

// Specialised on the band count; the component loop stays rolled so that only one component's coefficients are
// held in scalar registers at a time (with everything unrolled the two argument blocks no longer fit the SGPR file).
template <int NB, bool MM, bool U8>
__global__ __launch_bounds__(PCA_THREADS) void k3_project(pca_args a, proj_args pr, int64_t n, uint32_t *__restrict__ mm)
{
    __shared__ float lut[U8 ? NB * 256 : 1];
    pca_fill_lut<NB, U8>(a, lut);
    const int64_t n4 = n >> 2;
    float lmn[PCA_MAXB], lmx[PCA_MAXB];  // MM: running extrema per component (indexed by the rolled loop: LDS-free, registers)
#pragma unroll
    for (int c = 0; c < PCA_MAXB; c++) { lmn[c] = INFINITY; lmx[c] = -INFINITY; }
    for (int64_t i = (int64_t)blockIdx.x * PCA_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * PCA_THREADS) {
        float x[4][NB];
        if (U8) {
            uint32_t w[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) w[b] = ld_stream_u32(a.band[b], i);
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int p = 0; p < 4; p++) x[p][b] = lut[b * 256 + ((w[b] >> (8 * p)) & 255u)];
        } else {
            float4 v[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) v[b] = ld_stream_f4(a.band[b], i);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                x[0][b] = pca_x(a, b, v[b].x);
                x[1][b] = pca_x(a, b, v[b].y);
                x[2][b] = pca_x(a, b, v[b].z);
                x[3][b] = pca_x(a, b, v[b].w);
            }
        }
#pragma nounroll
        for (int c = 0; c < pr.nc; c++) {
            float y[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                float s = 0.f;
#pragma unroll
                for (int b = 0; b < NB; b++) s = __fmaf_rn(x[p][b], pr.comp[c][b], s);
                y[p] = s - pr.offs[c];
            }
            reinterpret_cast<float4 *>(pr.out[c])[i] = make_float4(y[0], y[1], y[2], y[3]);
            if (MM) {
                float lo = INFINITY, hi = -INFINITY;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const float v = y[p] != y[p] ? 0.f : y[p];
                    lo = fminf(lo, v);
                    hi = fmaxf(hi, v);
                }
#pragma unroll
                for (int cc = 0; cc < PCA_MAXB; cc++)  // static indices: the arrays stay in registers
                    if (cc == c) { lmn[cc] = fminf(lmn[cc], lo); lmx[cc] = fmaxf(lmx[cc], hi); }
            }
        }
    }
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * PCA_THREADS + threadIdx.x;
    if (t < n) {
        float x[NB];
#pragma unroll
        for (int b = 0; b < NB; b++)
            x[b] = U8 ? lut[b * 256 + reinterpret_cast<const uint8_t *>(a.band[b])[t]] : pca_x(a, b, reinterpret_cast<const float *>(a.band[b])[t]);
#pragma nounroll
        for (int c = 0; c < pr.nc; c++) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < NB; b++) s = __fmaf_rn(x[b], pr.comp[c][b], s);
            const float yv = s - pr.offs[c];
            pr.out[c][t] = yv;
            if (MM) {
                const float v = yv != yv ? 0.f : yv;
#pragma unroll
                for (int cc = 0; cc < PCA_MAXB; cc++)
                    if (cc == c) { lmn[cc] = fminf(lmn[cc], v); lmx[cc] = fmaxf(lmx[cc], v); }
            }
        }
    }
    if (MM) {
#pragma unroll
        for (int cc = 0; cc < PCA_MAXB; cc++)
            if (cc < pr.nc) mm_commit_wg(mm + 2 * cc, lmn[cc], lmx[cc]);
    }
}

// Fused: the seven spectral indices (K2) AND the projection on the principal components in ONE pass over the raw bands
// (both read the same robust-normalised values: the five bands the indices use are normalised once instead of twice and
// read once instead of twice — 28 B/px in, 28 + 4 (normalised NIR) + 4 nc out, against 20 + 32 and 28 + 4 nc).  Bands 0..4
// are blue, green, red, nir, swir1 (the reference's TM order).  Extrema slots (MM): 0..6 the indices, 7.. the components.
struct fuse_args {
    float *idx[7];
    float *norm[5];
    evi_coef_t evi;
    // optional: the texture chain's input — trunc(robust_normalize(normalised NIR; q_lo, q_hi) * q_mult) as uint8
    // (calculate_glcm_features re-normalises the band it receives and quantises it, indices.py:265-268)
    uint8_t *q;
    float q_lo, q_hi, q_den, q_mult;
};

#define FUSE_CHUNK_V 4096
template <int NB, bool MM, bool U8>
__global__ __launch_bounds__(PCA_THREADS) void k3_indices_project(pca_args a, proj_args pr, fuse_args fz, int64_t n, uint32_t *__restrict__ mm, int chunked)
{
    static_assert(NB >= 5, "the indices read bands 0..4");
    __shared__ float lutn[U8 ? 5 * 256 : 1];    // robust_normalize of a byte (bands 0..4)
    __shared__ float lutx[U8 ? NB * 256 : 1];   // pca_x of a byte
    auto quant = [&](float nir_norm) -> uint32_t { return (uint32_t)(uint8_t)(int)(norm1(nir_norm, fz.q_lo, fz.q_hi, fz.q_den) * fz.q_mult); };
    if (U8) {
        for (int i = threadIdx.x; i < 5 * 256; i += PCA_THREADS) lutn[i] = norm1((float)(i & 255), a.nlo[i >> 8], a.nhi[i >> 8], a.nden[i >> 8]);
        for (int i = threadIdx.x; i < NB * 256; i += PCA_THREADS) lutx[i] = pca_x(a, i >> 8, (float)(i & 255));
        __syncthreads();
    }
    // pca_x on an already normalised value (the second half of pca_x)
    auto scale_x = [&](int b, float v) -> float {
        if (!a.scaled) return v;
        const float d = v - a.center[b];
        if (a.slow_div) return (float)((double)d / a.scale[b]);
        const double sc = a.scale[b], y = a.rinv[b];
        const double q = (double)d * y;
        const double r = fma(-q, sc, (double)d);
        return (float)fma(r, y, q);
    };
    const int64_t n4 = n >> 2;
    float lmn[7 + PCA_MAXB], lmx[7 + PCA_MAXB];
#pragma unroll
    for (int c = 0; c < 7 + PCA_MAXB; c++) { lmn[c] = INFINITY; lmx[c] = -INFINITY; }
    auto track = [&](int slot, float v) {
        const float z = v != v ? 0.f : v;
        lmn[slot] = fminf(lmn[slot], z);
        lmx[slot] = fmaxf(lmx[slot], z);
    };
    // ONE loop for both workgroup -> pixel mappings (no lambda around the body: a closure that captures the extrema arrays
    // by reference keeps them in scratch memory — 64 - 128 B per lane, +5 ... +14 B/px of HBM traffic in the r04 PMC tables).
    // chunked: a workgroup owns contiguous chunks of FUSE_CHUNK_V 16-byte vectors (16 tiles of 1024 px, the k-means kernels'
    // mapping): with 7 planes read and 11 written, 18 streams per workgroup, the bare pattern runs 3.26 ms in this mapping
    // against 3.66-3.72 ms grid-strided at 16384^2 (profiles/r04_streams_write_heavy.json)
    const int64_t nchunk = (n4 + FUSE_CHUNK_V - 1) / FUSE_CHUNK_V;
    int64_t cur_chunk = blockIdx.x, gi = (int64_t)blockIdx.x * PCA_THREADS + threadIdx.x;
    int cur_t = 0;
    for (;;) {
        int64_t i;
        if (chunked) {
            if (cur_chunk >= nchunk) break;
            i = cur_chunk * FUSE_CHUNK_V + (int64_t)cur_t * PCA_THREADS + threadIdx.x;
            if (++cur_t == FUSE_CHUNK_V / PCA_THREADS) { cur_t = 0; cur_chunk += gridDim.x; }
            if (i >= n4) continue;
        } else {
            if (gi >= n4) break;
            i = gi;
            gi += (int64_t)gridDim.x * PCA_THREADS;
        }
        float nbv[4][5], x[4][NB];
        if (U8) {
            uint32_t w[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) w[b] = ld_stream_u32(a.band[b], i);
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const uint32_t v = (w[b] >> (8 * p)) & 255u;
                    x[p][b] = lutx[b * 256 + v];
                    if (b < 5) nbv[p][b] = lutn[b * 256 + v];
                }
        } else {
            float4 v[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) v[b] = ld_stream_f4(a.band[b], i);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const float vv[4] = {v[b].x, v[b].y, v[b].z, v[b].w};
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const float nv = norm1(vv[p], a.nlo[b], a.nhi[b], a.nden[b]);
                    if (b < 5) nbv[p][b] = nv;
                    x[p][b] = scale_x(b, nv);
                }
            }
        }
        float o[4][7];
#pragma unroll
        for (int p = 0; p < 4; p++) indices_pixel(fz.evi, nbv[p], o[p]);
#pragma unroll
        for (int j = 0; j < 7; j++) {
            if (fz.idx[j]) reinterpret_cast<float4 *>(fz.idx[j])[i] = make_float4(o[0][j], o[1][j], o[2][j], o[3][j]);
            if (MM) {
#pragma unroll
                for (int p = 0; p < 4; p++) track(j, o[p][j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 5; j++)
            if (fz.norm[j]) reinterpret_cast<float4 *>(fz.norm[j])[i] = make_float4(nbv[0][j], nbv[1][j], nbv[2][j], nbv[3][j]);
        if (fz.q) reinterpret_cast<uint32_t *>(fz.q)[i] = quant(nbv[0][3]) | (quant(nbv[1][3]) << 8) | (quant(nbv[2][3]) << 16) | (quant(nbv[3][3]) << 24);
#pragma nounroll
        for (int c = 0; c < pr.nc; c++) {
            float y[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                float s = 0.f;
#pragma unroll
                for (int b = 0; b < NB; b++) s = __fmaf_rn(x[p][b], pr.comp[c][b], s);
                y[p] = s - pr.offs[c];
            }
            reinterpret_cast<float4 *>(pr.out[c])[i] = make_float4(y[0], y[1], y[2], y[3]);
            if (MM) {
#pragma unroll
                for (int cc = 0; cc < PCA_MAXB; cc++)   // static indices: the arrays stay in registers
                    if (cc == c) {
#pragma unroll
                        for (int p = 0; p < 4; p++) track(7 + cc, y[p]);
                    }
            }
        }
    }
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * PCA_THREADS + threadIdx.x;
    if (t < n) {
        float nbv[5], x[NB], o[7];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (U8) {
                const uint32_t v = reinterpret_cast<const uint8_t *>(a.band[b])[t];
                x[b] = lutx[b * 256 + v];
                if (b < 5) nbv[b] = lutn[b * 256 + v];
            } else {
                const float nv = norm1(reinterpret_cast<const float *>(a.band[b])[t], a.nlo[b], a.nhi[b], a.nden[b]);
                if (b < 5) nbv[b] = nv;
                x[b] = scale_x(b, nv);
            }
        }
        indices_pixel(fz.evi, nbv, o);
#pragma unroll
        for (int j = 0; j < 7; j++) {
            if (fz.idx[j]) fz.idx[j][t] = o[j];
            if (MM) track(j, o[j]);
        }
#pragma unroll
        for (int j = 0; j < 5; j++)
            if (fz.norm[j]) fz.norm[j][t] = nbv[j];
        if (fz.q) fz.q[t] = (uint8_t)quant(nbv[3]);
#pragma nounroll
        for (int c = 0; c < pr.nc; c++) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < NB; b++) s = __fmaf_rn(x[b], pr.comp[c][b], s);
            const float yv = s - pr.offs[c];
            pr.out[c][t] = yv;
            if (MM) {
#pragma unroll
                for (int cc = 0; cc < PCA_MAXB; cc++)
                    if (cc == c) track(7 + cc, yv);
            }
        }
    }
    if (MM) {
#pragma unroll
        for (int j = 0; j < 7; j++) mm_commit_wg(mm + 2 * j, lmn[j], lmx[j]);
#pragma unroll
        for (int cc = 0; cc < PCA_MAXB; cc++)
            if (cc < pr.nc) mm_commit_wg(mm + 2 * (7 + cc), lmn[7 + cc], lmx[7 + cc]);
    }
}

// Column sums of the partial table (nblk rows of PCA_PSTRIDE long longs): the host gets ONE row (720 B) instead of the table
// (1.5 MB at 2048 blocks) and a 90 x 2048 loop.  Limb sums stay far below 2^63 (hi < 2^37, lo < 2^40 per row, <= 2049 rows).
__global__ __launch_bounds__(256) void k3_reduce_partials(const long long *__restrict__ partial, int nblk, long long *__restrict__ out)
{
    const int col = blockIdx.x;
    long long s = 0;
    for (int g = threadIdx.x; g < nblk; g += 256) s += partial[(size_t)g * PCA_PSTRIDE + col];
    s = wave_sum(s);
    __shared__ long long sh[4];
    if (lane_id() == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[col] = sh[0] + sh[1] + sh[2] + sh[3];
}

// cyclic Jacobi for a small symmetric matrix (float64).  V columns = eigenvectors.
static void jacobi_eigh(int n, double A[PCA_MAXB][PCA_MAXB], double V[PCA_MAXB][PCA_MAXB], double *w)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0.0;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) off += A[p][q] * A[p][q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i][i];
}

struct fuse_req {   // non-null: produce the spectral indices (+ normalised bands, + the quantised texture band) in the projection pass
    float *const *d_idx;
    float *const *d_norm;
    const float *evi_coef;
    uint8_t *d_q;
    float q_lo, q_hi, q_mult;
};
static int pca_core(rsseg_ctx *ctx, const void *const *d_bands, bool u8, int nb, int64_t n_local, int64_t fit_off, int64_t fit_n, const float *lohi,
                    const float *center, const double *scale, int n_components, float *const *d_out, float *components,
                    float *explained_variance_ratio, float *mean, float *explained_variance, const fuse_req *fuse = nullptr);

extern "C" int rsseg_pca_fit_transform_f32(rsseg_ctx *ctx, const float *const *d_bands, int nb, int64_t n_local,
                                           const float *center, const double *scale, int n_components, float *const *d_out,
                                           float *components, float *explained_variance_ratio, float *mean,
                                           float *explained_variance)
{
    return pca_core(ctx, (const void *const *)d_bands, false, nb, n_local, 0, n_local, nullptr, center, scale, n_components, d_out, components, explained_variance_ratio,
                    mean, explained_variance);
}

extern "C" int rsseg_pca_fit_transform_raw_f32(rsseg_ctx *ctx, const float *const *d_bands, int nb, int64_t n_local, const float *lohi,
                                               const float *center, const double *scale, int n_components, float *const *d_out,
                                               float *components, float *explained_variance_ratio, float *mean,
                                               float *explained_variance)
{
    if (ctx && !lohi) return rs_fail(ctx, RSSEG_ERR_INVALID, "pca_raw: lohi is required");
    return pca_core(ctx, (const void *const *)d_bands, false, nb, n_local, 0, n_local, lohi, center, scale, n_components, d_out, components, explained_variance_ratio,
                    mean, explained_variance);
}

extern "C" int rsseg_pca_fit_transform_ext_f32(rsseg_ctx *ctx, const float *const *d_bands, int nb, int64_t n_local, int64_t fit_off,
                                               int64_t fit_n, const float *lohi, const float *center, const double *scale, int n_components,
                                               float *const *d_out, float *components, float *explained_variance_ratio, float *mean,
                                               float *explained_variance)
{
    if (ctx && (fit_off < 0 || fit_n < 0 || fit_off + fit_n > n_local)) return rs_fail(ctx, RSSEG_ERR_INVALID, "pca_ext: fit range outside the planes");
    return pca_core(ctx, (const void *const *)d_bands, false, nb, n_local, fit_off, fit_n, lohi, center, scale, n_components, d_out, components, explained_variance_ratio,
                    mean, explained_variance);
}

extern "C" int rsseg_pca_fit_transform_ext_u8(rsseg_ctx *ctx, const uint8_t *const *d_bands, int nb, int64_t n_local, int64_t fit_off,
                                              int64_t fit_n, const float *lohi, const float *center, const double *scale, int n_components,
                                              float *const *d_out, float *components, float *explained_variance_ratio, float *mean,
                                              float *explained_variance)
{
    if (ctx && (fit_off < 0 || fit_n < 0 || fit_off + fit_n > n_local)) return rs_fail(ctx, RSSEG_ERR_INVALID, "pca_ext: fit range outside the planes");
    return pca_core(ctx, (const void *const *)d_bands, true, nb, n_local, fit_off, fit_n, lohi, center, scale, n_components, d_out, components, explained_variance_ratio,
                    mean, explained_variance);
}

static int indices_pca_entry(rsseg_ctx *ctx, const void *const *d_bands, bool u8, int nb, int64_t n_local, int64_t fit_off, int64_t fit_n,
                             const float *lohi, const float *center, const double *scale, int n_components, const float *evi_coef,
                             float *const *d_idx, float *const *d_norm, float *const *d_pc, uint8_t *d_q, float q_lo, float q_hi, float q_mult,
                             float *components, float *explained_variance_ratio, float *mean, float *explained_variance)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (nb < 5) return rs_fail(ctx, RSSEG_ERR_INVALID, "indices_pca: the indices need bands 0..4 (blue, green, red, nir, swir1)");
    if (!lohi || !d_idx || !d_pc) return rs_fail(ctx, RSSEG_ERR_INVALID, "indices_pca: lohi, index planes and component planes are required");
    if (fit_off < 0 || fit_n < 0 || fit_off + fit_n > n_local) return rs_fail(ctx, RSSEG_ERR_INVALID, "indices_pca: fit range outside the planes");
    for (int j = 0; j < 7; j++)
        if (d_idx[j] && ((uintptr_t)d_idx[j] & 15)) return rs_fail(ctx, RSSEG_ERR_INVALID, "indices_pca: index plane %d unaligned", j);
    for (int j = 0; j < 5; j++)
        if (d_norm && d_norm[j] && ((uintptr_t)d_norm[j] & 15)) return rs_fail(ctx, RSSEG_ERR_INVALID, "indices_pca: norm plane %d unaligned", j);
    if (d_q && ((uintptr_t)d_q & 3)) return rs_fail(ctx, RSSEG_ERR_INVALID, "indices_pca: quantised plane unaligned");
    const fuse_req fr = {d_idx, d_norm, evi_coef, d_q, q_lo, q_hi, q_mult};
    return pca_core(ctx, d_bands, u8, nb, n_local, fit_off, fit_n, lohi, center, scale, n_components, d_pc, components, explained_variance_ratio, mean,
                    explained_variance, &fr);
}

extern "C" int rsseg_indices_pca_f32(rsseg_ctx *ctx, const float *const *d_bands, int nb, int64_t n_local, int64_t fit_off, int64_t fit_n,
                                     const float *lohi, const float *center, const double *scale, int n_components, const float *evi_coef,
                                     float *const *d_idx, float *const *d_norm, float *const *d_pc, uint8_t *d_q, float q_lo, float q_hi,
                                     float q_mult, float *components, float *explained_variance_ratio, float *mean, float *explained_variance)
{
    return indices_pca_entry(ctx, (const void *const *)d_bands, false, nb, n_local, fit_off, fit_n, lohi, center, scale, n_components, evi_coef, d_idx, d_norm,
                             d_pc, d_q, q_lo, q_hi, q_mult, components, explained_variance_ratio, mean, explained_variance);
}

extern "C" int rsseg_indices_pca_u8(rsseg_ctx *ctx, const uint8_t *const *d_bands, int nb, int64_t n_local, int64_t fit_off, int64_t fit_n,
                                    const float *lohi, const float *center, const double *scale, int n_components, const float *evi_coef,
                                    float *const *d_idx, float *const *d_norm, float *const *d_pc, uint8_t *d_q, float q_lo, float q_hi,
                                    float q_mult, float *components, float *explained_variance_ratio, float *mean, float *explained_variance)
{
    return indices_pca_entry(ctx, (const void *const *)d_bands, true, nb, n_local, fit_off, fit_n, lohi, center, scale, n_components, evi_coef, d_idx, d_norm,
                             d_pc, d_q, q_lo, q_hi, q_mult, components, explained_variance_ratio, mean, explained_variance);
}

static int pca_core(rsseg_ctx *ctx, const void *const *d_bands, bool u8, int nb, int64_t n_local, int64_t fit_off, int64_t fit_n, const float *lohi,
                    const float *center, const double *scale, int n_components, float *const *d_out, float *components,
                    float *explained_variance_ratio, float *mean, float *explained_variance, const fuse_req *fuse)
{
    const size_t esz = u8 ? 1 : 4;
    if (!ctx) return RSSEG_ERR_INVALID;
    if (d_bands && nb > PCA_MAXB)   // a capacity of the kernels, not a limit of the reference: reported as UNSUPPORTED
        return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "pca: %d band planes, the kernels take at most %d", nb, PCA_MAXB);
    if (!d_bands || nb < 1 || n_components < 1 || n_components > nb || n_local < 0)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "pca: bad arguments (nb=%d, n_components=%d)", nb, n_components);
    if ((center == nullptr) != (scale == nullptr)) return rs_fail(ctx, RSSEG_ERR_INVALID, "pca: center and scale must both be given or both be NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pca_args a;
    memset(&a, 0, sizeof(a));
    a.nb = nb;
    a.normalise = lohi != nullptr;
    a.scaled = center != nullptr;
    for (int b = 0; b < nb; b++) {
        if (!d_bands[b] || ((uintptr_t)d_bands[b] & 15)) return rs_fail(ctx, RSSEG_ERR_INVALID, "pca: band %d null or unaligned", b);
        a.band[b] = d_bands[b];
        a.center[b] = center ? center[b] : 0.f;
        if (lohi) {
            a.nlo[b] = lohi[2 * b];
            a.nhi[b] = lohi[2 * b + 1];
            a.nden[b] = norm_den(a.nlo[b], a.nhi[b]);
        }
        a.scale[b] = scale ? scale[b] : 1.0;
        a.rinv[b] = 1.0 / a.scale[b];
        uint64_t sb;
        memcpy(&sb, &a.scale[b], 8);
        if (!std::isnormal(a.scale[b]) || !std::isnormal(a.rinv[b]) || (sb & 0xfffffffffffffull) == 0xfffffffffffffull) a.slow_div = 1;
    }
    if (n_local > ((int64_t)1 << 31)) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "pca: more than 2^31 pixels per GPU");
    // ---- range of the fitted values: robust-normalised bands lie in [0,1] by construction (clip); any other input is
    // measured (one extra pass over the fitted pixels, extrema all-reduced), so that the fixed-point quantum below fits
    // whatever the caller passes (raw DN 0-255, reflectances, ...)
    double vmin[PCA_MAXB], vmax[PCA_MAXB];
    for (int b = 0; b < nb; b++) { vmin[b] = 0.0; vmax[b] = 1.0; }
    if (!a.normalise && u8) {
        for (int b = 0; b < nb; b++) { vmin[b] = 0.0; vmax[b] = 255.0; }   // any 8-bit plane
    } else if (!a.normalise) {
        const int rgrid = (int)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div64(fit_n, PCA_THREADS * 8)));
        RSCHK(ws_reserve(ctx, sizeof(float) * (size_t)rgrid * 2 * PCA_MAXB));
        RSCHK(pin_reserve(ctx, sizeof(float) * (size_t)rgrid * 2 * PCA_MAXB));
        double mm[2 * PCA_MAXB];
        for (int b = 0; b < nb; b++) mm[2 * b] = mm[2 * b + 1] = -INFINITY;
        if (fit_n > 0) {
            {
                prof_scope ps(ctx, "range");
                hipLaunchKernelGGL(k3_range, dim3(rgrid), dim3(PCA_THREADS), 0, ctx->stream, a, fit_off, fit_n, (float *)ctx->d_ws);
            }
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, ctx->d_ws, sizeof(float) * (size_t)rgrid * 2 * PCA_MAXB, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, rs_sync(ctx));
            const float *hp = (const float *)ctx->h_pin;
            for (int g = 0; g < rgrid; g++)
                for (int b = 0; b < nb; b++) {
                    mm[2 * b] = std::max(mm[2 * b], -(double)hp[(size_t)g * 2 * PCA_MAXB + 2 * b]);      // MAX-reduce of the negated minimum
                    mm[2 * b + 1] = std::max(mm[2 * b + 1], (double)hp[(size_t)g * 2 * PCA_MAXB + 2 * b + 1]);
                }
        }
        RSCHK(comm_allreduce_host(ctx, mm, 2 * nb, RSSEG_F64, RSSEG_MAX));
        for (int b = 0; b < nb; b++) {
            vmin[b] = -mm[2 * b];
            vmax[b] = mm[2 * b + 1];
            if (!(vmin[b] <= vmax[b])) vmin[b] = vmax[b] = 0.0;  // no finite value at all: the NaN check below reports it
            if (std::isinf(vmin[b]) || std::isinf(vmax[b])) return rs_fail(ctx, RSSEG_ERR_INVALID, "pca: Input X contains infinity");
        }
    }
    double bound = 1.0;
    for (int b = 0; b < nb; b++) {
        double c = a.scaled ? (double)a.center[b] : 0.0, s = a.scaled ? a.scale[b] : 1.0;
        bound = std::max(bound, std::max(std::fabs(vmin[b] - c), std::fabs(vmax[b] - c)) / std::fabs(s) * 1.000001);
    }
    int e2;
    std::frexp(bound * bound, &e2);           // bound^2 < 2^e2
    const int Q = std::min(38, 48 - e2);      // |x*x| * 2^Q < 2^48: 4096 pixels per thread stay below 2^60
    if (Q < -40) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "pca: scaled band range too large for exact accumulation (bound %.3g)", bound);
    a.fx_scale = std::ldexp(1.0, Q);

    // ---- Gram / mean over the fitted pixels [fit_off, fit_off + fit_n): a scalar head up to the first 16-byte
    // boundary, then the vectorised body
    struct seg { int64_t off, n; int grid; size_t poff; };
    seg segs[2];
    int nseg = 0;
    size_t pcount = 0;
    {
        const int64_t head = std::min<int64_t>(fit_n, (4 - (fit_off & 3)) & 3);
        if (head > 0) { segs[nseg++] = {fit_off, head, 1, pcount}; pcount += PCA_PSTRIDE; }
        if (fit_n - head > 0) {
            const int g = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ceil_div64((fit_n - head) >> 2, PCA_THREADS)));
            segs[nseg++] = {fit_off + head, fit_n - head, g, pcount};
            pcount += (size_t)g * PCA_PSTRIDE;
        }
    }
    RSCHK(ws_reserve(ctx, sizeof(long long) * (pcount + PCA_PSTRIDE)));
    RSCHK(pin_reserve(ctx, sizeof(long long) * PCA_PSTRIDE));
    long long *d_part = (long long *)ctx->d_ws, *d_tot = d_part + pcount;
    // float32 planes the select has just seen to hold only the integers 0..255: the table form first (the kernel verifies every
    // value, so a stale hint costs a pass, never a result)
    bool try_tab = !u8 && a.normalise;
    for (int b = 0; b < nb && try_tab; b++) {
        bool hit = false;
        for (const auto &h : ctx->byte_valued) hit = hit || (h.first == d_bands[b] && fit_off + fit_n <= h.second);
        try_tab = hit;
    }
    auto run_gram = [&](bool tab) -> int {
        for (int si = 0; si < nseg; si++) {
            pca_args as = a;
            for (int b = 0; b < nb; b++) as.band[b] = (const char *)a.band[b] + (size_t)segs[si].off * esz;
            prof_scope ps(ctx, "gram");
            switch (nb) {
#define GRAM_GO(NBV)                                                                                                                          \
    case NBV:                                                                                                                                 \
        if (u8) hipLaunchKernelGGL((k3_gram<NBV, true>), dim3(segs[si].grid), dim3(PCA_THREADS), 0, ctx->stream, as, segs[si].n, d_part + segs[si].poff);  \
        else if (tab) hipLaunchKernelGGL((k3_gram<NBV, false, true>), dim3(segs[si].grid), dim3(PCA_THREADS), 0, ctx->stream, as, segs[si].n, d_part + segs[si].poff);  \
        else hipLaunchKernelGGL((k3_gram<NBV, false>), dim3(segs[si].grid), dim3(PCA_THREADS), 0, ctx->stream, as, segs[si].n, d_part + segs[si].poff);   \
        break;
                GRAM_GO(1) GRAM_GO(2) GRAM_GO(3) GRAM_GO(4) GRAM_GO(5) GRAM_GO(6) GRAM_GO(7) GRAM_GO(8)
#undef GRAM_GO
            }
        }
        HIPCHK(ctx, hipGetLastError());
        if (pcount) {
            hipLaunchKernelGGL(k3_reduce_partials, dim3(PCA_PSTRIDE), dim3(256), 0, ctx->stream, (const long long *)d_part, (int)(pcount / PCA_PSTRIDE), d_tot);
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_tot, sizeof(long long) * PCA_PSTRIDE, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, rs_sync(ctx));
        }
        return RSSEG_OK;
    };
    RSCHK(run_gram(try_tab));
    if (try_tab && pcount && ((const long long *)ctx->h_pin)[2 * PCA_NACC + 1] != 0) RSCHK(run_gram(false));   // some value was not an integer 0..255
    typedef __int128 i128;
    long long lim[2 * PCA_NACC + 2];
    {
        const long long *hp = (const long long *)ctx->h_pin;
        for (int i = 0; i < PCA_NACC; i++) {
            const i128 s = pcount ? ((i128)hp[2 * i] << 32) + (i128)hp[2 * i + 1] : (i128)0;
            lim[2 * i] = (long long)(s >> 32);
            lim[2 * i + 1] = (long long)(s & 0xffffffffLL);
        }
        lim[2 * PCA_NACC] = fit_n;
        lim[2 * PCA_NACC + 1] = pcount ? hp[2 * PCA_NACC] : 0;   // (slot 2 * PCA_NACC + 1 of the table: the try-pass's misses, zero here)
    }
    RSCHK(comm_allreduce_host(ctx, lim, 2 * PCA_NACC + 2, RSSEG_I64, RSSEG_SUM));
    if (lim[2 * PCA_NACC + 1] != 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "pca: Input X contains NaN.");
    const int64_t N = lim[2 * PCA_NACC];
    if (N < 2) return rs_fail(ctx, RSSEG_ERR_INVALID, "pca: needs at least 2 samples");
    const double inv = std::ldexp(1.0, -Q);
    auto val = [&](int i) { return (double)((((i128)lim[2 * i]) << 32) + (i128)lim[2 * i + 1]) * inv; };
    // sklearn's float32 steps on exactly accumulated sums
    float mu[PCA_MAXB];
    for (int b = 0; b < nb; b++) {
        volatile float s = (float)val(b);
        volatile float m = s / (float)N;
        mu[b] = m;
    }
    double Cd[PCA_MAXB][PCA_MAXB];
    {
        int t = PCA_MAXB;
        for (int b = 0; b < PCA_MAXB; b++)
            for (int c = b; c < PCA_MAXB; c++, t++) {
                if (b >= nb || c >= nb) continue;
                volatile float g = (float)val(t);                // X.T @ X
                volatile float nm = (float)N * mu[b];            // n_samples * mean (column)
                volatile float nmm = nm * mu[c];                 //   ... * mean (row)
                volatile float cc = g - nmm;                     // C -= ...
                volatile float cv = cc / (float)(N - 1);         // C /= n_samples - 1
                Cd[b][c] = Cd[c][b] = (double)cv;
            }
    }
    double V[PCA_MAXB][PCA_MAXB], w[PCA_MAXB];
    jacobi_eigh(nb, Cd, V, w);
    int order[PCA_MAXB];
    for (int i = 0; i < nb; i++) order[i] = i;
    std::stable_sort(order, order + nb, [&](int x, int y) { return w[x] > w[y]; });
    float ev[PCA_MAXB], total = 0.f;
    for (int i = 0; i < nb; i++) {
        float e = (float)w[order[i]];
        ev[i] = e < 0.f ? 0.f : e;
    }
    {
        volatile float t = 0.f;
        for (int i = 0; i < nb; i++) t = t + ev[i];
        total = t;
    }
    proj_args pr;
    memset(&pr, 0, sizeof(pr));
    pr.nc = n_components;
    for (int c = 0; c < n_components; c++) {
        float row[PCA_MAXB];
        int am = 0;
        for (int b = 0; b < nb; b++) {
            row[b] = (float)V[b][order[c]];
            if (std::fabs(row[b]) > std::fabs(row[am])) am = b;
        }
        const float sg = row[am] < 0.f ? -1.f : 1.f;  // svd_flip(u_based_decision=False)
        volatile float o = 0.f;
        for (int b = 0; b < nb; b++) {
            pr.comp[c][b] = row[b] * sg;
            o = fmaf(mu[b], pr.comp[c][b], o);
            if (components) components[c * nb + b] = pr.comp[c][b];
        }
        pr.offs[c] = o;
        if (explained_variance) explained_variance[c] = ev[c];
        if (explained_variance_ratio) explained_variance_ratio[c] = ev[c] / total;
        if (d_out) {
            if (!d_out[c] || ((uintptr_t)d_out[c] & 15)) return rs_fail(ctx, RSSEG_ERR_INVALID, "pca: output plane %d null or unaligned", c);
            pr.out[c] = d_out[c];
        }
    }
    if (mean)
        for (int b = 0; b < nb; b++) mean[b] = mu[b];
    if (d_out && n_local > 0 && fuse) {
        fuse_args fz;
        memset(&fz, 0, sizeof(fz));
        for (int j = 0; j < 7; j++) fz.idx[j] = fuse->d_idx[j];
        for (int j = 0; j < 5; j++) fz.norm[j] = fuse->d_norm ? fuse->d_norm[j] : nullptr;
        fz.evi.L = fuse->evi_coef ? fuse->evi_coef[0] : 1.0f;
        fz.evi.C1 = fuse->evi_coef ? fuse->evi_coef[1] : 6.0f;
        fz.evi.C2 = fuse->evi_coef ? fuse->evi_coef[2] : 7.5f;
        fz.evi.G = fuse->evi_coef ? fuse->evi_coef[3] : 2.5f;
        fz.q = fuse->d_q;
        fz.q_lo = fuse->q_lo;
        fz.q_hi = fuse->q_hi;
        fz.q_den = norm_den(fuse->q_lo, fuse->q_hi);
        fz.q_mult = fuse->q_mult;
        RSCHK(mm_begin(ctx, 7 + n_components));
        {
            prof_scope ps(ctx, "indices_project");
            // large planes: one contiguous chunk of 16 tiles per workgroup (up to 16384 workgroups); small ones (fewer than 1024
            // chunks): 16-byte vectors dealt grid-stride to up to 2048 workgroups, as before.  profiles/r04_fuse_sweep.json:
            // 3.86 ms chunked against 4.07-4.22 ms grid-strided at 16384^2 on the same box
            int chunked = ceil_div64(n_local >> 2, FUSE_CHUNK_V) >= 1024 ? 1 : 0;
            if (const char *e = getenv("RSSEG_FUSE_MAP")) chunked = atoi(e);                  // experiments (profiles/r04_fuse_sweep.py)
            int64_t fuse_cap = chunked ? 16384 : 2048;
            if (const char *e = getenv("RSSEG_FUSE_GRID")) fuse_cap = std::max(1, atoi(e));
            const int64_t units = chunked ? ceil_div64(n_local >> 2, FUSE_CHUNK_V) : ceil_div64(n_local >> 2, PCA_THREADS);
            const dim3 pg((int)std::min<int64_t>(fuse_cap, std::max<int64_t>(1, units)));
            switch (nb) {
#define FUSE_GO(NBV)                                                                                                                   \
    case NBV:                                                                                                                          \
        if (u8) {                                                                                                                      \
            if (ctx->mm_collect) hipLaunchKernelGGL((k3_indices_project<NBV, true, true>), pg, dim3(PCA_THREADS), 0, ctx->stream, a, pr, fz, n_local, ctx->d_mm, chunked); \
            else hipLaunchKernelGGL((k3_indices_project<NBV, false, true>), pg, dim3(PCA_THREADS), 0, ctx->stream, a, pr, fz, n_local, (uint32_t *)nullptr, chunked);      \
        } else {                                                                                                                       \
            if (ctx->mm_collect) hipLaunchKernelGGL((k3_indices_project<NBV, true, false>), pg, dim3(PCA_THREADS), 0, ctx->stream, a, pr, fz, n_local, ctx->d_mm, chunked); \
            else hipLaunchKernelGGL((k3_indices_project<NBV, false, false>), pg, dim3(PCA_THREADS), 0, ctx->stream, a, pr, fz, n_local, (uint32_t *)nullptr, chunked);      \
        }                                                                                                                              \
        break;
                FUSE_GO(5) FUSE_GO(6) FUSE_GO(7) FUSE_GO(8)
#undef FUSE_GO
            default: return rs_fail(ctx, RSSEG_ERR_INVALID, "indices_pca: %d bands", nb);
            }
        }
        HIPCHK(ctx, hipGetLastError());
        RSCHK(mm_end(ctx, 7 + n_components));
    } else if (d_out && n_local > 0) {
        RSCHK(mm_begin(ctx, n_components));
        {
            prof_scope ps(ctx, "project");
            const dim3 pg((int)std::min<int64_t>(2048, std::max<int64_t>(1, ceil_div64(n_local >> 2, PCA_THREADS))));
            switch (nb) {
#define PROJ_GO(NBV)                                                                                                               \
    case NBV:                                                                                                                      \
        if (u8) {                                                                                                                  \
            if (ctx->mm_collect) hipLaunchKernelGGL((k3_project<NBV, true, true>), pg, dim3(PCA_THREADS), 0, ctx->stream, a, pr, n_local, ctx->d_mm); \
            else hipLaunchKernelGGL((k3_project<NBV, false, true>), pg, dim3(PCA_THREADS), 0, ctx->stream, a, pr, n_local, (uint32_t *)nullptr);      \
        } else {                                                                                                                   \
            if (ctx->mm_collect) hipLaunchKernelGGL((k3_project<NBV, true, false>), pg, dim3(PCA_THREADS), 0, ctx->stream, a, pr, n_local, ctx->d_mm); \
            else hipLaunchKernelGGL((k3_project<NBV, false, false>), pg, dim3(PCA_THREADS), 0, ctx->stream, a, pr, n_local, (uint32_t *)nullptr);      \
        }                                                                                                                          \
        break;
                PROJ_GO(1) PROJ_GO(2) PROJ_GO(3) PROJ_GO(4) PROJ_GO(5) PROJ_GO(6) PROJ_GO(7) PROJ_GO(8)
#undef PROJ_GO
            }
        }
        HIPCHK(ctx, hipGetLastError());
        RSCHK(mm_end(ctx, n_components));
    }
    return ctx->mm_collect ? RSSEG_OK : stream_sync(ctx);   // the extrema read-back has already waited for the stream
}
