// K9/K10 — MinMax scaling, k-means++ seeding and Lloyd iterations on feature-planar rasters.
//
// Replaces unsupervised_kmeans_classification (reference modules/features/extract.py:568-579), i.e.
// sklearn 1.7.2 MinMaxScaler + KMeans(random_state, n_init=1, init='k-means++').fit_predict:
//   sklearn/preprocessing/_data.py:508-522,555-556   sklearn/cluster/_kmeans.py:213-287,699-748,1454-1530
//   sklearn/cluster/_k_means_lloyd.pyx:29-218        sklearn/cluster/_k_means_common.pyx:167-311
//   sklearn/metrics/pairwise.py:391-437,582-644
//
// Numerics contract (identical to oracle/kmeans_impl.h, which tests compare against bit for bit):
//   * the value clustered is fl(fl(fl(x*scale)+min_) - mean) in the input type T, recomputed on the
//     fly from the raw planes — the scaled matrix is never materialised in HBM;
//   * per-pixel arithmetic keeps sklearn's order: Lloyd distances are fma chains in T
//     (||c||^2 - 2 x.c), k-means++ distances are the float64 "upcast" formula rounded to T;
//   * every reduction over pixels (means, variances, potentials, per-cluster sums) is an exact
//     fixed-point sum (quantum 2^-40), so results do not depend on the launch geometry, on atomics
//     or on how the raster is sharded over GPUs.
//
// Data movement per Lloyd iteration: F planes read once (4F B/px for float32) + 1 B/px label read
// + 1 B/px label write; a lane keeps the F values of its pixels in registers from one batch of loads, so the
// per-cluster accumulation (LDS atomics on bank-private copies) needs no second read.  HBM-bound.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <mutex>
#include <type_traits>

#include "common.h"

typedef unsigned __int128 u128;
typedef __int128 i128;

#define KM_THREADS 256
#define KM_TILES_PER_CHUNK 16

template <typename T> struct vt;
template <> struct vt<float> {
    typedef float4 vec;
    static constexpr int PXL = 4;
};
template <> struct vt<double> {
    typedef double2 vec;
    static constexpr int PXL = 2;
};
template <typename T> __host__ __device__ constexpr int km_tile() { return KM_THREADS * vt<T>::PXL; }
template <typename T> __host__ __device__ constexpr int km_chunk() { return km_tile<T>() * KM_TILES_PER_CHUNK; }

struct planes_t {
    const void *p[RSSEG_MAX_FEATURES];
};

// per-feature scaler parameters, in T, resident in device memory (uniform loads)
template <typename T> struct scaler_t {
    T scale[RSSEG_MAX_FEATURES];
    T minv[RSSEG_MAX_FEATURES];
    T mean[RSSEG_MAX_FEATURES];
};

template <typename T> __device__ __forceinline__ T tfma(T a, T b, T c);
template <> __device__ __forceinline__ float tfma<float>(float a, float b, float c) { return __fmaf_rn(a, b, c); }
template <> __device__ __forceinline__ double tfma<double>(double a, double b, double c) { return fma(a, b, c); }

template <typename T> __device__ __forceinline__ void unpack(const typename vt<T>::vec &v, T *o);
template <> __device__ __forceinline__ void unpack<float>(const float4 &v, float *o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void unpack<double>(const double2 &v, double *o) { o[0] = v.x; o[1] = v.y; }

// loads PXL consecutive pixels of plane f starting at pixel `base` (bounds-checked against n)
template <typename T>
__device__ __forceinline__ void load_px(const void *plane, int64_t base, int64_t n, T *o)
{
    constexpr int PXL = vt<T>::PXL;
    const T *p = reinterpret_cast<const T *>(plane);
    if (base + PXL <= n) {
        typename vt<T>::vec v = *reinterpret_cast<const typename vt<T>::vec *>(p + base);
        unpack<T>(v, o);
    } else {
#pragma unroll
        for (int i = 0; i < PXL; i++) o[i] = (base + i < n) ? p[base + i] : (T)0;
    }
}

// FULL: the whole tile lies inside the plane, so the 16-byte load needs no bounds handling
template <typename T, bool FULL>
__device__ __forceinline__ void load_pxf(const void *plane, int64_t base, int64_t n, T *o)
{
    if constexpr (FULL) {
        const T *p = reinterpret_cast<const T *>(plane);
        typedef T ntvec __attribute__((ext_vector_type(vt<T>::PXL)));
        // non-temporal: a sweep streams 16 GB of planes once; nothing is read again before it would be evicted (measured on one
        // box against plain loads: km_moment 2.83 -> 2.65 ms, --data hard 173.1 -> 171.1 ms per step)
        const ntvec v = __builtin_nontemporal_load(reinterpret_cast<const ntvec *>(p + base));
#pragma unroll
        for (int i = 0; i < vt<T>::PXL; i++) o[i] = v[i];
    } else {
        if (base < n) load_px<T>(plane, base, n, o);
        else {
#pragma unroll
            for (int i = 0; i < vt<T>::PXL; i++) o[i] = (T)0;
        }
    }
}

template <typename T> __device__ __forceinline__ T scaled(T x, T sc, T mn)
{
    if (x != x) x = (T)0;  // extract.py:548-556  NaN -> 0
    T xs = x * sc;
    return xs + mn;
}

// ------------------------------------------------------------------------------------------------
// pass A: per-feature min / max (NaN -> 0).  grid (nblk, F); partial[f][blk] = {min, max}
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void km_minmax(planes_t pl, int64_t n, T *__restrict__ pmin, T *__restrict__ pmax)
{
    constexpr int PXL = vt<T>::PXL;
    const int f = blockIdx.y;
    T mn = (T)INFINITY, mx = (T)-INFINITY;
    for (int64_t base = ((int64_t)blockIdx.x * KM_THREADS + threadIdx.x) * PXL; base < n; base += (int64_t)gridDim.x * KM_THREADS * PXL) {
        T v[PXL];
        load_px<T>(pl.p[f], base, n, v);
#pragma unroll
        for (int i = 0; i < PXL; i++)
            if (base + i < n) {
                T x = v[i];
                if (x != x) x = (T)0;
                mn = x < mn ? x : mn;
                mx = x > mx ? x : mx;
            }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    __shared__ T smn[4], smx[4];
    if (lane_id() == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { mn = smn[w] < mn ? smn[w] : mn; mx = smx[w] > mx ? smx[w] : mx; }
        pmin[(size_t)f * gridDim.x + blockIdx.x] = mn;
        pmax[(size_t)f * gridDim.x + blockIdx.x] = mx;
    }
}

// ------------------------------------------------------------------------------------------------
// mean pass: exact sums of fixed(xs) (the np.var numerators ride on the first k-means++ pass, km_kpp MODE 0).
// grid (nblk, F); partial[f][blk] int64
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void km_moment(planes_t pl, int64_t n, const scaler_t<T> *__restrict__ sp,
                                                        long long *__restrict__ partial)
{
    constexpr int PXL = vt<T>::PXL;
    const int f = blockIdx.y;
    const T sc = sp->scale[f], mnv = sp->minv[f];
    long long acc = 0;
    constexpr int UNR = 4;  // vector loads in flight per lane
    const int64_t stride = (int64_t)gridDim.x * KM_THREADS * PXL;
    int64_t base = ((int64_t)blockIdx.x * KM_THREADS + threadIdx.x) * PXL;
    for (; base + (UNR - 1) * stride + PXL <= n; base += UNR * stride) {
        T v[UNR][PXL];
#pragma unroll
        for (int u = 0; u < UNR; u++) load_pxf<T, true>(pl.p[f], base + u * stride, n, v[u]);
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
            for (int i = 0; i < PXL; i++) acc += to_fixed40((double)scaled<T>(v[u][i], sc, mnv));
    }
    for (; base < n; base += stride) {
        T v[PXL];
        load_px<T>(pl.p[f], base, n, v);
#pragma unroll
        for (int i = 0; i < PXL; i++)
            if (base + i < n) acc += to_fixed40((double)scaled<T>(v[i], sc, mnv));
    }
    acc = wave_sum(acc);
    __shared__ long long sacc[4];
    if (lane_id() == 0) sacc[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)f * gridDim.x + blockIdx.x] = sacc[0] + sacc[1] + sacc[2] + sacc[3];
}

// ------------------------------------------------------------------------------------------------
// gather one scaled+centred row (F values of T) at local pixel index idx
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void km_gather_row(planes_t pl, int F, int64_t idx, const scaler_t<T> *__restrict__ sp, T *__restrict__ out)
{
    const int f = threadIdx.x;
    if (f < F) {
        T x = reinterpret_cast<const T *>(pl.p[f])[idx];
        out[f] = scaled<T>(x, sp->scale[f], sp->minv[f]) - sp->mean[f];
    }
}

// ------------------------------------------------------------------------------------------------
// k-means++ pass (_kmeans.py:225-262).  candT: [F][KPP_STRIDE] rows upcast to float64 (transposed: one scalar
// load per feature brings all of them): columns 0..L-1 are this round's candidates, column KPP_MAXL is the
// PENDING centre (the one chosen in the previous round); cc: their float64 squared norms, same layout.
//   MODE 0 (first centre):  partial[0][chunk] = sum fixed(d2(cand0, x)), and the np.var numerators
//                           partial[1 + f][chunk] = sum fixed(fl((xs-m)*(xs-m))) ride on the same sweep;
//                           nothing is written per pixel.
//   MODE 1 (round 1):       closest <- d2(pending, x)
//   MODE 2 (later rounds):  closest <- min(closest, d2(pending, x))            (np.minimum of the chosen candidate)
//   MODE 1/2 then:          partial[l][chunk] = sum fixed(min(closest, d2(cand_l, x)))   for l < L
// One workgroup per chunk (km_chunk<T>() pixels), so partial[l][] is at once the potential of candidate l and,
// for the candidate the host picks, the prefix table used to locate the next sampled pixel (np.searchsorted on
// stable_cumsum, _kmeans.py:243-246): the chosen candidate's min-plane is never stored, the next round
// re-derives it as its pending update (same arithmetic, same bits) and km_kpp_chunkq does so for the one chunk a
// sample falls into.  Traffic per round: 4F + 8 B/px (read F planes + closest, write closest).
// As in km_lloyd the F feature vectors of a lane's pixels are requested back to back into registers.
// ------------------------------------------------------------------------------------------------
#define KPP_MAXL 8
#define KPP_STRIDE (KPP_MAXL + 1)

template <typename T, int NL, int FR, int MODE>
__global__ __launch_bounds__(KM_THREADS) void km_kpp(planes_t pl, int F, int64_t n, const scaler_t<T> *__restrict__ sp,
                                                     const double *__restrict__ candT, const double *__restrict__ cc, int L,
                                                     T *__restrict__ closest, unsigned long long *__restrict__ partial, int64_t nchunks)
{
    constexpr int PXL = vt<T>::PXL;
    constexpr bool VAR = MODE == 0;
    unsigned long long acc[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) acc[l] = 0;
    long long vacc[VAR ? FR : 1];
#pragma unroll
    for (int f = 0; f < (VAR ? FR : 1); f++) vacc[f] = 0;
    const int64_t chunk0 = (int64_t)blockIdx.x * km_chunk<T>();
    auto tile_body = [&](auto full_t, int64_t base) {
        constexpr bool FULL = decltype(full_t)::value;
        T x[FR][PXL];
#pragma unroll
        for (int f = 0; f < FR; f++) {
            if (f < F) load_pxf<T, FULL>(pl.p[f], base, n, x[f]);
            else {
#pragma unroll
                for (int p = 0; p < PXL; p++) x[f][p] = (T)0;
            }
        }
        T cl[PXL];
#pragma unroll
        for (int p = 0; p < PXL; p++) cl[p] = (T)0;
        if constexpr (MODE == 2) load_pxf<T, FULL>(closest, base, n, cl);
        double dot[NL][PXL], dotp[PXL], yy[PXL];
#pragma unroll
        for (int p = 0; p < PXL; p++) {
            yy[p] = 0.0;
            dotp[p] = 0.0;
#pragma unroll
            for (int l = 0; l < NL; l++) dot[l][p] = 0.0;
        }
#pragma unroll
        for (int f = 0; f < FR; f++) {
            if (f < F) {
                const T sc = sp->scale[f], mnv = sp->minv[f], me = sp->mean[f];
                double cd[NL];  // unconditional: candT rows are padded (zeros beyond L)
#pragma unroll
                for (int l = 0; l < NL; l++) cd[l] = candT[f * KPP_STRIDE + l];
                const double cdp = candT[f * KPP_STRIDE + KPP_MAXL];
#pragma unroll
                for (int p = 0; p < PXL; p++) {
                    const T yt = scaled<T>(x[f][p], sc, mnv) - me;
                    if constexpr (VAR) {
                        const T dd = yt * yt;
                        if (FULL || base + p < n) vacc[f] += to_fixed40((double)dd);
                    }
                    const double y = (double)yt;
                    yy[p] = fma(y, y, yy[p]);
                    if constexpr (MODE != 0) dotp[p] = fma(cdp, y, dotp[p]);
#pragma unroll
                    for (int l = 0; l < NL; l++) dot[l][p] = fma(cd[l], y, dot[l][p]);
                }
            }
        }
        if constexpr (MODE != 0) {
            const double ccp = cc[KPP_MAXL];
            bool moved = MODE != 2;   // MODE 2: did the pending centre come closer to any of the thread's pixels?
#pragma unroll
            for (int p = 0; p < PXL; p++) {
                double d = -2.0 * dotp[p];
                d = d + ccp;
                d = d + yy[p];
                T dt = (T)d;
                dt = dt > (T)0 ? dt : (T)0;
                if constexpr (MODE == 2) {
                    moved = moved || dt < cl[p];
                    dt = cl[p] < dt ? cl[p] : dt;
                }
                cl[p] = dt;
            }
            if (FULL || base + PXL <= n) {
                // the closest-distance plane is rewritten only where it changed (a new centre takes over ~1 / (r + 1) of the pixels
                // in round r): less of the write stream that costs a sweep ~10 % of its read rate (profiles/r02_streams.json)
                if (moved) {
                    typename vt<T>::vec o;
                    if constexpr (PXL == 4) o = make_float4(cl[0], cl[1], cl[2], cl[3]);
                    else o = make_double2(cl[0], cl[1]);
                    *reinterpret_cast<typename vt<T>::vec *>(closest + base) = o;
                }
            } else {
                for (int p = 0; p < PXL; p++)
                    if (base + p < n) closest[base + p] = cl[p];
            }
        }
#pragma unroll
        for (int l = 0; l < NL; l++) {
            if (l < L) {
                const double ccl = cc[l];
#pragma unroll
                for (int p = 0; p < PXL; p++) {
                    double d = -2.0 * dot[l][p];
                    d = d + ccl;
                    d = d + yy[p];
                    T dt = (T)d;
                    dt = dt > (T)0 ? dt : (T)0;                               // np.maximum(distances, 0)
                    if constexpr (MODE != 0) dt = cl[p] < dt ? cl[p] : dt;    // np.minimum(closest, d)
                    if (FULL || base + p < n) acc[l] += (unsigned long long)to_fixed40((double)dt);
                }
            }
        }
    };
    for (int t = 0; t < KM_TILES_PER_CHUNK; t++) {
        const int64_t tbase = chunk0 + (int64_t)t * km_tile<T>();
        if (tbase >= n) break;
        const int64_t base = tbase + (int64_t)threadIdx.x * PXL;
        if (tbase + km_tile<T>() <= n) tile_body(std::true_type{}, base);
        else tile_body(std::false_type{}, base);
    }
    __shared__ unsigned long long sacc[4][NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
        unsigned long long s = wave_sum(acc[l]);
        if (lane_id() == 0) sacc[threadIdx.x >> 6][l] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < NL && (int)threadIdx.x < L)
        partial[(size_t)threadIdx.x * nchunks + blockIdx.x] =
            sacc[0][threadIdx.x] + sacc[1][threadIdx.x] + sacc[2][threadIdx.x] + sacc[3][threadIdx.x];
    if constexpr (VAR) {
        __shared__ long long svar[4][FR];
#pragma unroll
        for (int f = 0; f < FR; f++) {
            long long s = wave_sum(vacc[f]);
            if (lane_id() == 0) svar[threadIdx.x >> 6][f] = s;
        }
        __syncthreads();
        if ((int)threadIdx.x < F)
            partial[(size_t)(1 + threadIdx.x) * nchunks + blockIdx.x] =
                (unsigned long long)(svar[0][threadIdx.x] + svar[1][threadIdx.x] + svar[2][threadIdx.x] + svar[3][threadIdx.x]);
    }
}

// Sampling step of a k-means++ round on the device (np.searchsorted(stable_cumsum(closest), r), _kmeans.py:243-246, for the
// chunk the host has already located from the chunk sums).  One workgroup per candidate:
//   mode 1: km_kpp_chunkq re-derives the current closest-distance values of the chunk [c0, c0 + cn) — what the next km_kpp
//           round will store for it: with_old ? min(closest, d2(pending, x)) : d2(pending, x), same operation sequence as
//           km_kpp, so the same bits — as fixed-point integers; km_kpp_sample finds the first pixel whose running sum
//           reaches `rem` (the target minus the sum of everything before the chunk) with a two-level search;
//   mode 2: the pixel is given (np.clip of an index past the end);   mode 0: another rank owns this candidate.
// Writes out[l] = {global index as double, the F scaled+centred values of that pixel as double}; one device-to-host copy
// then serves all candidates of the round (instead of a chunk copy and a row copy per candidate).
struct kpp_sample_args {
    int64_t c0[KPP_MAXL], cn[KPP_MAXL], direct[KPP_MAXL];
    unsigned long long rem[KPP_MAXL];
    int mode[KPP_MAXL];
    int with_old;
    int64_t offset;
};
// step 1 (grid: chunk / 256 x L): the fixed-point images of the current closest distances of candidate l's chunk
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void km_kpp_chunkq(planes_t pl, int F, const scaler_t<T> *__restrict__ sp,
                                                            const double *__restrict__ candT, const double *__restrict__ cc,
                                                            const T *__restrict__ closest, const kpp_sample_args *__restrict__ ap,
                                                            unsigned long long *__restrict__ qv)
{
    const kpp_sample_args &a = *ap;
    const int l = blockIdx.y;
    if (a.mode[l] != 1) return;
    const int64_t c0 = a.c0[l], cn = a.cn[l];
    const int64_t i = (int64_t)blockIdx.x * KM_THREADS + threadIdx.x;
    if (i >= cn) return;
    const double ccp = cc[KPP_MAXL];
    double yy = 0.0, dotp = 0.0;
    for (int f = 0; f < F; f++) {
        const T xv = reinterpret_cast<const T *>(pl.p[f])[c0 + i];
        const T yt = scaled<T>(xv, sp->scale[f], sp->minv[f]) - sp->mean[f];
        const double y = (double)yt;
        yy = fma(y, y, yy);
        dotp = fma(candT[f * KPP_STRIDE + KPP_MAXL], y, dotp);
    }
    double d = -2.0 * dotp;
    d = d + ccp;
    d = d + yy;
    T dt = (T)d;
    dt = dt > (T)0 ? dt : (T)0;
    if (a.with_old) {
        const T c = closest[c0 + i];
        dt = c < dt ? c : dt;
    }
    qv[(size_t)l * km_chunk<T>() + i] = (unsigned long long)to_fixed40((double)dt);
}

// step 2 (one workgroup per candidate): two-level search over the exact prefix sums, then the row of the pixel found
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void km_kpp_sample(planes_t pl, int F, const scaler_t<T> *__restrict__ sp,
                                                            const unsigned long long *__restrict__ qv, const kpp_sample_args *__restrict__ ap,
                                                            double *__restrict__ out)
{
    const kpp_sample_args &a = *ap;
    const int l = blockIdx.x;
    double *o = out + (size_t)l * (1 + RSSEG_MAX_FEATURES);
    if (a.mode[l] == 0) return;
    constexpr int ROWS = (int)(km_chunk<T>() / KM_THREADS);  // rows of 256 consecutive pixels
    __shared__ unsigned long long rowsum[ROWS];
    __shared__ unsigned long long rowq[KM_THREADS];
    __shared__ long long s_li;
    __shared__ int s_row;
    __shared__ unsigned long long s_rem;
    const int64_t c0 = a.c0[l], cn = a.cn[l];
    if (a.mode[l] == 1) {
        const unsigned long long *q = qv + (size_t)l * km_chunk<T>();
        const int wv = threadIdx.x >> 6, ln = lane_id();
        static_assert(ROWS % 4 == 0, "rows are dealt to the four waves");
        // row sums: wave w takes rows w, w + 4, ...; a lane's loads are independent and issued together
#pragma unroll
        for (int rr = 0; rr < ROWS / 4; rr++) {
            const int r = rr * 4 + wv;
            unsigned long long v = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t i = (int64_t)r * KM_THREADS + j * 64 + ln;
                v += i < cn ? q[i] : 0ull;
            }
            v = wave_sum(v);
            if (ln == 0) rowsum[r] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long run = 0;
            int r = 0;
            for (; r < ROWS; r++) {
                if (run + rowsum[r] >= a.rem[l]) break;
                run += rowsum[r];
            }
            s_row = r;                 // ROWS: never reached (cannot happen: the chunk sum reaches the target)
            s_rem = a.rem[l] - run;    // what the pixels of row r still have to cover
            s_li = cn - 1;             // np.clip: nothing reaches the target (cannot happen) -> the chunk's last pixel
        }
        __syncthreads();
        if (s_row < ROWS) {
            // first pixel of the row whose inclusive running sum reaches s_rem: wave-level scans + the waves' totals
            const int64_t i = (int64_t)s_row * KM_THREADS + threadIdx.x;
            const unsigned long long v = i < cn ? q[i] : 0ull;
            unsigned long long pfx = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long t = __shfl_up(pfx, o, 64);
                if (ln >= o) pfx += t;
            }
            if (ln == 63) rowq[wv] = pfx;
            __syncthreads();
            unsigned long long base = 0;
            for (int w2 = 0; w2 < wv; w2++) base += rowq[w2];
            const bool hit = i < cn && base + pfx >= s_rem;
            const unsigned long long m = __ballot(hit);
            if (m && ln == (int)__builtin_ctzll(m)) atomicMin((unsigned long long *)&s_li, (unsigned long long)i);
        }
        __syncthreads();
    } else {
        if (threadIdx.x == 0) s_li = a.direct[l] - c0;
        __syncthreads();
    }
    const int64_t idx = c0 + s_li;  // local pixel index
    if (threadIdx.x == 0) o[0] = (double)(a.offset + idx);
    if ((int)threadIdx.x < F) {
        const int f = threadIdx.x;
        const T x = reinterpret_cast<const T *>(pl.p[f])[idx];
        o[1 + f] = (double)(scaled<T>(x, sp->scale[f], sp->minv[f]) - sp->mean[f]);
    }
}

// ------------------------------------------------------------------------------------------------
// Lloyd iteration (lloyd_iter_chunked_dense + _update_chunk_dense).
// cenT: [F][KMAX] centres (transposed so one scalar load brings the KMAX values of a feature),
// csq: [KMAX].  labels: uint8 per pixel.  partial (UPDATE only): [(KMAX*F + KMAX + 1)][nchunks]
// int64: per-cluster fixed-point sums, per-cluster counts, number of changed labels.
//
// One workgroup walks one chunk (16 tiles).  Per tile each lane owns PXL consecutive pixels:
//   load     all F feature vectors of the lane's pixels are requested back to back (F x 16 B per
//            lane in flight: a wave keeps F KiB of HBM reads outstanding, which is what hides the
//            latency at the 3 workgroups per CU the register budget allows) and stay in REGISTERS
//            for the whole tile — the per-cluster accumulation needs them again after the argmin;
//   phase A  scale/centre, accumulate the k dot products (fma chain over f, as sklearn's gemm);
//   argmin   strict '<' (lowest index wins), label written as uint8;
//   phase B  the 2^-40 fixed-point images of the lane's values are added to per-cluster 64-bit LDS
//            accumulators.  The accumulators are replicated 32 times (copy = lane & 31, copy stride
//            = 2 banks mod 64) so that the 64 lanes of a ds_add_u64 never collide on a bank when
//            they carry the same label — the common case, labels being spatially coherent.
//            Integer adds commute, so the result is independent of any order.
// LDS: 32 accumulator copies only (33 KB at KMAX = 8, F <= 15).
// ------------------------------------------------------------------------------------------------
static_assert(RSSEG_MAX_FEATURES <= 64, "km_gather_row / km_kpp_sample give one of their first 64 threads to each feature");
#define KM_COPIES 32
__host__ __device__ inline int km_copy_stride(int KMAX, int F)
{
    // in 8-byte words; an odd stride puts copy c on banks {2c, 2c+1} (mod 64)
    return (KMAX * F + KMAX) | 1;
}

template <typename T, int KMAX, int FR, bool UPDATE>
__global__ __launch_bounds__(KM_THREADS) void km_lloyd(planes_t pl, int F, int k, int64_t n,
                                                       const scaler_t<T> *__restrict__ sp, const T *__restrict__ cenT,
                                                       const T *__restrict__ csq, uint8_t *__restrict__ labels,
                                                       long long *__restrict__ partial, int64_t nchunks, int ncopies,
                                                       const int *__restrict__ done, int32_t *__restrict__ out32)
{
    // out32 (final E-step only, !UPDATE): the caller's int32 label plane is written directly and the uint8 working plane
    // is neither read nor written (r04; before, a separate kernel widened the uint8 plane afterwards)
    if (done && *done) return;   // a speculatively enqueued iteration behind the one that converged (kl_update)
    constexpr int PXL = vt<T>::PXL;
    constexpr int TILE = KM_THREADS * PXL;
    extern __shared__ __align__(16) char smem[];
    unsigned long long *S = reinterpret_cast<unsigned long long *>(smem);  // [ncopies][stride]
    const int stride = km_copy_stride(KMAX, F);
    const int lane = threadIdx.x & 63;
    unsigned long long *myS = S + (size_t)(lane & (ncopies - 1)) * stride;
    if (UPDATE) {
        for (int i = threadIdx.x; i < ncopies * stride; i += KM_THREADS) S[i] = 0ull;
        __syncthreads();
    }
    T cs[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; j++) cs[j] = csq[j];
    const int64_t chunk0 = (int64_t)blockIdx.x * km_chunk<T>();
    int my_changed = 0;
    auto tile_body = [&](auto full_t, int64_t base) {
        constexpr bool FULL = decltype(full_t)::value;
        T x[FR][PXL];
#pragma unroll
        for (int f = 0; f < FR; f++) {
            if (f < F) load_pxf<T, FULL>(pl.p[f], base, n, x[f]);
            else {
#pragma unroll
                for (int p = 0; p < PXL; p++) x[f][p] = (T)0;
            }
        }
        T acc[KMAX][PXL];
#pragma unroll
        for (int j = 0; j < KMAX; j++)
#pragma unroll
            for (int p = 0; p < PXL; p++) acc[j][p] = (T)0;
#pragma unroll
        for (int f = 0; f < FR; f++) {
            if (f < F) {
                const T sc = sp->scale[f], mnv = sp->minv[f], me = sp->mean[f];
                T cj[KMAX];
#pragma unroll
                for (int j = 0; j < KMAX; j++) cj[j] = cenT[f * KMAX + j];
#pragma unroll
                for (int p = 0; p < PXL; p++) x[f][p] = scaled<T>(x[f][p], sc, mnv) - me;
#pragma unroll
                for (int j = 0; j < KMAX; j++)
#pragma unroll
                    for (int p = 0; p < PXL; p++) acc[j][p] = tfma<T>(x[f][p], cj[j], acc[j][p]);
            }
        }
        // argmin with strict '<' (lowest index wins ties), _k_means_lloyd.pyx:206-213
        int lab[PXL];
#pragma unroll
        for (int p = 0; p < PXL; p++) {
            T bd = tfma<T>((T)-2, acc[0][p], cs[0]);
            int bl = 0;
#pragma unroll
            for (int j = 1; j < KMAX; j++) {
                if (j < k) {
                    T d = tfma<T>((T)-2, acc[j][p], cs[j]);
                    if (d < bd) { bd = d; bl = j; }
                }
            }
            lab[p] = bl;
        }
        if (!UPDATE && out32) {
            if (FULL || base + PXL <= n) {
                if constexpr (PXL == 4) *reinterpret_cast<int4 *>(out32 + base) = make_int4(lab[0], lab[1], lab[2], lab[3]);
                else *reinterpret_cast<int2 *>(out32 + base) = make_int2(lab[0], lab[1]);
            } else {
                for (int p = 0; p < PXL; p++)
                    if (base + p < n) out32[base + p] = lab[p];
            }
        } else if (!labels) {
            my_changed = 1;   // untracked sweep (see kmeans_fit): "some label changed" as far as the stopping rules go
        } else if (FULL || base + PXL <= n) {
            if constexpr (PXL == 4) {
                uchar4 old = *reinterpret_cast<const uchar4 *>(labels + base);
                my_changed += (old.x != lab[0]) + (old.y != lab[1]) + (old.z != lab[2]) + (old.w != lab[3]);
                *reinterpret_cast<uchar4 *>(labels + base) = make_uchar4(lab[0], lab[1], lab[2], lab[3]);
            } else {
                uchar2 old = *reinterpret_cast<const uchar2 *>(labels + base);
                my_changed += (old.x != lab[0]) + (old.y != lab[1]);
                *reinterpret_cast<uchar2 *>(labels + base) = make_uchar2(lab[0], lab[1]);
            }
        } else {
            for (int p = 0; p < PXL; p++)
                if (base + p < n) {
                    my_changed += labels[base + p] != lab[p];
                    labels[base + p] = (uint8_t)lab[p];
                }
        }
        if (UPDATE) {
            // phase B.  Out-of-range pixels were loaded as 0 and must not be counted.
            bool valid[PXL];
#pragma unroll
            for (int p = 0; p < PXL; p++) valid[p] = FULL || base + p < n;
            bool same = valid[PXL - 1];
#pragma unroll
            for (int p = 1; p < PXL; p++) same = same && lab[p] == lab[0];
            if (same) {
                atomicAdd(&myS[KMAX * F + lab[0]], (unsigned long long)PXL);
            } else {
#pragma unroll
                for (int p = 0; p < PXL; p++)
                    if (valid[p]) atomicAdd(&myS[KMAX * F + lab[p]], 1ull);
            }
#pragma unroll
            for (int f = 0; f < FR; f++) {
                if (f < F) {
                    // RAW bit patterns of fma(x, 2^40, 1.5 * 2^52): the constant's pattern is taken off once per chunk and
                    // cluster (count x constant, modulo 2^64) in the epilogue instead of once per value (as in k3_gram)
                    unsigned long long q[PXL];
#pragma unroll
                    for (int p = 0; p < PXL; p++) q[p] = (unsigned long long)__double_as_longlong(fma((double)x[f][p], 1099511627776.0, FX_MAGIC));
                    if (same) {  // one add for the lane's PXL pixels
                        unsigned long long qs = q[0];
#pragma unroll
                        for (int p = 1; p < PXL; p++) qs += q[p];
                        atomicAdd(&myS[lab[0] * F + f], qs);
                    } else {
#pragma unroll
                        for (int p = 0; p < PXL; p++)
                            if (valid[p]) atomicAdd(&myS[lab[p] * F + f], q[p]);
                    }
                }
            }
        }
    };
    for (int t = 0; t < KM_TILES_PER_CHUNK; t++) {
        const int64_t tbase = chunk0 + (int64_t)t * TILE;
        if (tbase >= n) break;
        const int64_t base = tbase + (int64_t)threadIdx.x * PXL;
        if (tbase + TILE <= n) tile_body(std::true_type{}, base);
        else tile_body(std::false_type{}, base);
    }
    if (UPDATE) {
        int ch = wave_sum(my_changed);
        __shared__ int changed_w[4];
        if (lane == 0) changed_w[threadIdx.x >> 6] = ch;
        __syncthreads();
        const int M = KMAX * F + KMAX + 1;
        for (int i = threadIdx.x; i < M; i += KM_THREADS) {
            long long v;
            if (i < KMAX * F + KMAX) {
                unsigned long long a = 0;
                for (int c = 0; c < ncopies; c++) a += S[(size_t)c * stride + i];
                if (i < KMAX * F) {   // a sum: take count x bits(1.5 * 2^52) off the raw patterns
                    unsigned long long cnt = 0;
                    for (int c = 0; c < ncopies; c++) cnt += S[(size_t)c * stride + KMAX * F + i / F];
                    a -= cnt * (unsigned long long)__double_as_longlong(FX_MAGIC);
                }
                v = (long long)a;
            } else {
                v = changed_w[0] + changed_w[1] + changed_w[2] + changed_w[3];
            }
            partial[(size_t)i * nchunks + blockIdx.x] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Feature-BLOCKED forms for 32 < F <= RSSEG_MAX_FEATURES (the reference's default key selection clusters every 2-D
// plane of the stage-2 dictionary: 55 planes, extract.py:516-522).  F register-resident planes per lane do not fit
// any more, so a tile walks the planes in blocks of KM_FB: the k dot products (and the k-means++ sums) are carried
// across the blocks in the same per-feature fma order as the register-resident kernels — same bits — and the Lloyd
// update re-reads the blocks after the argmin (the second read of a tile comes from the L2: 2 x 4F B/px requested,
// not an HBM-bound path any more but the same arithmetic).
// ------------------------------------------------------------------------------------------------
#define KM_FB 16

template <typename T, bool FULL>
__device__ __forceinline__ void load_block(const planes_t &pl, int F, int f0, int64_t base, int64_t n, T (&x)[KM_FB][vt<T>::PXL])
{
#pragma unroll
    for (int j = 0; j < KM_FB; j++) {
        if (f0 + j < F) load_pxf<T, FULL>(pl.p[f0 + j], base, n, x[j]);
        else {
#pragma unroll
            for (int p = 0; p < vt<T>::PXL; p++) x[j][p] = (T)0;
        }
    }
}

// np.var numerators of one plane per workgroup row (grid: nchunks x F): partial[(1 + f) * nchunks + chunk], the rows
// km_kpp<MODE 0> fills for F <= 32
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void km_var(planes_t pl, int64_t n, const scaler_t<T> *__restrict__ sp,
                                                     unsigned long long *__restrict__ partial, int64_t nchunks)
{
    constexpr int PXL = vt<T>::PXL;
    const int f = blockIdx.y;
    const T sc = sp->scale[f], mnv = sp->minv[f], me = sp->mean[f];
    const int64_t chunk0 = (int64_t)blockIdx.x * km_chunk<T>();
    long long acc = 0;
    for (int t = 0; t < KM_TILES_PER_CHUNK; t++) {
        const int64_t base = chunk0 + (int64_t)t * km_tile<T>() + (int64_t)threadIdx.x * PXL;
        if (base >= n) break;
        T v[PXL];
        load_px<T>(pl.p[f], base, n, v);
#pragma unroll
        for (int i = 0; i < PXL; i++)
            if (base + i < n) {
                const T yt = scaled<T>(v[i], sc, mnv) - me;
                const T dd = yt * yt;
                acc += to_fixed40((double)dd);
            }
    }
    acc = wave_sum(acc);
    __shared__ long long sacc[4];
    if (lane_id() == 0) sacc[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)(1 + f) * nchunks + blockIdx.x] = (unsigned long long)(sacc[0] + sacc[1] + sacc[2] + sacc[3]);
}

// km_kpp for F > 32 (MODE 0 without the variance rows: km_var writes them)
template <typename T, int NL, int MODE>
__global__ __launch_bounds__(KM_THREADS) void km_kpp_blk(planes_t pl, int F, int64_t n, const scaler_t<T> *__restrict__ sp,
                                                         const double *__restrict__ candT, const double *__restrict__ cc, int L,
                                                         T *__restrict__ closest, unsigned long long *__restrict__ partial, int64_t nchunks)
{
    constexpr int PXL = vt<T>::PXL;
    unsigned long long acc[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) acc[l] = 0;
    const int64_t chunk0 = (int64_t)blockIdx.x * km_chunk<T>();
    auto tile_body = [&](auto full_t, int64_t base) {
        constexpr bool FULL = decltype(full_t)::value;
        T cl[PXL];
#pragma unroll
        for (int p = 0; p < PXL; p++) cl[p] = (T)0;
        if constexpr (MODE == 2) load_pxf<T, FULL>(closest, base, n, cl);
        double dot[NL][PXL], dotp[PXL], yy[PXL];
#pragma unroll
        for (int p = 0; p < PXL; p++) {
            yy[p] = 0.0;
            dotp[p] = 0.0;
#pragma unroll
            for (int l = 0; l < NL; l++) dot[l][p] = 0.0;
        }
        for (int f0 = 0; f0 < F; f0 += KM_FB) {
            T x[KM_FB][PXL];
            load_block<T, FULL>(pl, F, f0, base, n, x);
#pragma unroll
            for (int j = 0; j < KM_FB; j++) {
                const int f = f0 + j;
                if (f < F) {
                    const T sc = sp->scale[f], mnv = sp->minv[f], me = sp->mean[f];
                    double cd[NL];
#pragma unroll
                    for (int l = 0; l < NL; l++) cd[l] = candT[f * KPP_STRIDE + l];
                    const double cdp = candT[f * KPP_STRIDE + KPP_MAXL];
#pragma unroll
                    for (int p = 0; p < PXL; p++) {
                        const T yt = scaled<T>(x[j][p], sc, mnv) - me;
                        const double y = (double)yt;
                        yy[p] = fma(y, y, yy[p]);
                        if constexpr (MODE != 0) dotp[p] = fma(cdp, y, dotp[p]);
#pragma unroll
                        for (int l = 0; l < NL; l++) dot[l][p] = fma(cd[l], y, dot[l][p]);
                    }
                }
            }
        }
        if constexpr (MODE != 0) {
            const double ccp = cc[KPP_MAXL];
            bool moved = MODE != 2;   // MODE 2: did the pending centre come closer to any of the thread's pixels?
#pragma unroll
            for (int p = 0; p < PXL; p++) {
                double d = -2.0 * dotp[p];
                d = d + ccp;
                d = d + yy[p];
                T dt = (T)d;
                dt = dt > (T)0 ? dt : (T)0;
                if constexpr (MODE == 2) {
                    moved = moved || dt < cl[p];
                    dt = cl[p] < dt ? cl[p] : dt;
                }
                cl[p] = dt;
            }
            if (FULL || base + PXL <= n) {
                // the closest-distance plane is rewritten only where it changed (a new centre takes over ~1 / (r + 1) of the pixels
                // in round r): less of the write stream that costs a sweep ~10 % of its read rate (profiles/r02_streams.json)
                if (moved) {
                    typename vt<T>::vec o;
                    if constexpr (PXL == 4) o = make_float4(cl[0], cl[1], cl[2], cl[3]);
                    else o = make_double2(cl[0], cl[1]);
                    *reinterpret_cast<typename vt<T>::vec *>(closest + base) = o;
                }
            } else {
                for (int p = 0; p < PXL; p++)
                    if (base + p < n) closest[base + p] = cl[p];
            }
        }
#pragma unroll
        for (int l = 0; l < NL; l++) {
            if (l < L) {
                const double ccl = cc[l];
#pragma unroll
                for (int p = 0; p < PXL; p++) {
                    double d = -2.0 * dot[l][p];
                    d = d + ccl;
                    d = d + yy[p];
                    T dt = (T)d;
                    dt = dt > (T)0 ? dt : (T)0;
                    if constexpr (MODE != 0) dt = cl[p] < dt ? cl[p] : dt;
                    if (FULL || base + p < n) acc[l] += (unsigned long long)to_fixed40((double)dt);
                }
            }
        }
    };
    for (int t = 0; t < KM_TILES_PER_CHUNK; t++) {
        const int64_t tbase = chunk0 + (int64_t)t * km_tile<T>();
        if (tbase >= n) break;
        const int64_t base = tbase + (int64_t)threadIdx.x * PXL;
        if (tbase + km_tile<T>() <= n) tile_body(std::true_type{}, base);
        else tile_body(std::false_type{}, base);
    }
    __shared__ unsigned long long sacc[4][NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
        unsigned long long s = wave_sum(acc[l]);
        if (lane_id() == 0) sacc[threadIdx.x >> 6][l] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < NL && (int)threadIdx.x < L)
        partial[(size_t)threadIdx.x * nchunks + blockIdx.x] =
            sacc[0][threadIdx.x] + sacc[1][threadIdx.x] + sacc[2][threadIdx.x] + sacc[3][threadIdx.x];
}

// km_lloyd for F > 32: same partial layout, same LDS accumulators
template <typename T, int KMAX, bool UPDATE>
__global__ __launch_bounds__(KM_THREADS) void km_lloyd_blk(planes_t pl, int F, int k, int64_t n,
                                                           const scaler_t<T> *__restrict__ sp, const T *__restrict__ cenT,
                                                           const T *__restrict__ csq, uint8_t *__restrict__ labels,
                                                           long long *__restrict__ partial, int64_t nchunks, int ncopies,
                                                           const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr int PXL = vt<T>::PXL;
    constexpr int TILE = KM_THREADS * PXL;
    extern __shared__ __align__(16) char smem[];
    unsigned long long *S = reinterpret_cast<unsigned long long *>(smem);
    const int stride = km_copy_stride(KMAX, F);
    const int lane = threadIdx.x & 63;
    unsigned long long *myS = S + (size_t)(lane & (ncopies - 1)) * stride;
    if (UPDATE) {
        for (int i = threadIdx.x; i < ncopies * stride; i += KM_THREADS) S[i] = 0ull;
        __syncthreads();
    }
    const int64_t chunk0 = (int64_t)blockIdx.x * km_chunk<T>();
    int my_changed = 0;
    auto tile_body = [&](auto full_t, int64_t base) {
        constexpr bool FULL = decltype(full_t)::value;
        T acc[KMAX][PXL];
#pragma unroll
        for (int j = 0; j < KMAX; j++)
#pragma unroll
            for (int p = 0; p < PXL; p++) acc[j][p] = (T)0;
        for (int f0 = 0; f0 < F; f0 += KM_FB) {
            T x[KM_FB][PXL];
            load_block<T, FULL>(pl, F, f0, base, n, x);
#pragma unroll
            for (int jf = 0; jf < KM_FB; jf++) {
                const int f = f0 + jf;
                if (f < F) {
                    const T sc = sp->scale[f], mnv = sp->minv[f], me = sp->mean[f];
#pragma unroll
                    for (int p = 0; p < PXL; p++) x[jf][p] = scaled<T>(x[jf][p], sc, mnv) - me;
#pragma unroll
                    for (int j = 0; j < KMAX; j++) {
                        const T cj = cenT[f * KMAX + j];
#pragma unroll
                        for (int p = 0; p < PXL; p++) acc[j][p] = tfma<T>(x[jf][p], cj, acc[j][p]);
                    }
                }
            }
        }
        int lab[PXL];
#pragma unroll
        for (int p = 0; p < PXL; p++) {
            T bd = tfma<T>((T)-2, acc[0][p], csq[0]);
            int bl = 0;
#pragma unroll
            for (int j = 1; j < KMAX; j++) {
                if (j < k) {
                    T d = tfma<T>((T)-2, acc[j][p], csq[j]);
                    if (d < bd) { bd = d; bl = j; }
                }
            }
            lab[p] = bl;
        }
        if (!labels) {
            my_changed = 1;   // untracked sweep
        } else if (FULL || base + PXL <= n) {
            if constexpr (PXL == 4) {
                uchar4 old = *reinterpret_cast<const uchar4 *>(labels + base);
                my_changed += (old.x != lab[0]) + (old.y != lab[1]) + (old.z != lab[2]) + (old.w != lab[3]);
                *reinterpret_cast<uchar4 *>(labels + base) = make_uchar4(lab[0], lab[1], lab[2], lab[3]);
            } else {
                uchar2 old = *reinterpret_cast<const uchar2 *>(labels + base);
                my_changed += (old.x != lab[0]) + (old.y != lab[1]);
                *reinterpret_cast<uchar2 *>(labels + base) = make_uchar2(lab[0], lab[1]);
            }
        } else {
            for (int p = 0; p < PXL; p++)
                if (base + p < n) {
                    my_changed += labels[base + p] != lab[p];
                    labels[base + p] = (uint8_t)lab[p];
                }
        }
        if (UPDATE) {
            bool valid[PXL];
#pragma unroll
            for (int p = 0; p < PXL; p++) valid[p] = FULL || base + p < n;
#pragma unroll
            for (int p = 0; p < PXL; p++)
                if (valid[p]) atomicAdd(&myS[KMAX * F + lab[p]], 1ull);
            for (int f0 = 0; f0 < F; f0 += KM_FB) {
                T x[KM_FB][PXL];
                load_block<T, FULL>(pl, F, f0, base, n, x);
#pragma unroll
                for (int jf = 0; jf < KM_FB; jf++) {
                    const int f = f0 + jf;
                    if (f < F) {
                        const T sc = sp->scale[f], mnv = sp->minv[f], me = sp->mean[f];
#pragma unroll
                        for (int p = 0; p < PXL; p++)
                            if (valid[p]) {
                                const T xv = scaled<T>(x[jf][p], sc, mnv) - me;
                                atomicAdd(&myS[lab[p] * F + f], (unsigned long long)to_fixed40((double)xv));
                            }
                    }
                }
            }
        }
    };
    for (int t = 0; t < KM_TILES_PER_CHUNK; t++) {
        const int64_t tbase = chunk0 + (int64_t)t * TILE;
        if (tbase >= n) break;
        const int64_t base = tbase + (int64_t)threadIdx.x * PXL;
        if (tbase + TILE <= n) tile_body(std::true_type{}, base);
        else tile_body(std::false_type{}, base);
    }
    if (UPDATE) {
        int ch = wave_sum(my_changed);
        __shared__ int changed_w[4];
        if (lane == 0) changed_w[threadIdx.x >> 6] = ch;
        __syncthreads();
        const int M = KMAX * F + KMAX + 1;
        for (int i = threadIdx.x; i < M; i += KM_THREADS) {
            long long v;
            if (i < KMAX * F + KMAX) {
                unsigned long long a = 0;
                for (int c = 0; c < ncopies; c++) a += S[(size_t)c * stride + i];
                v = (long long)a;
            } else {
                v = changed_w[0] + changed_w[1] + changed_w[2] + changed_w[3];
            }
            partial[(size_t)i * nchunks + blockIdx.x] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// _relocate_empty_clusters_dense (sklearn/cluster/_k_means_common.pyx:167-211), device part: the pixel
// farthest from the OLD centre of its label, distance = ((X - centers_old[labels])**2).sum(axis=1) in T
// with NumPy's pairwise summation order over the F columns; ties -> lowest pixel index; pixels whose
// global index is in `taken` are skipped.  One (distance, index) pair per chunk; the host finishes.
// ------------------------------------------------------------------------------------------------
template <typename T, int FR>
__device__ __forceinline__ T np_pairwise_sum_dev(const T (&a)[FR], int n)
{
    if (n < 8) {
        T res = (T)0;
#pragma unroll
        for (int i = 0; i < 8 && i < FR; i++)
            if (i < n) res = res + a[i];
        return res;
    }
    T r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = a[j < FR ? j : 0];
    const int nb = n / 8;
#pragma unroll
    for (int b = 1; b < FR / 8; b++)
        if (b < nb) {
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = r[j] + a[b * 8 + j];
        }
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int i = 8; i < FR; i++)
        if (i >= nb * 8 && i < n) res = res + a[i];
    return res;
}

struct taken_t {
    long long idx[RSSEG_MAX_CLUSTERS];
    int n;
};

template <typename T, int KMAX, int FR>
__global__ __launch_bounds__(KM_THREADS) void km_farthest(planes_t pl, int F, int64_t n, int64_t offset,
                                                          const scaler_t<T> *__restrict__ sp, const T *__restrict__ cenT,
                                                          const uint8_t *__restrict__ labels, taken_t taken,
                                                          T *__restrict__ pdist, long long *__restrict__ pidx)
{
    const int64_t chunk0 = (int64_t)blockIdx.x * km_chunk<T>();
    T best = (T)-1;
    long long bidx = 0x7fffffffffffffffLL;
    for (int64_t i = chunk0 + threadIdx.x; i < chunk0 + km_chunk<T>() && i < n; i += KM_THREADS) {
        bool skip = false;
        for (int t = 0; t < taken.n; t++) skip = skip || taken.idx[t] == offset + i;
        if (skip) continue;
        const int lab = labels[i];
        T dd[FR];
#pragma unroll
        for (int f = 0; f < FR; f++) {
            if (f < F) {
                const T x = scaled<T>(reinterpret_cast<const T *>(pl.p[f])[i], sp->scale[f], sp->minv[f]) - sp->mean[f];
                const T t = x - cenT[f * KMAX + lab];
                dd[f] = t * t;
            } else {
                dd[f] = (T)0;
            }
        }
        const T d = np_pairwise_sum_dev<T, FR>(dd, F);
        if (d > best) { best = d; bidx = offset + i; }  // ascending i within a thread: first max kept
    }
    // (max distance, min index) over the workgroup
    for (int o = 32; o > 0; o >>= 1) {
        const T ob = __shfl_xor(best, o, 64);
        const long long oi = __shfl_xor(bidx, o, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    __shared__ T sb[4];
    __shared__ long long si[4];
    if (lane_id() == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (sb[w] > best || (sb[w] == best && si[w] < bidx)) { best = sb[w]; bidx = si[w]; }
        pdist[blockIdx.x] = best;
        pidx[blockIdx.x] = bidx;
    }
}

// column sums of partial[M][nchunks] -> out[M][2] = {sum of (v >> 32), sum of (v & 0xffffffff)}
// row m lands in out[2 * (m * ostride + ooff)] (ostride = 1, ooff = 0: consecutive rows)
__global__ __launch_bounds__(KM_THREADS) void km_reduce_cols(const long long *__restrict__ partial, int64_t nchunks,
                                                            long long *__restrict__ out, const int *__restrict__ done, int ostride, int ooff)
{
    if (done && *done) return;
    const int m = blockIdx.x;
    long long hi = 0, lo = 0;
    const long long *row = partial + (size_t)m * nchunks;
    int64_t c = threadIdx.x;
    for (; c + 7 * KM_THREADS < nchunks; c += 8 * KM_THREADS) {   // eight loads in flight per thread
        long long v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = row[c + u * KM_THREADS];
#pragma unroll
        for (int u = 0; u < 8; u++) { hi += v[u] >> 32; lo += v[u] & 0xffffffffLL; }
    }
    for (; c < nchunks; c += KM_THREADS) {
        const long long v = row[c];
        hi += v >> 32;
        lo += v & 0xffffffffLL;
    }
    hi = wave_sum(hi);
    lo = wave_sum(lo);
    __shared__ long long sh[4], sl[4];
    if (lane_id() == 0) { sh[threadIdx.x >> 6] = hi; sl[threadIdx.x >> 6] = lo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * (m * ostride + ooff)] = sh[0] + sh[1] + sh[2] + sh[3];
        out[2 * (m * ostride + ooff) + 1] = sl[0] + sl[1] + sl[2] + sl[3];
    }
}

__global__ __launch_bounds__(KM_THREADS) void km_labels_out(const uint8_t *__restrict__ lab, int32_t *__restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * KM_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * KM_THREADS) out[i] = lab[i];
}

// ------------------------------------------------------------------------------------------------
// Lloyd iterations without the host in the loop.  The state of _kmeans_single_lloyd lives in device memory; per
// iteration the host enqueues  kl_prepare -> km_lloyd<update> -> km_reduce_cols -> [all-reduce hook] -> kl_update  and
// looks at the state only every few iterations.  kl_update does what sklearn does between two E-steps
// (_average_centers, _center_shift, the two stopping rules of _kmeans.py:717-732) with the same operations in the same
// order as the host code it replaces (every T operation is one IEEE operation: -ffp-contract=off), and raises `done`;
// the iterations already enqueued behind it see the flag and return at once.  An EMPTY cluster needs
// _relocate_empty_clusters_dense, which stays a host path: kl_update then parks the reduced sums of that iteration
// (done = 3) and the host loop takes over from exactly there.
// ------------------------------------------------------------------------------------------------
template <typename T> struct km_state {
    // Lloyd (kl_prepare / kl_update)
    T C[RSSEG_MAX_CLUSTERS][RSSEG_MAX_FEATURES];  // k-means++: the centres chosen so far; Lloyd: centres_old of the next E-step
    T tol;
    int it;        // iterations completed
    int done;      // 0 running | 1 labels unchanged (strict) | 2 centre shift <= tol | 3 empty cluster: host takes over | 4 max_iter
    int max_iter;
    int best;      // k-means++: the row of the chunk-partial table that is the current prefix table (the chosen candidate's)
    // geometry of the (possibly row-sharded) raster
    long long n_all[RSSEG_MAX_RANKS];
    long long N, offset, n_local, nchunks;
    int rank, world, k, F, L, last_rank;
    double tol_in;
    // k-means++ (_kmeans.py:213-270)
    double uniforms[(RSSEG_MAX_CLUSTERS - 1) * KPP_MAXL];   // rng.uniform(size=L) of every round, drawn by the host up front
    long long init_idx[RSSEG_MAX_CLUSTERS];
    long long cand_idx[KPP_MAXL];
    T rows[KPP_MAXL][RSSEG_MAX_FEATURES];                   // this round's candidate rows
    T mean[RSSEG_MAX_FEATURES];
    T current_pot;
    u128 rank_tot[RSSEG_MAX_RANKS];                         // every rank's total of the closest-distance sums
};
template <typename T> using lloyd_state = km_state<T>;

template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return __fsqrt_rn(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return __dsqrt_rn(x); }

// cenT / csq of the centres in the state, as kl_prepare writes them (all threads of the workgroup)
template <typename T> __device__ __forceinline__ void kl_fill_cenT(const km_state<T> *st, int k, int F, int KMAX, T *__restrict__ cenT)
{
    for (int i = threadIdx.x; i < KMAX * RSSEG_MAX_FEATURES + KMAX; i += KM_THREADS) {
        T v = (T)0;
        if (i < KMAX * RSSEG_MAX_FEATURES) {
            const int f = i / KMAX, j = i - f * KMAX;
            if (j < k && f < F) v = st->C[j][f];
        } else {
            const int j = i - KMAX * RSSEG_MAX_FEATURES;
            if (j < k)
                for (int f = 0; f < F; f++) v = tfma<T>(st->C[j][f], st->C[j][f], v);
        }
        cenT[i] = v;
    }
}

// cenT[f * KMAX + j] = C[j][f] (zeros elsewhere), csq[j] = fma chain of C[j][f]^2 over f — row_norms(centers, squared=True)
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void kl_prepare(const lloyd_state<T> *__restrict__ st, int k, int F, int KMAX, T *__restrict__ cenT,
                                                         int force)
{
    if (st->done && !force) return;
    kl_fill_cenT<T>(st, k, F, KMAX, cenT);
}

__device__ __forceinline__ i128 dev_limbs(long long hi, long long lo) { return ((i128)hi << 32) + (i128)lo; }
template <typename T> __device__ __forceinline__ T dev_fixed_to_T(i128 s) { return (T)((double)s * (1.0 / 1099511627776.0)); }

template <typename T> __device__ T dev_pairwise_sum(const T *a, int n)   // numpy pairwise sum, n <= 128
{
    if (n < 8) {
        T res = (T)0;
        for (int i = 0; i < n; i++) res = res + a[i];
        return res;
    }
    T r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + a[i];
    return res;
}

// red: [M][2] limb sums of this iteration over all ranks (km_reduce_cols, all-reduced), M = KMAX * F + KMAX + 1
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void kl_update(lloyd_state<T> *st, int k, int F, int KMAX, const long long *__restrict__ red,
                                                        long long *__restrict__ red_saved, T *__restrict__ cenT)
{
    if (st->done) return;
    __shared__ long long cnt[RSSEG_MAX_CLUSTERS];
    __shared__ T shift2[RSSEG_MAX_CLUSTERS];
    __shared__ int n_empty, s_done;
    __shared__ __align__(16) char kl_scratch[sizeof(T) * RSSEG_MAX_CLUSTERS * RSSEG_MAX_FEATURES];
    const int M = KMAX * F + KMAX + 1;
    if (threadIdx.x == 0) n_empty = 0;
    __syncthreads();
    if ((int)threadIdx.x < k) {
        const long long c = (long long)dev_limbs(red[2 * (KMAX * F + threadIdx.x)], red[2 * (KMAX * F + threadIdx.x) + 1]);
        cnt[threadIdx.x] = c;
        if (c == 0) atomicAdd(&n_empty, 1);
    }
    __syncthreads();
    if (n_empty > 0) {   // _relocate_empty_clusters_dense is a host path: keep this iteration's sums for it
        for (int i = threadIdx.x; i < 2 * M; i += KM_THREADS) red_saved[i] = red[i];
        __syncthreads();
        if (threadIdx.x == 0) st->done = 3;
        return;
    }
    // _average_centers for all (cluster, feature) pairs in parallel, then _center_shift per cluster (one thread per cluster
    // adds the squared differences in the host's order: groups of four, then the remainder)
    T *dsq = reinterpret_cast<T *>(kl_scratch);   // [k][F] squared differences
    for (int i = threadIdx.x; i < k * F; i += KM_THREADS) {
        const int j = i / F, f = i - j * F;
        const T w = (T)cnt[j];
        const T alpha = (T)(1.0 / (double)w);
        const T sT = dev_fixed_to_T<T>(dev_limbs(red[2 * (j * F + f)], red[2 * (j * F + f) + 1]));
        const T cnew = sT * alpha;
        const T d = cnew - st->C[j][f];
        dsq[i] = d * d;
        st->C[j][f] = cnew;
    }
    __syncthreads();
    if ((int)threadIdx.x < k) {
        const int j = threadIdx.x;
        const T *t = dsq + j * F;
        T result = (T)0;
        const int n4 = F / 4, rem = F % 4;
        for (int g = 0; g < n4; g++, t += 4) {
            const T g1 = t[0] + t[1];
            const T g2 = g1 + t[2];
            const T g4 = g2 + t[3];
            result = result + g4;
        }
        for (int r = 0; r < rem; r++) result = result + t[r];
        const T sh = t_sqrt<T>(result);
        shift2[j] = sh * sh;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long changed = (long long)dev_limbs(red[2 * (KMAX * F + KMAX)], red[2 * (KMAX * F + KMAX) + 1]);
        const int it = st->it + 1;
        st->it = it;
        int done = 0;
        if (changed == 0) done = 1;
        else {
            const T tot = dev_pairwise_sum<T>(shift2, k);
            if (tot <= st->tol) done = 2;
            else if (it >= st->max_iter) done = 4;
        }
        st->done = done;
        s_done = done;
    }
    __syncthreads();
    if (!s_done) kl_fill_cenT<T>(st, k, F, KMAX, cenT);   // the next E-step's centres (a converged loop keeps the last E-step's)
}

// ------------------------------------------------------------------------------------------------
// k-means++ without the host in the loop: what the host did between the sweeps (means, tolerance, the D^2 sampling of a
// round's candidates, the choice of the best candidate) as single-workgroup kernels over km_state, same operations in
// the same order.  Values that cross ranks sit in `x` (the communication buffer when world > 1) and are all-reduced
// by the hook between two of these kernels, stream-ordered.
// ------------------------------------------------------------------------------------------------
// ceil(r * 2^40) for a finite r >= 0 as an exact integer (the host computed ceill((long double)r * 2^40))
__device__ __forceinline__ u128 kc_ceil_scaled(double r)
{
    if (!(r > 0.0)) return (u128)0;
    const unsigned long long b = (unsigned long long)__double_as_longlong(r);
    const int be = (int)((b >> 52) & 0x7ff);
    unsigned long long m = b & 0xfffffffffffffull;
    int e;
    if (be == 0) e = -1074;                    // subnormal
    else { m |= 1ull << 52; e = be - 1075; }
    const int sft = e + 40;
    if (sft >= 0) return (u128)m << sft;
    const int sh = -sft;
    if (sh >= 64) return (u128)1;              // 0 < r * 2^40 < 1
    const unsigned long long q = m >> sh, rem = m & ((1ull << sh) - 1ull);
    return (u128)q + (rem ? 1 : 0);
}
__device__ __forceinline__ u128 kc_ulimbs(long long hi, long long lo) { return ((u128)(unsigned long long)hi << 32) + (u128)(unsigned long long)lo; }

// the (not yet centred) row of the first centre as bit patterns behind the F limb pairs of the column sums (which
// km_reduce_cols leaves in x[0 .. 2F) from km_moment's block partials)
template <typename T>
__global__ __launch_bounds__(64) void kc_mean_pack(const km_state<T> *__restrict__ st, const T *__restrict__ row, int own0, long long *__restrict__ x)
{
    const int F = st->F, f = threadIdx.x;
    if (f >= F) return;
    if (st->n_local <= 0) { x[2 * f] = 0; x[2 * f + 1] = 0; }
    double rv = own0 ? (double)row[f] : 0.0;
    if (rv == 0.0) rv = 0.0;   // -0.0 would not survive the integer sum as a zero
    x[2 * F + f] = __double_as_longlong(rv);
}

// X.mean(axis=0) (-> the scaler on the device), the first centre, and the candidate table of the first k-means++ sweep
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void kc_mean_apply(km_state<T> *__restrict__ st, scaler_t<T> *__restrict__ sp, const long long *__restrict__ x,
                                                            double *__restrict__ candT)
{
    const int F = st->F;
    double *cc = candT + (size_t)KPP_STRIDE * RSSEG_MAX_FEATURES;
    for (int i = threadIdx.x; i < KPP_STRIDE * RSSEG_MAX_FEATURES + KPP_STRIDE; i += KM_THREADS) candT[i] = 0.0;
    __syncthreads();
    const T Nt = (T)st->N;
    if ((int)threadIdx.x < F) {
        const int f = threadIdx.x;
        const T sT = dev_fixed_to_T<T>(dev_limbs(x[2 * f], x[2 * f + 1]));
        const T m = sT / Nt;
        sp->mean[f] = m;
        st->mean[f] = m;
        const T raw = (T)__longlong_as_double(x[2 * F + f]);
        const T c0 = raw - m;     // the centring the sweeps apply: fl(fl(fl(x * scale) + min) - mean)
        st->C[0][f] = c0;
        candT[f * KPP_STRIDE] = (double)c0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int f = 0; f < F; f++) a = fma((double)st->C[0][f], (double)st->C[0][f], a);
        cc[0] = a;
    }
}

// tol = mean(np.var(X, axis=0)) * tol_in, every rank's total of the first potential, current_pot
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void kc_tol(km_state<T> *__restrict__ st, const long long *__restrict__ x)
{
    __shared__ T var[RSSEG_MAX_FEATURES];
    const int F = st->F;
    const T Nt = (T)st->N;
    if ((int)threadIdx.x < F) {
        const T sT = dev_fixed_to_T<T>(dev_limbs(x[2 * threadIdx.x], x[2 * threadIdx.x + 1]));
        var[threadIdx.x] = sT / Nt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const T m = dev_pairwise_sum<T>(var, F);
        const T m2 = m / (T)F;
        st->tol = m2 * (T)st->tol_in;
        u128 total = 0;
        for (int r = 0; r < st->world; r++) {
            st->rank_tot[r] = kc_ulimbs(x[2 * F + 2 * r], x[2 * F + 2 * r + 1]);
            total += st->rank_tot[r];
        }
        st->current_pot = (T)((double)total * (1.0 / 1099511627776.0));
        st->best = 0;
    }
}

// Start of round c: the centre chosen last becomes the PENDING column of the candidate table (its distances are folded
// into the closest plane by this round's sweep), and the round's L samples are located: owner rank from the ranks'
// totals, chunk from this rank's prefix table (row `best` of the chunk partials).  sa: what km_kpp_chunkq / km_kpp_sample read.
template <typename T> __device__ __forceinline__ void kc_pick_body(km_state<T> *st, int c, const long long *x);

// pick_c > 0: first close round pick_c (kc_pick) with the potentials in `lim`.  Then both exchange regions are cleared
// for this round: xbuf (the sampled rows, filled by their owner) and lim (the potentials, filled per rank).
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void kc_targets(km_state<T> *st, int c, int pick_c, const unsigned long long *__restrict__ part,
                                                         double *__restrict__ candT, kpp_sample_args *__restrict__ sa_out, double *xbuf, long long *lim)
{
    if (pick_c > 0) kc_pick_body<T>(st, pick_c, lim);
    for (int i = threadIdx.x; i < KPP_MAXL * (1 + RSSEG_MAX_FEATURES); i += KM_THREADS) xbuf[i] = 0.0;
    for (int i = threadIdx.x; i < 2 * KPP_MAXL * RSSEG_MAX_RANKS; i += KM_THREADS)
        if (i < 2 * st->L * st->world) lim[i] = 0;
    const int F = st->F, L = st->L;
    double *cc = candT + (size_t)KPP_STRIDE * RSSEG_MAX_FEATURES;
    for (int i = threadIdx.x; i < KPP_STRIDE * RSSEG_MAX_FEATURES + KPP_STRIDE; i += KM_THREADS) candT[i] = 0.0;
    __syncthreads();
    if ((int)threadIdx.x < F) candT[threadIdx.x * KPP_STRIDE + KPP_MAXL] = (double)st->C[c - 1][threadIdx.x];
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int f = 0; f < F; f++) a = fma((double)st->C[c - 1][f], (double)st->C[c - 1][f], a);
        cc[KPP_MAXL] = a;
    }
    __shared__ kpp_sample_args sa;
    __shared__ u128 ssum[KM_THREADS];   // sums of KM_THREADS contiguous slices of this rank's prefix table
    __shared__ u128 gsum[16];           // sums of 16 groups of 16 slices
    for (int i = threadIdx.x; i < (int)(sizeof(sa) / 4); i += KM_THREADS) reinterpret_cast<int *>(&sa)[i] = 0;
    const long long nchunks = st->nchunks, n = st->n_local;
    const unsigned long long *pre = part + (size_t)st->best * nchunks;
    const long long slice = (nchunks + KM_THREADS - 1) / KM_THREADS;
    {
        u128 mine = 0;
        const long long c_lo = (long long)threadIdx.x * slice, c_hi = c_lo + slice < nchunks ? c_lo + slice : nchunks;
        if (n > 0) {
            long long ch = c_lo;
            for (; ch + 8 <= c_hi; ch += 8) {   // eight loads in flight
                unsigned long long v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = pre[ch + u];
#pragma unroll
                for (int u = 0; u < 8; u++) mine += v[u];
            }
            for (; ch < c_hi; ch++) mine += pre[ch];
        }
        ssum[threadIdx.x] = mine;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        u128 g = 0;
        for (int t = 0; t < 16; t++) g += ssum[threadIdx.x * 16 + t];
        gsum[threadIdx.x] = g;
    }
    if (threadIdx.x == 0) {
        sa.with_old = c > 1 ? 1 : 0;
        sa.offset = st->offset;
    }
    __syncthreads();
    // the L samples are independent: lane 0 of wave w locates candidates w, w + 4, ...
    if ((threadIdx.x & 63) == 0) {
        for (int l = threadIdx.x >> 6; l < L; l += KM_THREADS / 64) {
            const double r = st->uniforms[(c - 1) * L + l] * (double)st->current_pot;   // uniform(size=L) * current_pot
            const u128 target = kc_ceil_scaled(r);
            u128 before = 0;
            int owner = -1;
            for (int rk = 0; rk < st->world; rk++) {   // first rank whose inclusive prefix reaches the target
                if (st->n_all[rk] > 0 && before + st->rank_tot[rk] >= target) { owner = rk; break; }
                before += st->rank_tot[rk];
            }
            if (owner < 0) {                           // beyond the total: np.clip(candidate_ids, None, N - 1)
                if (st->rank == st->last_rank) {
                    sa.mode[l] = 2;
                    sa.c0[l] = 0;
                    sa.direct[l] = n - 1;
                }
                continue;
            }
            if (owner != st->rank) continue;
            u128 run = before;
            int g = 0;
            for (; g < 16; g++) {
                if (run + gsum[g] >= target) break;
                run += gsum[g];
            }
            int t = g * 16;
            if (g < 16)
                for (; t < g * 16 + 16; t++) {
                    if (run + ssum[t] >= target) break;
                    run += ssum[t];
                }
            long long ch = (long long)t * slice;
            if (g < 16 && t < g * 16 + 16)
                for (; ch < nchunks; ch++) {
                    if (run + pre[ch] >= target) break;
                    run += pre[ch];
                }
            if (ch >= nchunks) ch = nchunks - 1;   // cannot happen: this rank's total reaches the target
            sa.mode[l] = 1;
            sa.c0[l] = ch * (long long)km_chunk<T>();
            const long long left = n - sa.c0[l];
            sa.cn[l] = left < (long long)km_chunk<T>() ? left : (long long)km_chunk<T>();
            sa.rem[l] = (unsigned long long)(target - run);   // <= the chunk's own sum: fits 64 bits
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (int)(sizeof(sa) / 4); i += KM_THREADS) reinterpret_cast<int *>(sa_out)[i] = reinterpret_cast<const int *>(&sa)[i];
}

// the sampled rows (all ranks now hold them: {global index, F values} per candidate, stride 1 + RSSEG_MAX_FEATURES) become
// this round's candidate columns
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void kc_cands(km_state<T> *__restrict__ st, const double *__restrict__ xbuf, double *__restrict__ candT)
{
    const int F = st->F, L = st->L;
    double *cc = candT + (size_t)KPP_STRIDE * RSSEG_MAX_FEATURES;
    for (int i = threadIdx.x; i < L * F; i += KM_THREADS) {
        const int l = i / F, f = i - l * F;
        const T v = (T)xbuf[(size_t)l * (1 + RSSEG_MAX_FEATURES) + 1 + f];
        st->rows[l][f] = v;
        candT[f * KPP_STRIDE + l] = (double)v;
    }
    __syncthreads();
    if ((int)threadIdx.x < L) {
        const int l = threadIdx.x;
        st->cand_idx[l] = (long long)xbuf[(size_t)l * (1 + RSSEG_MAX_FEATURES)];
        double a = 0.0;
        for (int f = 0; f < F; f++) a = fma((double)st->rows[l][f], (double)st->rows[l][f], a);
        cc[l] = a;
    }
}

// potentials of the L candidates from every rank's chunk-sum totals (x: [L][world] limb pairs), the best one wins
template <typename T> __device__ __forceinline__ void kc_pick_body(km_state<T> *st, int c, const long long *x)
{
    __shared__ int s_best;
    const int L = st->L, W = st->world;
    if (threadIdx.x == 0) {
        int best = 0;
        T best_pot = (T)0;
        for (int l = 0; l < L; l++) {
            u128 tot = 0;
            for (int r = 0; r < W; r++) tot += kc_ulimbs(x[2 * (l * W + r)], x[2 * (l * W + r) + 1]);
            const T pt = (T)((double)tot * (1.0 / 1099511627776.0));
            if (l == 0 || pt < best_pot) { best = l; best_pot = pt; }
        }
        st->current_pot = best_pot;
        st->init_idx[c] = st->cand_idx[best];
        for (int r = 0; r < W; r++) st->rank_tot[r] = kc_ulimbs(x[2 * (best * W + r)], x[2 * (best * W + r) + 1]);
        st->best = best;   // its chunk sums are the next prefix table (its min-plane is re-derived by the next round)
        s_best = best;
    }
    __syncthreads();
    if ((int)threadIdx.x < st->F) st->C[c][threadIdx.x] = st->rows[s_best][threadIdx.x];
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(KM_THREADS) void kc_pick(km_state<T> *st, int c, const long long *x)
{
    kc_pick_body<T>(st, c, x);
}

// ================================================================================================
// host side
// ================================================================================================
namespace {

// MT19937 as seeded by numpy.random.RandomState(int) (init_genrand) — _kmeans.py:1465 check_random_state
struct mt19937 {
    uint32_t mt[624];
    int idx;
    explicit mt19937(uint32_t s)
    {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    uint32_t next()
    {
        if (idx >= 624) {
            for (int k = 0; k < 624; k++) {
                uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double random_sample()  // genrand_res53
    {
        uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// s_i = p + p + ... (i terms, sequential float64 additions starting from 0.0) in O(#binades).
// This is np.cumsum of the constant probability vector inside RandomState.choice (mtrand: cdf = p.cumsum()).
double seq_sum(double p, int64_t count)
{
    double s = 0.0;
    int64_t left = count;
    while (left > 0) {
        double s1 = s + p;  // one real step
        left--;
        if (left == 0) return s1;
        int e0, e1;
        frexp(s, &e0);
        frexp(s1, &e1);
        s = s1;
        if (s == 0.0) continue;
        // probe the increment now in force; constant while the result stays in this binade
        double s2 = s + p;
        int e2;
        frexp(s2, &e2);
        if (e2 != e1) continue;  // next step leaves the binade: take it as a real step
        const double d = s2 - s;  // exact (same binade)
        if (d <= 0.0) return s;   // p no longer changes the sum
        const double top = ldexp(1.0, e1);  // exclusive upper bound of the binade
        // largest j with s + j*d < top
        double jf = floor((top - s) / d);
        int64_t j = (int64_t)jf;
        while (j > 0 && s + (double)j * d >= top) j--;
        // rounding of the tie case alternates only on the first step, which was taken for real above;
        // verify the increment after that step, fall back to single steps if it differs
        double s3 = s2 + p;
        int e3;
        frexp(s3, &e3);
        if (e3 == e1 && (s3 - s2) != d) {
            s = s2;
            left--;
            continue;
        }
        if (j > left) j = left;
        if (j <= 0) continue;
        s = s + (double)j * d;  // exact: multiples of the binade's ulp
        left -= j;
    }
    return s;
}

// RandomState.choice(n, p = ones/ones.sum()) for one draw u (mtrand.pyx: cdf = p.cumsum(); cdf /= cdf[-1];
// idx = cdf.searchsorted(u, side='right'))
int64_t uniform_choice(int64_t n, int dtype, double u)
{
    double p;
    if (dtype == RSSEG_F32) {
        float s = (float)n;
        float pf = 1.0f / s;
        p = (double)pf;
    } else {
        p = 1.0 / (double)n;
    }
    const double total = seq_sum(p, n);
    int64_t lo = 0, hi = n;  // smallest idx in [0, n] with cdf[idx] > u
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        double c = seq_sum(p, mid + 1) / total;
        if (c > u) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

template <typename T> T np_pairwise_sum(const T *a, int n)
{
    if (n < 8) {
        T res = (T)0;
        for (int i = 0; i < n; i++) res = res + a[i];
        return res;
    }
    T r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + a[i];
    return res;
}

template <typename T> T fixed_to_T(i128 s) { return (T)((double)s * (1.0 / 1099511627776.0)); }

template <typename T> struct teps;
template <> struct teps<float> { static constexpr float v = 1.1920928955078125e-07f; };
template <> struct teps<double> { static constexpr double v = 2.220446049250313e-16; };

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// combine {hi, lo} limb sums (km_reduce_cols) into one signed 128-bit value
i128 limbs(long long hi, long long lo) { return ((i128)hi << 32) + (i128)lo; }

template <typename T, int KMAX, int FR>
int launch_lloyd2(rsseg_ctx *ctx, bool update, int64_t nchunks, size_t lds, planes_t pl, int F, int k, int64_t n,
                  const scaler_t<T> *sp, const T *cenT, const T *csq, uint8_t *labels, long long *partial, int ncopies, const int *done,
                  int32_t *out32)
{
    if (update) {
        // per device: the dynamic-LDS limit already granted to this instantiation (contexts of several threads may race here)
        static std::mutex mu;
        static size_t attr[64] = {0};
        {
            std::lock_guard<std::mutex> g(mu);
            size_t &have = attr[ctx->device & 63];
            if (have == 0) have = 48 * 1024;  // default limit without the attribute
            if (lds > have) {
                HIPCHK(ctx, hipFuncSetAttribute((const void *)km_lloyd<T, KMAX, FR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                have = lds;
            }
        }
        hipLaunchKernelGGL((km_lloyd<T, KMAX, FR, true>), dim3((unsigned)nchunks), dim3(KM_THREADS), lds, ctx->stream, pl, F, k, n, sp,
                           cenT, csq, labels, partial, nchunks, ncopies, done, (int32_t *)nullptr);
    } else {
        hipLaunchKernelGGL((km_lloyd<T, KMAX, FR, false>), dim3((unsigned)nchunks), dim3(KM_THREADS), 0, ctx->stream, pl, F, k, n, sp,
                           cenT, csq, labels, partial, nchunks, ncopies, done, out32);
    }
    return RSSEG_OK;
}

template <typename T, int KMAX>
int launch_lloyd(rsseg_ctx *ctx, bool update, int64_t nchunks, size_t lds, planes_t pl, int F, int k, int64_t n,
                 const scaler_t<T> *sp, const T *cenT, const T *csq, uint8_t *labels, long long *partial, int ncopies, const int *done,
                 int32_t *out32 = nullptr)
{
    // out32: only the register-resident kernels (F <= 32) write the int32 plane themselves; lloyd_writes_int32() tells the caller
    // float64 planes with 33 ... 64 clusters: the register-resident kernels for 9 ... 32 planes need more than 512 registers
    // (252 - 628 bytes of scratch per lane in the r03 code object) — the feature-blocked kernel below holds one
    // block of planes at a time and does not spill at any size (346 - 370 registers), same arithmetic, same bits
    constexpr bool wide64 = std::is_same<T, double>::value && KMAX == 64;
    if (F <= 8) return launch_lloyd2<T, KMAX, 8>(ctx, update, nchunks, lds, pl, F, k, n, sp, cenT, csq, labels, partial, ncopies, done, out32);
    if (!wide64) {
        if (F <= 16) return launch_lloyd2<T, KMAX, 16>(ctx, update, nchunks, lds, pl, F, k, n, sp, cenT, csq, labels, partial, ncopies, done, out32);
        if (F <= 32) return launch_lloyd2<T, KMAX, 32>(ctx, update, nchunks, lds, pl, F, k, n, sp, cenT, csq, labels, partial, ncopies, done, out32);
    }
    // 32 < F <= RSSEG_MAX_FEATURES: the feature-blocked kernel
    if (update) {
        static std::mutex mu;
        static size_t attr[64] = {0};
        {
            std::lock_guard<std::mutex> g(mu);
            size_t &have = attr[ctx->device & 63];
            if (have == 0) have = 48 * 1024;
            if (lds > have) {
                HIPCHK(ctx, hipFuncSetAttribute((const void *)km_lloyd_blk<T, KMAX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                have = lds;
            }
        }
        hipLaunchKernelGGL((km_lloyd_blk<T, KMAX, true>), dim3((unsigned)nchunks), dim3(KM_THREADS), lds, ctx->stream, pl, F, k, n, sp,
                           cenT, csq, labels, partial, nchunks, ncopies, done);
    } else {
        hipLaunchKernelGGL((km_lloyd_blk<T, KMAX, false>), dim3((unsigned)nchunks), dim3(KM_THREADS), 0, ctx->stream, pl, F, k, n, sp,
                           cenT, csq, labels, partial, nchunks, ncopies, done);
    }
    return RSSEG_OK;
}

template <typename T, int NL, int FR>
void launch_kpp3(rsseg_ctx *ctx, int64_t nchunks, planes_t pl, int F, int64_t n, const scaler_t<T> *sp, const double *cand, const double *cc,
                 int L, int mode, T *closest, unsigned long long *partial)
{
#define KPP_GO(M) hipLaunchKernelGGL((km_kpp<T, NL, FR, M>), dim3((unsigned)nchunks), dim3(KM_THREADS), 0, ctx->stream, pl, F, n, sp, cand, cc, L, closest, partial, nchunks)
    if constexpr (NL == 1) {
        KPP_GO(0);
    } else {
        if (mode == 1) KPP_GO(1);
        else KPP_GO(2);
    }
#undef KPP_GO
}
template <typename T, int NL>
void launch_kpp2(rsseg_ctx *ctx, int64_t nchunks, planes_t pl, int F, int64_t n, const scaler_t<T> *sp, const double *cand, const double *cc,
                 int L, int mode, T *closest, unsigned long long *partial)
{
    if (F <= 8) launch_kpp3<T, NL, 8>(ctx, nchunks, pl, F, n, sp, cand, cc, L, mode, closest, partial);
    else if (F <= 16) launch_kpp3<T, NL, 16>(ctx, nchunks, pl, F, n, sp, cand, cc, L, mode, closest, partial);
    else if (F <= 32) launch_kpp3<T, NL, 32>(ctx, nchunks, pl, F, n, sp, cand, cc, L, mode, closest, partial);
    else {   // feature-blocked kernels; the variance rows of the first pass come from km_var
#define KPP_GOB(M) hipLaunchKernelGGL((km_kpp_blk<T, NL, M>), dim3((unsigned)nchunks), dim3(KM_THREADS), 0, ctx->stream, pl, F, n, sp, cand, cc, L, closest, partial, nchunks)
        if constexpr (NL == 1) {
            KPP_GOB(0);
            hipLaunchKernelGGL((km_var<T>), dim3((unsigned)nchunks, F), dim3(KM_THREADS), 0, ctx->stream, pl, n, sp, partial, nchunks);
        } else {
            if (mode == 1) KPP_GOB(1);
            else KPP_GOB(2);
        }
#undef KPP_GOB
    }
}
// mode 0: first centre (L == 1, no per-pixel output); 1: first sampling round; 2: later rounds
template <typename T>
void launch_kpp(rsseg_ctx *ctx, int64_t nchunks, planes_t pl, int F, int64_t n, const scaler_t<T> *sp, const double *cand, const double *cc,
                int L, int mode, T *closest, unsigned long long *partial)
{
    if (mode == 0) launch_kpp2<T, 1>(ctx, nchunks, pl, F, n, sp, cand, cc, 1, 0, closest, partial);
    else if (L <= 4) launch_kpp2<T, 4>(ctx, nchunks, pl, F, n, sp, cand, cc, L, mode, closest, partial);
    else launch_kpp2<T, 8>(ctx, nchunks, pl, F, n, sp, cand, cc, L, mode, closest, partial);
}

template <typename T, int KMAX>
void launch_farthest(rsseg_ctx *ctx, int64_t nchunks, planes_t pl, int F, int64_t n, int64_t offset, const scaler_t<T> *sp,
                     const T *cenT, const uint8_t *labels, const taken_t &tk, T *pdist, long long *pidx)
{
    if (F <= 8) hipLaunchKernelGGL((km_farthest<T, KMAX, 8>), dim3((unsigned)nchunks), dim3(KM_THREADS), 0, ctx->stream, pl, F, n, offset, sp, cenT, labels, tk, pdist, pidx);
    else if (F <= 16) hipLaunchKernelGGL((km_farthest<T, KMAX, 16>), dim3((unsigned)nchunks), dim3(KM_THREADS), 0, ctx->stream, pl, F, n, offset, sp, cenT, labels, tk, pdist, pidx);
    else if (F <= 32) hipLaunchKernelGGL((km_farthest<T, KMAX, 32>), dim3((unsigned)nchunks), dim3(KM_THREADS), 0, ctx->stream, pl, F, n, offset, sp, cenT, labels, tk, pdist, pidx);
    else hipLaunchKernelGGL((km_farthest<T, KMAX, 64>), dim3((unsigned)nchunks), dim3(KM_THREADS), 0, ctx->stream, pl, F, n, offset, sp, cenT, labels, tk, pdist, pidx);
}

template <typename T>
int kmeans_fit(rsseg_ctx *ctx, const void *const *d_planes, int F, int64_t n, int k, uint32_t seed, int max_iter, double tol_in,
               int32_t *d_labels, double *centers_out, rsseg_kmeans_info *info, const double *known_min, const double *known_max)
{
    const double t_start = now_ms();
    const int KMAX = k <= 8 ? 8 : (k <= 16 ? 16 : (k <= 32 ? 32 : 64));
    const int L = 2 + (int)std::log((double)k);
    if (L > KPP_MAXL) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "kmeans: too many local trials");
    constexpr int TILE = km_tile<T>();
    constexpr int CHUNK = km_chunk<T>();
    const int64_t nchunks = std::max<int64_t>(1, ceil_div64(n, CHUNK));
    const int nblk = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ceil_div64(n, (int64_t)KM_THREADS * vt<T>::PXL)));
    const int M = KMAX * F + KMAX + 1;

    // ---- rank geometry: every rank's pixel count travels in the MinMax all-reduce below (one collective fewer) ----
    int64_t n_all[RSSEG_MAX_RANKS] = {0};
    n_all[ctx->rank] = n;
    int64_t N = 0, offset = 0;
    // ---- workspace layout ----
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_sp = carve(sizeof(scaler_t<T>));
    // centres + their squared norms, candidates + theirs: each pair contiguous, so that one copy uploads both
    const size_t o_cen = carve(sizeof(T) * ((size_t)KMAX * RSSEG_MAX_FEATURES + KMAX));
    const size_t o_cand = carve(sizeof(double) * ((size_t)KPP_STRIDE * RSSEG_MAX_FEATURES + KPP_STRIDE));
    const size_t o_row = carve(sizeof(T) * RSSEG_MAX_FEATURES);
    const size_t o_samp = carve(sizeof(double) * KPP_MAXL * (1 + RSSEG_MAX_FEATURES));
    const size_t o_qv = carve(sizeof(unsigned long long) * KPP_MAXL * (size_t)CHUNK);
    const size_t o_red = carve(sizeof(long long) * std::max<size_t>({(size_t)2 * M, (size_t)3 * RSSEG_MAX_FEATURES + 2 * RSSEG_MAX_RANKS,
                                                                    (size_t)2 * KPP_MAXL * RSSEG_MAX_RANKS + KPP_MAXL * (1 + RSSEG_MAX_FEATURES) + 64}));
    const size_t o_sa = carve(sizeof(kpp_sample_args));
    const size_t o_redsave = carve(sizeof(long long) * 2 * M);
    const size_t o_lst = carve(sizeof(lloyd_state<T>));
    const size_t o_mm = carve(sizeof(T) * 2 * (size_t)nblk * F);
    const size_t o_mom = carve(sizeof(long long) * (size_t)nblk * F);
    const size_t o_part = carve(sizeof(long long) * std::max<size_t>((size_t)M, KPP_MAXL) * (size_t)nchunks);
    const size_t dist_stride = (sizeof(T) * (size_t)std::max<int64_t>(n, 1) + 255) & ~(size_t)255;
    const size_t o_closest = carve(dist_stride);            // the ONE closest-distance plane of k-means++
    const size_t o_lab = carve((size_t)std::max<int64_t>(n, 1) + 64);
    RSCHK(ws_reserve(ctx, off));
    const size_t pin_need = std::max<size_t>({sizeof(T) * 2 * (size_t)nblk * F, sizeof(long long) * (size_t)nblk * F + sizeof(double) * RSSEG_MAX_FEATURES + 64,
                                              sizeof(long long) * (size_t)std::max(KPP_MAXL, F + 1) * (size_t)nchunks, sizeof(T) * (size_t)CHUNK,
                                              sizeof(long long) * 2 * (size_t)M + sizeof(km_state<T>), (size_t)65536});
    // small host-to-device uploads (the state, centres of the host-side iterations) go through a ring of 4 pinned slots behind
    // the read-back area: the copy is then asynchronous for real and needs no synchronisation before the stack buffer it came
    // from dies.  Invariant: a slot is reused only after a host synchronisation since its last use — the device-resident
    // loop makes at most two uploads between two waits today, and the ring enforces it: a fifth upload without a wait in
    // between (ctx->host_syncs is the epoch) waits for the stream first.
    constexpr size_t UP_SLOT = 64 * 1024;
    const size_t pin_main = (pin_need + 255) & ~(size_t)255;
    RSCHK(pin_reserve(ctx, pin_main + 4 * UP_SLOT));
    unsigned up_i = 0, up_since_sync = 0;
    long long up_epoch = ctx->host_syncs;
    auto upload = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        if (ctx->host_syncs != up_epoch) { up_epoch = ctx->host_syncs; up_since_sync = 0; }
        if (up_since_sync == 4) {          // every slot is (possibly) still being copied
            const hipError_t e = rs_sync(ctx);
            if (e != hipSuccess) return e;
            up_epoch = ctx->host_syncs;
            up_since_sync = 0;
        }
        up_since_sync++;
        char *slot = ctx->h_pin + pin_main + (size_t)(up_i++ & 3u) * UP_SLOT;
        memcpy(slot, src, bytes);
        return hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream);
    };
    char *ws = ctx->d_ws;
    scaler_t<T> *d_sp = (scaler_t<T> *)(ws + o_sp);
    T *d_cen = (T *)(ws + o_cen);
    T *d_csq = d_cen + (size_t)KMAX * RSSEG_MAX_FEATURES;
    double *d_cand = (double *)(ws + o_cand);
    double *d_cc = d_cand + (size_t)KPP_STRIDE * RSSEG_MAX_FEATURES;
    T *d_row = (T *)(ws + o_row);
    double *d_samp = (double *)(ws + o_samp);
    unsigned long long *d_qv = (unsigned long long *)(ws + o_qv);
    long long *d_red = (long long *)(ws + o_red);
    long long *d_redsave = (long long *)(ws + o_redsave);
    lloyd_state<T> *d_lst = (lloyd_state<T> *)(ws + o_lst);
    T *d_mm = (T *)(ws + o_mm);
    long long *d_mom = (long long *)(ws + o_mom);
    long long *d_part = (long long *)(ws + o_part);
    T *d_closest = (T *)(ws + o_closest);
    uint8_t *d_lab = (uint8_t *)(ws + o_lab);
    hipStream_t st = ctx->stream;

    planes_t pl;
    memset(&pl, 0, sizeof(pl));
    for (int f = 0; f < F; f++) {
        // a rank without pixels (n_local == 0) may pass anything, also NULL: its planes are never read
        if (n > 0 && (!d_planes[f] || ((uintptr_t)d_planes[f] & 15))) return rs_fail(ctx, RSSEG_ERR_INVALID, "kmeans: plane %d null or not 16-byte aligned", f);
        pl.p[f] = d_planes[f];
    }

    // ---- MinMaxScaler.fit ----
    scaler_t<T> sp;
    memset(&sp, 0, sizeof(sp));
    {
        const bool known = known_min != nullptr && known_max != nullptr;  // this rank's extrema came with the planes
        if (n > 0 && !known) {
            hipLaunchKernelGGL((km_minmax<T>), dim3(nblk, F), dim3(KM_THREADS), 0, st, pl, n, d_mm, d_mm + (size_t)nblk * F);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_mm, sizeof(T) * 2 * (size_t)nblk * F, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, rs_sync(ctx));
        }
        const T *hmn = (const T *)ctx->h_pin, *hmx = hmn + (size_t)nblk * F;
        double mm[2 * RSSEG_MAX_FEATURES + RSSEG_MAX_RANKS];
        for (int f = 0; f < F; f++) {
            T mn = (T)INFINITY, mx = (T)-INFINITY;
            if (n > 0 && known) {
                mn = (T)known_min[f];
                mx = (T)known_max[f];
            } else if (n > 0)
                for (int b = 0; b < nblk; b++) {
                    mn = std::min(mn, hmn[(size_t)f * nblk + b]);
                    mx = std::max(mx, hmx[(size_t)f * nblk + b]);
                }
            mm[f] = -(double)mn;  // MAX-reduce of the negated minimum
            mm[F + f] = (double)mx;
        }
        for (int r = 0; r < ctx->world; r++) mm[2 * F + r] = r == ctx->rank ? (double)n : 0.0;   // exact below 2^53; MAX of {n, 0, ...}
        RSCHK(comm_allreduce_host(ctx, mm, 2 * F + ctx->world, RSSEG_F64, RSSEG_MAX));
        for (int r = 0; r < ctx->world; r++) {
            n_all[r] = (int64_t)mm[2 * F + r];
            if (r < ctx->rank) offset += n_all[r];
            N += n_all[r];
        }
        if (N <= 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "kmeans: no pixels");
        if (N < k) return rs_fail(ctx, RSSEG_ERR_INVALID, "kmeans: n_samples=%lld should be >= n_clusters=%d", (long long)N, k);
        for (int f = 0; f < F; f++) {
            volatile T mn = (T)(-mm[f]), mx = (T)mm[F + f];
            volatile T range = mx - mn;
            if (range < (T)10 * teps<T>::v) range = (T)1;  // _handle_zeros_in_scale
            volatile T sc = (T)1 / range;
            volatile T t = mn * sc;
            volatile T mv = (T)0 - t;
            sp.scale[f] = sc;
            sp.minv[f] = mv;
        }
    }
    HIPCHK(ctx, hipMemcpyAsync(d_sp, &sp, sizeof(sp), hipMemcpyHostToDevice, st));

    // ---- everything from here to the final labels is driven from device-resident state (km_state): the host enqueues
    // sweeps, small control kernels and (world > 1) stream-ordered all-reduces, and waits only for the Lloyd loop's
    // convergence flag and for the final read-back ----
    mt19937 rng(seed);
    T C[RSSEG_MAX_CLUSTERS][RSSEG_MAX_FEATURES];
    int64_t init_idx[RSSEG_MAX_CLUSTERS];
    int last_rank = 0;
    for (int rk = 0; rk < ctx->world; rk++)
        if (n_all[rk] > 0) last_rank = rk;
    // values that cross ranks: in the communication buffer when world > 1 (the hook reduces them in place), else in the workspace
    const size_t x_need = sizeof(long long) * std::max<size_t>({(size_t)2 * M, (size_t)3 * F, (size_t)2 * F + 2 * (size_t)ctx->world,
                                                                (size_t)2 * KPP_MAXL * RSSEG_MAX_RANKS + KPP_MAXL * (1 + RSSEG_MAX_FEATURES) + 64});
    if (ctx->comm_on && x_need > ctx->comm_bytes) return rs_fail(ctx, RSSEG_ERR_COMM, "kmeans: comm buffer too small (%zu > %zu)", x_need, ctx->comm_bytes);
    long long *d_x = ctx->comm_on ? (long long *)ctx->d_comm : d_red;
    auto dev_allreduce = [&](int64_t byte_off, int64_t count, int dtype, int op) -> int {   // stream-ordered: no staging copy, no host synchronisation
        if (!ctx->comm_on) return RSSEG_OK;
        const auto t0c = std::chrono::steady_clock::now();
        const int rc = ctx->allreduce(ctx->comm_user, byte_off, count, dtype, op);
        if (rc != 0) return rs_fail(ctx, RSSEG_ERR_COMM, "all-reduce hook returned %d", rc);
        if (ctx->prof_on) {
            prof_entry &e = ctx->prof["allreduce"];
            e.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0c).count();
            e.launches++;
        }
        return RSSEG_OK;
    };
    {
        // The first k-means++ centre (RandomState.choice over N equal weights: the first draw of the generator) is known as
        // soon as N is; the uniform(size=L) draws of the later rounds follow in the generator's order.
        km_state<T> hs;
        memset(&hs, 0, sizeof(hs));
        const double u0 = rng.random_sample();
        init_idx[0] = uniform_choice(N, std::is_same<T, float>::value ? RSSEG_F32 : RSSEG_F64, u0);
        if (init_idx[0] >= N) init_idx[0] = N - 1;
        for (int i = 0; i < (k - 1) * L; i++) hs.uniforms[i] = rng.random_sample();
        hs.init_idx[0] = init_idx[0];
        hs.max_iter = max_iter;
        for (int r = 0; r < ctx->world; r++) hs.n_all[r] = n_all[r];
        hs.N = N; hs.offset = offset; hs.n_local = n; hs.nchunks = nchunks;
        hs.rank = ctx->rank; hs.world = ctx->world; hs.k = k; hs.F = F; hs.L = L; hs.last_rank = last_rank;
        hs.tol_in = tol_in;
        static_assert(sizeof(hs) <= UP_SLOT, "upload slot too small");
        HIPCHK(ctx, upload(d_lst, &hs, sizeof(hs)));
    }
    kpp_sample_args *d_sa = (kpp_sample_args *)(ws + o_sa);
    // ---- X.mean(axis=0): exact sums; the first centre's row rides on the same all-reduce (the owner gathers the scaled,
    // NOT yet centred row — the scaler on the device still has mean 0) ----
    {
        const int64_t li0 = init_idx[0] - offset;
        const bool own0 = li0 >= 0 && li0 < n;
        if (n > 0) {
            {
                prof_scope ps(ctx, "moment");
                hipLaunchKernelGGL((km_moment<T>), dim3(nblk, F), dim3(KM_THREADS), 0, st, pl, n, d_sp, d_mom);
            }
            if (own0) hipLaunchKernelGGL((km_gather_row<T>), dim3(1), dim3(64), 0, st, pl, F, li0, d_sp, d_row);
        }
        if (n > 0) hipLaunchKernelGGL(km_reduce_cols, dim3(F), dim3(KM_THREADS), 0, st, (const long long *)d_mom, (int64_t)nblk, d_x, (const int *)nullptr, 1, 0);
        hipLaunchKernelGGL((kc_mean_pack<T>), dim3(1), dim3(64), 0, st, (const km_state<T> *)d_lst, (const T *)d_row, own0 ? 1 : 0, d_x);
        HIPCHK(ctx, hipGetLastError());
        RSCHK(dev_allreduce(0, 3 * F, RSSEG_I64, RSSEG_SUM));
        hipLaunchKernelGGL((kc_mean_apply<T>), dim3(1), dim3(KM_THREADS), 0, st, d_lst, d_sp, (const long long *)d_x, d_cand);
        HIPCHK(ctx, hipGetLastError());
    }
    if (info)
        for (int f = 0; f < F; f++) {
            info->scale[f] = (double)sp.scale[f];
            info->min[f] = (double)sp.minv[f];
        }

    // ---- k-means++ (_kmeans.py:213-270) ----
    // fetch scaled+centred rows of global pixel indices; every rank ends up with all rows (empty-cluster relocation only)
    auto fetch_rows = [&](const int64_t *gidx, int cnt, T rows[][RSSEG_MAX_FEATURES]) -> int {
        double buf[KPP_MAXL * RSSEG_MAX_FEATURES];
        memset(buf, 0, sizeof(buf));
        for (int l = 0; l < cnt; l++) {
            int64_t li = gidx[l] - offset;
            if (li >= 0 && li < n) {
                hipLaunchKernelGGL((km_gather_row<T>), dim3(1), dim3(64), 0, st, pl, F, li, d_sp, d_row);
                HIPCHK(ctx, hipGetLastError());
                HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_row, sizeof(T) * F, hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, rs_sync(ctx));
                for (int f = 0; f < F; f++) buf[l * F + f] = (double)((const T *)ctx->h_pin)[f];
            }
        }
        RSCHK(comm_allreduce_host(ctx, buf, (int64_t)cnt * F, RSSEG_F64, RSSEG_SUM));
        for (int l = 0; l < cnt; l++)
            for (int f = 0; f < F; f++) rows[l][f] = (T)buf[l * F + f];
        return RSSEG_OK;
    };
    // first sweep: the potential of the first centre per chunk (row 0) and the np.var numerators (rows 1..F)
    if (n > 0) {
        prof_scope ps(ctx, "kpp");
        launch_kpp<T>(ctx, nchunks, pl, F, n, d_sp, d_cand, d_cc, 1, 0, d_closest, (unsigned long long *)d_part);
    }
    HIPCHK(ctx, hipGetLastError());
    {   // np.var(X, axis=0) and every rank's total of row 0 in ONE all-reduce: x = [F limb pairs][world limb pairs]
        HIPCHK(ctx, hipMemsetAsync(d_x, 0, sizeof(long long) * (2 * (size_t)F + 2 * (size_t)ctx->world), st));
        if (n > 0) {
            hipLaunchKernelGGL(km_reduce_cols, dim3(F), dim3(KM_THREADS), 0, st, (const long long *)d_part + nchunks, nchunks, d_x, (const int *)nullptr, 1, 0);
            hipLaunchKernelGGL(km_reduce_cols, dim3(1), dim3(KM_THREADS), 0, st, (const long long *)d_part, nchunks, d_x + 2 * F, (const int *)nullptr, 1, ctx->rank);
        }
        HIPCHK(ctx, hipGetLastError());
        RSCHK(dev_allreduce(0, 2 * F + 2 * ctx->world, RSSEG_I64, RSSEG_SUM));
        hipLaunchKernelGGL((kc_tol<T>), dim3(1), dim3(KM_THREADS), 0, st, d_lst, (const long long *)d_x);
        HIPCHK(ctx, hipGetLastError());
    }
    // exchange regions of a round inside x: the sampled rows at byte 0, the candidates' potentials behind them
    constexpr size_t LIM_OFF = ((sizeof(double) * KPP_MAXL * (1 + RSSEG_MAX_FEATURES)) + 511) & ~(size_t)511;
    double *d_xbuf = (double *)d_x;
    long long *d_lim = (long long *)((char *)d_x + LIM_OFF);
    for (int c = 1; c < k; c++) {
        // centre c-1 is pending: its distances are folded into the closest plane by this round's sweep, and by
        // km_kpp_chunkq for the chunks the samples fall into.  (The previous round's choice is made at the head of this kernel.)
        hipLaunchKernelGGL((kc_targets<T>), dim3(1), dim3(KM_THREADS), 0, st, d_lst, c, c > 1 ? c - 1 : 0, (const unsigned long long *)d_part, d_cand, d_sa,
                           d_xbuf, d_lim);
        // xbuf[l] = {global pixel index, its F scaled+centred values}: filled by the rank that owns the pixel, zeros
        // elsewhere, so ONE sum all-reduce hands every rank both the sampled indices and the candidate rows
        if (n > 0) {
            hipLaunchKernelGGL((km_kpp_chunkq<T>), dim3((unsigned)(CHUNK / KM_THREADS), L), dim3(KM_THREADS), 0, st, pl, F, d_sp, d_cand, d_cc,
                               (const T *)d_closest, (const kpp_sample_args *)d_sa, d_qv);
            hipLaunchKernelGGL((km_kpp_sample<T>), dim3(L), dim3(KM_THREADS), 0, st, pl, F, d_sp, (const unsigned long long *)d_qv, (const kpp_sample_args *)d_sa, d_xbuf);
        }
        HIPCHK(ctx, hipGetLastError());
        RSCHK(dev_allreduce(0, (int64_t)L * (1 + RSSEG_MAX_FEATURES), RSSEG_F64, RSSEG_SUM));
        hipLaunchKernelGGL((kc_cands<T>), dim3(1), dim3(KM_THREADS), 0, st, d_lst, (const double *)d_xbuf, d_cand);
        if (n > 0) {
            prof_scope ps(ctx, "kpp");
            launch_kpp<T>(ctx, nchunks, pl, F, n, d_sp, d_cand, d_cc, L, c > 1 ? 2 : 1, d_closest, (unsigned long long *)d_part);
        }
        HIPCHK(ctx, hipGetLastError());
        // potentials of all L candidates with ONE all-reduce: slot [l][rank] = this rank's chunk-sum total
        if (n > 0) hipLaunchKernelGGL(km_reduce_cols, dim3(L), dim3(KM_THREADS), 0, st, (const long long *)d_part, nchunks, d_lim, (const int *)nullptr, ctx->world, ctx->rank);
        RSCHK(dev_allreduce((int64_t)LIM_OFF, 2 * (int64_t)L * ctx->world, RSSEG_I64, RSSEG_SUM));
    }
    if (k > 1) {
        hipLaunchKernelGGL((kc_pick<T>), dim3(1), dim3(KM_THREADS), 0, st, d_lst, k - 1, (const long long *)d_lim);
        HIPCHK(ctx, hipGetLastError());
    }
    const double t_init = now_ms();
    T tol = (T)0;   // known on the host only after the state has come back (empty-cluster path, info)

    // ---- Lloyd (_kmeans_single_lloyd) ----
    // accumulator copies: 32 when they fit in ~48 KB (3 workgroups per CU), fewer for large k * F
    int ncopies = KM_COPIES;
    while (ncopies > 1 && sizeof(long long) * (size_t)ncopies * km_copy_stride(KMAX, F) > 48 * 1024) ncopies >>= 1;
    const size_t lds = sizeof(long long) * (size_t)ncopies * km_copy_stride(KMAX, F);
    if (lds > 150 * 1024) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "kmeans: F=%d, k=%d needs %zu B of LDS", F, k, lds);
    // one E-step (+ per-cluster sums when `update`).  from_state: the centres come from the device state (kl_prepare builds
    // their transposed copy and norms; `done`-guarded unless forced); otherwise from the host array C.
    // track: the sweep reads the previous labels of the uint8 working plane, counts the changes and writes the new ones.
    // The device-resident loop runs its update sweeps UNTRACKED (r04): "no label changed" only lets sklearn skip the final
    // E-step — with centres that are exact integer sums / counts, equal labels give bit-equal centres, so the shift is
    // exactly 0 <= tol and the loop ends in the SAME iteration through the tolerance rule, and the final E-step it then runs
    // reproduces those labels (they are argmin of the same centres): same labels, same n_iter, one more E-step in the
    // strictly converged case, and every update sweep loses its read-modify-write plane (62 -> 60 B/px and a read-only
    // stream: 2.9 -> 2.6 ms at 16384^2).  The host-side loop (empty clusters: relocation makes the centres depend on more
    // than the labels) keeps tracking.
    auto run_lloyd = [&](bool update, bool from_state, bool force, int32_t *out32 = nullptr, bool track = true) -> int {
        if (from_state) {
            // d_cen already holds the transposed centres of the state: kl_prepare before the first iteration, kl_update after each
        } else {
            // upload centres (transposed) and their squared norms (fma chain in T, row_norms of centres)
            T cenT[RSSEG_MAX_FEATURES * RSSEG_MAX_CLUSTERS + RSSEG_MAX_CLUSTERS];  // [MAX_FEATURES][KMAX] centres, then KMAX norms
            T *csq = cenT + (size_t)KMAX * RSSEG_MAX_FEATURES;
            memset(cenT, 0, sizeof(cenT));
            for (int j = 0; j < k; j++) {
                T a = (T)0;
                for (int f = 0; f < F; f++) {
                    cenT[f * KMAX + j] = C[j][f];
                    a = std::is_same<T, float>::value ? (T)fmaf((float)C[j][f], (float)C[j][f], (float)a) : (T)std::fma((double)C[j][f], (double)C[j][f], (double)a);
                }
                csq[j] = a;
            }
            static_assert(sizeof(cenT) <= UP_SLOT, "upload slot too small");
            HIPCHK(ctx, upload(d_cen, cenT, sizeof(T) * ((size_t)KMAX * RSSEG_MAX_FEATURES + KMAX)));
        }
        const int *dflag = (from_state && !force) ? &d_lst->done : nullptr;
        if (n > 0) {
            {
                prof_scope ps(ctx, "lloyd");
                const size_t l2 = update ? lds : 0;
                uint8_t *lab_arg = (update && !track) ? nullptr : d_lab;
                int lrc;
                switch (KMAX) {
                case 8: lrc = launch_lloyd<T, 8>(ctx, update, nchunks, l2, pl, F, k, n, d_sp, d_cen, d_csq, lab_arg, d_part, ncopies, dflag, out32); break;
                case 16: lrc = launch_lloyd<T, 16>(ctx, update, nchunks, l2, pl, F, k, n, d_sp, d_cen, d_csq, lab_arg, d_part, ncopies, dflag, out32); break;
                case 32: lrc = launch_lloyd<T, 32>(ctx, update, nchunks, l2, pl, F, k, n, d_sp, d_cen, d_csq, lab_arg, d_part, ncopies, dflag, out32); break;
                default: lrc = launch_lloyd<T, 64>(ctx, update, nchunks, l2, pl, F, k, n, d_sp, d_cen, d_csq, lab_arg, d_part, ncopies, dflag, out32); break;
                }
                if (lrc != RSSEG_OK) return lrc;
            }
            HIPCHK(ctx, hipGetLastError());
        }
        return RSSEG_OK;
    };

    bool strict = false;
    int it = 0, relocated = 0;
    std::vector<long long> red((size_t)2 * M);

    // What _kmeans_single_lloyd does between two E-steps, on the host: used from the iteration in which a cluster ran
    // empty (the device loop hands over with that iteration's reduced sums in `red`).  Returns 1 when the loop ends.
    auto host_finish_iteration = [&]() -> int {
        int64_t cnt[RSSEG_MAX_CLUSTERS];
        int n_empty = 0;
        for (int j = 0; j < k; j++) {
            cnt[j] = (int64_t)limbs(red[2 * (KMAX * F + j)], red[2 * (KMAX * F + j) + 1]);
            if (cnt[j] == 0) n_empty++;
        }
        const int64_t changed = (int64_t)limbs(red[2 * (KMAX * F + KMAX)], red[2 * (KMAX * F + KMAX) + 1]);
        // exact per-cluster sums as 128-bit integers
        std::vector<i128> Sv((size_t)RSSEG_MAX_CLUSTERS * RSSEG_MAX_FEATURES);
        auto S = [&](int j, int f) -> i128 & { return Sv[(size_t)j * RSSEG_MAX_FEATURES + f]; };
        for (int j = 0; j < k; j++)
            for (int f = 0; f < F; f++) S(j, f) = limbs(red[2 * (j * F + f)], red[2 * (j * F + f) + 1]);
        if (n_empty > 0) {
            // _relocate_empty_clusters_dense: the e-th empty cluster takes the e-th farthest pixel
            taken_t tk;
            tk.n = 0;
            T dmax_all = (T)0;
            int e = 0;
            for (int j = 0; j < k && e < n_empty; j++) {
                if (cnt[j] != 0) continue;
                double best[2] = {-1.0, 0.0};
                long long bidx = 0x7fffffffffffffffLL;
                if (n > 0) {
                    T *pd = (T *)d_part;
                    long long *pi = d_part + nchunks;  // after the distances (sizeof(T) <= 8)
                    switch (KMAX) {
                    case 8: launch_farthest<T, 8>(ctx, nchunks, pl, F, n, offset, d_sp, d_cen, d_lab, tk, pd, pi); break;
                    case 16: launch_farthest<T, 16>(ctx, nchunks, pl, F, n, offset, d_sp, d_cen, d_lab, tk, pd, pi); break;
                    case 32: launch_farthest<T, 32>(ctx, nchunks, pl, F, n, offset, d_sp, d_cen, d_lab, tk, pd, pi); break;
                    default: launch_farthest<T, 64>(ctx, nchunks, pl, F, n, offset, d_sp, d_cen, d_lab, tk, pd, pi); break;
                    }
                    HIPCHK(ctx, hipGetLastError());
                    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_part, sizeof(long long) * 2 * (size_t)nchunks, hipMemcpyDeviceToHost, st));
                    HIPCHK(ctx, rs_sync(ctx));
                    const T *hd = (const T *)ctx->h_pin;
                    const long long *hi = (const long long *)ctx->h_pin + nchunks;
                    for (int64_t c2 = 0; c2 < nchunks; c2++)
                        if ((double)hd[c2] > best[0] || ((double)hd[c2] == best[0] && hi[c2] < bidx)) { best[0] = (double)hd[c2]; bidx = hi[c2]; }
                }
                double gmax = best[0];
                RSCHK(comm_allreduce_host(ctx, &gmax, 1, RSSEG_F64, RSSEG_MAX));
                double gidx = (best[0] == gmax && bidx != 0x7fffffffffffffffLL) ? (double)bidx : 9.0e18;
                RSCHK(comm_allreduce_host(ctx, &gidx, 1, RSSEG_F64, RSSEG_MIN));
                if (e == 0) dmax_all = (T)gmax;
                if (dmax_all == (T)0 || gidx >= 9.0e18) break;  // np.max(distances) == 0: relocation is pointless
                const int64_t far = (int64_t)gidx;
                tk.idx[tk.n++] = far;
                T rows[KPP_MAXL][RSSEG_MAX_FEATURES];
                RSCHK(fetch_rows(&far, 1, rows));
                double oldlab = 0.0;
                if (far >= offset && far < offset + n) {
                    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_lab + (far - offset), 1, hipMemcpyDeviceToHost, st));
                    HIPCHK(ctx, rs_sync(ctx));
                    oldlab = (double)((const uint8_t *)ctx->h_pin)[0];
                }
                RSCHK(comm_allreduce_host(ctx, &oldlab, 1, RSSEG_F64, RSSEG_SUM));
                const int oj = (int)oldlab;
                for (int f = 0; f < F; f++) {
                    const i128 q = (i128)llrint((double)rows[0][f] * 1099511627776.0);
                    S(oj, f) -= q;
                    S(j, f) = q;
                }
                cnt[j] = 1;
                cnt[oj] -= 1;
                relocated++;
                e++;
            }
        }
        // _average_centers
        T Cnew[RSSEG_MAX_CLUSTERS][RSSEG_MAX_FEATURES];
        int argmax_w = 0;
        for (int j = 1; j < k; j++)
            if (cnt[j] > cnt[argmax_w]) argmax_w = j;
        for (int j = 0; j < k; j++) {
            if (cnt[j] <= 0) continue;
            volatile T w = (T)cnt[j];
            volatile T alpha = (T)(1.0 / (double)w);
            for (int f = 0; f < F; f++) {
                volatile T sT = fixed_to_T<T>(S(j, f));
                volatile T cnew = sT * alpha;
                Cnew[j][f] = cnew;
            }
        }
        for (int j = 0; j < k; j++)
            if (cnt[j] <= 0)
                for (int f = 0; f < F; f++) Cnew[j][f] = Cnew[argmax_w][f];
        T shift2[RSSEG_MAX_CLUSTERS];
        for (int j = 0; j < k; j++) {
            const T *a = Cnew[j], *b = C[j];
            volatile T result = (T)0;
            const int n4 = F / 4, rem = F % 4;
            for (int g = 0; g < n4; g++) {
                volatile T d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2], d3 = a[3] - b[3];
                volatile T t0 = d0 * d0, t1 = d1 * d1, t2 = d2 * d2, t3 = d3 * d3;
                volatile T g1 = t0 + t1;
                volatile T g2 = g1 + t2;
                volatile T g4 = g2 + t3;
                result = result + g4;
                a += 4;
                b += 4;
            }
            for (int r = 0; r < rem; r++) {
                volatile T d = a[r] - b[r];
                volatile T t = d * d;
                result = result + t;
            }
            volatile T sh = std::sqrt((T)result);
            volatile T s2 = sh * sh;
            shift2[j] = s2;
        }
        memcpy(C, Cnew, sizeof(C));
        it++;
        if (changed == 0) {
            strict = true;
            return 1;
        }
        const T tot = np_pairwise_sum<T>(shift2, k);
        if (tot <= tol) return 1;
        return it >= max_iter ? 1 : 0;
    };

    // ---- device-resident loop: batches of speculatively enqueued iterations, one look at the state per batch ----
    long long *d_sums = ctx->comm_on ? (long long *)ctx->d_comm : d_red;   // where km_reduce_cols leaves the limb sums
    if (ctx->comm_on && sizeof(long long) * 2 * (size_t)M > ctx->comm_bytes) return rs_fail(ctx, RSSEG_ERR_COMM, "kmeans: comm buffer too small");
    bool finished = false, host_mode = false;
    int batch = 4;
    km_state<T> *h_state = (km_state<T> *)ctx->h_pin;
    hipLaunchKernelGGL((kl_prepare<T>), dim3(1), dim3(KM_THREADS), 0, st, (const lloyd_state<T> *)d_lst, k, F, KMAX, d_cen, 1);
    HIPCHK(ctx, hipGetLastError());
    while (!finished && !host_mode) {
        const int todo = std::min(batch, max_iter - it);
        for (int b = 0; b < todo; b++) {
            RSCHK(run_lloyd(true, true, false, nullptr, false));
            hipLaunchKernelGGL(km_reduce_cols, dim3(M), dim3(KM_THREADS), 0, st, (const long long *)d_part, nchunks, d_sums, (const int *)&d_lst->done, 1, 0);
            HIPCHK(ctx, hipGetLastError());
            if (ctx->comm_on) {   // stream-ordered: no staging copy, no host synchronisation (include/rsseg.h, rsseg_allreduce_fn)
                const auto t0c = std::chrono::steady_clock::now();
                if (n <= 0) HIPCHK(ctx, hipMemsetAsync(d_sums, 0, sizeof(long long) * 2 * M, st));
                const int rc = ctx->allreduce(ctx->comm_user, 0, 2 * (int64_t)M, RSSEG_I64, RSSEG_SUM);
                if (rc != 0) return rs_fail(ctx, RSSEG_ERR_COMM, "all-reduce hook returned %d", rc);
                if (ctx->prof_on) {
                    prof_entry &e = ctx->prof["allreduce"];
                    e.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0c).count();
                    e.launches++;
                }
            }
            hipLaunchKernelGGL((kl_update<T>), dim3(1), dim3(KM_THREADS), 0, st, d_lst, k, F, KMAX, (const long long *)d_sums, d_redsave, d_cen);
            HIPCHK(ctx, hipGetLastError());
        }
        // one look at the state (all of it: when the loop has ended, the centres, tolerance, means and seeds are here already)
        HIPCHK(ctx, hipMemcpyAsync(h_state, d_lst, sizeof(km_state<T>), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, rs_sync(ctx));
        const int it_new = h_state->it, done = h_state->done;
        prof_retag(ctx, "lloyd", todo - (it_new - it) - (done == 3 ? 1 : 0), "lloyd_noop");   // launches that returned at once
        it = it_new;
        if (done == 1 || done == 2 || done == 4) {
            strict = done == 1;
            finished = true;
        } else if (done == 3) {
            host_mode = true;
        } else if (it >= max_iter) {
            finished = true;
        }
        batch = std::max(2, it / 3);   // short fits are not over-enqueued, long ones are looked at ever more rarely
    }
    {   // the state the device loop ended with: centres, tolerance, means, seeds (the host reports them, or continues from them)
        if (host_mode) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin + sizeof(km_state<T>), d_redsave, sizeof(long long) * 2 * M, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, rs_sync(ctx));
        }
        memcpy(C, h_state->C, sizeof(C));
        tol = h_state->tol;
        for (int f = 0; f < F; f++) sp.mean[f] = h_state->mean[f];
        for (int j = 0; j < k; j++) init_idx[j] = h_state->init_idx[j];
        if (host_mode) memcpy(red.data(), ctx->h_pin + sizeof(km_state<T>), sizeof(long long) * 2 * M);
        if (info) {
            info->tol = (double)tol;
            for (int f = 0; f < F; f++) info->mean[f] = (double)sp.mean[f];
            for (int j = 0; j < k; j++) info->init_indices[j] = init_idx[j];
        }
    }
    if (host_mode) {
        // the iteration in which a cluster ran empty: its E-step and its (all-reduced) sums exist, the centres the kernels
        // used (d_cen) are still those of that E-step — finish it here, then iterate on the host.  The untracked sweeps wrote
        // no labels: the relocation (distance to the ASSIGNED centre) and the tracked iterations that follow need them, so
        // that E-step is repeated once, labels only.  (Its change count is known to be non-zero: a cluster that holds
        // pixels in one iteration and none in the next has lost them.)
        RSCHK(run_lloyd(false, true, true));
        int end = host_finish_iteration();
        if (end < 0) return end;
        while (!end) {
            RSCHK(run_lloyd(true, false, true));
            if (n > 0) {
                hipLaunchKernelGGL(km_reduce_cols, dim3(M), dim3(KM_THREADS), 0, st, (const long long *)d_part, nchunks, d_red, (const int *)nullptr, 1, 0);
                HIPCHK(ctx, hipGetLastError());
                HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_red, sizeof(long long) * 2 * M, hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, rs_sync(ctx));
                        memcpy(red.data(), ctx->h_pin, sizeof(long long) * 2 * M);
            } else {
                std::fill(red.begin(), red.end(), 0ll);
            }
            RSCHK(comm_allreduce_host(ctx, red.data(), 2 * M, RSSEG_I64, RSSEG_SUM));
            end = host_finish_iteration();
            if (end < 0) return end;
        }
    }
    // the final E-step (labels of the final centres; skipped when the last iteration changed no label) writes the caller's
    // int32 plane itself when its kernel can (F <= 32, 16-byte aligned plane); otherwise the uint8 plane is widened
    const bool direct = !strict && n > 0 && ((uintptr_t)d_labels & 15) == 0 && (F <= 8 || (F <= 32 && !(std::is_same<T, double>::value && KMAX == 64)));
    if (!strict) RSCHK(run_lloyd(false, false, true, direct ? d_labels : nullptr));
    if (n > 0 && !direct) {
        prof_scope ps(ctx, "labels");
        hipLaunchKernelGGL(km_labels_out, dim3((unsigned)std::min<int64_t>(4096, ceil_div64(n, KM_THREADS))), dim3(KM_THREADS), 0, st,
                           d_lab, d_labels, n);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, rs_sync(ctx));
    if (centers_out)
        for (int j = 0; j < k; j++)
            for (int f = 0; f < F; f++) {
                volatile T v = C[j][f] + sp.mean[f];  // best_centers += X_mean
                centers_out[j * F + f] = (double)v;
            }
    if (info) {
        info->n_iter = it;
        info->relocated = relocated;
        info->ms_init = t_init - t_start;
        info->ms_lloyd = now_ms() - t_init;
    }
    return RSSEG_OK;
}

}  // namespace

static int kmeans_entry(rsseg_ctx *ctx, const void *const *d_planes, int F, int dtype, int64_t n_local, int k, uint32_t seed, int max_iter,
                        double tol, int32_t *d_labels, double *centers, rsseg_kmeans_info *info, const double *local_min,
                        const double *local_max)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_planes || F < 1) return rs_fail(ctx, RSSEG_ERR_INVALID, "kmeans: no feature planes");
    if (k < 1) return rs_fail(ctx, RSSEG_ERR_INVALID, "kmeans: n_clusters=%d must be >= 1", k);
    // capacities of the kernels, not limits of the reference: reported as UNSUPPORTED (never as an empty result)
    if (F > RSSEG_MAX_FEATURES) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "kmeans: %d feature planes, the kernels take at most %d", F, RSSEG_MAX_FEATURES);
    if (k > RSSEG_MAX_CLUSTERS) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "kmeans: n_clusters=%d, the kernels take at most %d", k, RSSEG_MAX_CLUSTERS);
    if (n_local < 0 || (n_local > 0 && !d_labels)) return rs_fail(ctx, RSSEG_ERR_INVALID, "kmeans: bad n_local / labels");
    if (max_iter < 1) return rs_fail(ctx, RSSEG_ERR_INVALID, "kmeans: max_iter must be >= 1");
    if (dtype != RSSEG_F32 && dtype != RSSEG_F64) return rs_fail(ctx, RSSEG_ERR_INVALID, "kmeans: dtype must be RSSEG_F32 or RSSEG_F64");
    if ((local_min == nullptr) != (local_max == nullptr)) return rs_fail(ctx, RSSEG_ERR_INVALID, "kmeans: local_min and local_max go together");
    if (local_min && n_local > 0)
        for (int f = 0; f < F; f++)
            if (!(local_min[f] <= local_max[f])) return rs_fail(ctx, RSSEG_ERR_INVALID, "kmeans: local extrema of plane %d are not ordered numbers", f);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (info) memset(info, 0, sizeof(*info));
    if (dtype == RSSEG_F32) return kmeans_fit<float>(ctx, d_planes, F, n_local, k, seed, max_iter, tol, d_labels, centers, info, local_min, local_max);
    return kmeans_fit<double>(ctx, d_planes, F, n_local, k, seed, max_iter, tol, d_labels, centers, info, local_min, local_max);
}

extern "C" int rsseg_kmeans_fit_predict(rsseg_ctx *ctx, const void *const *d_planes, int F, int dtype, int64_t n_local, int k,
                                        uint32_t seed, int max_iter, double tol, int32_t *d_labels, double *centers,
                                        rsseg_kmeans_info *info)
{
    return kmeans_entry(ctx, d_planes, F, dtype, n_local, k, seed, max_iter, tol, d_labels, centers, info, nullptr, nullptr);
}

extern "C" int rsseg_kmeans_fit_predict_mm(rsseg_ctx *ctx, const void *const *d_planes, int F, int dtype, int64_t n_local, int k,
                                           uint32_t seed, int max_iter, double tol, int32_t *d_labels, double *centers,
                                           rsseg_kmeans_info *info, const double *local_min, const double *local_max)
{
    return kmeans_entry(ctx, d_planes, F, dtype, n_local, k, seed, max_iter, tol, d_labels, centers, info, local_min, local_max);
}

// host-only helper (no GPU): the draws sklearn takes from RandomState(seed) for k-means++ on n samples.
// Exposed so the CPU test-suite can check the MT19937 / cumulative-probability restatement against NumPy.
extern "C" int rsseg_host_kmeans_draws(uint32_t seed, int64_t n, int dtype, int k, int64_t *center_id, double *uniforms)
{
    if (n < 1 || k < 1 || !center_id) return RSSEG_ERR_INVALID;
    mt19937 rng(seed);
    *center_id = uniform_choice(n, dtype, rng.random_sample());
    const int L = 2 + (int)std::log((double)k);
    if (uniforms)
        for (int i = 0; i < (k - 1) * L; i++) uniforms[i] = rng.random_sample();
    return RSSEG_OK;
}
