// K5-K8 — window operators of the level-2 features and of add_spatial_context:
//   bilinear upsample (cv2.resize INTER_LINEAR, indices.py:308), k x k box mean with float64 sums
//   (cv2.boxFilter / cv2.blur, indices.py:770-771, 537, 541), local variance / standard deviation
//   (indices.py:537-548), k x k morphology on uint8 (indices.py:421-433), 3x3 Sobel magnitude and Laplacian with
//   their global extrema (indices.py:472-480).  OpenCV semantics as restated in oracle/ref_np.py.
//
// K6-K8 are LDS-tiled separable stencils.  A 256-thread workgroup owns a tile of 256 columns x 32 output rows:
//   1. the (32 + 2R) x (256 + 8) input tile is staged in LDS with 16-byte loads (border rule applied while loading:
//      cv2.BORDER_REFLECT, BORDER_REFLECT_101, or edge replication for the morphology, whose out-of-image taps
//      never win);
//   2. a thread owns 4 columns x 8 output rows: for each of its 8 + 2R tile rows it reads 12 values (3 x ds_read_b128,
//      conflict-free), forms the 4 horizontal results of that row (row sums in float64 left to right; running
//      min / max; Sobel row terms) and keeps the last K rows of them in a register ring; every row from the K-th on
//      completes one output row (the K ring entries top to bottom) — the summation order fixed in oracle/ref_np.py, so the
//      float64 box sums stay bit-exact.
// HBM traffic is the plane once in, once out (halo re-reads are L2 hits); LDS traffic is 1.3 b128 reads per pixel.
// Rows form: every operator takes a plane of Hin rows and produces rows [y0, y1) of it, so a rank of a row-sharded
// raster passes its stripe plus R halo rows and gets exactly the rows of the un-sharded result.
#include <cmath>

#include "common.h"

#define BORDER_REPLICATE 2  // internal: erode / dilate (out-of-image taps never win == replicate the edge)

__device__ __forceinline__ int border_idx(int i, int n, int mode)
{
    // mode 0: BORDER_REFLECT (edge duplicated), mode 1: BORDER_REFLECT_101, mode 2: replicate
    if (n == 1) return 0;
    if (mode == BORDER_REPLICATE) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    while (i < 0 || i >= n) {
        if (i < 0) i = mode == 0 ? -i - 1 : -i;
        else i = mode == 0 ? 2 * n - 1 - i : 2 * n - 2 - i;
    }
    return i;
}

#define WG_X 64
#define WG_Y 4

// ---- tile geometry of the LDS stencils ---------------------------------------------------------------------------
#define TW 256                      // output columns per workgroup
#define TH 32                       // output rows per workgroup
#define CPT 4                       // columns per thread
#define RPT 8                       // output rows per thread
#define TPAD 4                      // tile columns left / right of the 256 (keeps every LDS read 16-byte aligned)
#define TSTRIDE (TW + 2 * TPAD)     // elements per tile row
#define WIN_MAXP 8

struct win_planes {
    const void *in[WIN_MAXP];
    void *out[WIN_MAXP];
};

// XCD-aware block -> tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2), so a
// row-major tile order puts horizontally adjacent tiles — which share the 128-byte lines their 4-column pads touch — on
// DIFFERENT XCDs, and the tile below 64 workgroups later.  Each XCD gets a contiguous run of tiles instead: neighbours in x
// follow each other on one XCD, and the halo rows of the tile above are still in that XCD's L2.  grid.x = 8 * chunk.
struct tile_map {
    int ntx, nt, chunk;
};
static inline tile_map make_tile_map(int nrows, int W)
{
    tile_map m;
    m.ntx = (W + TW - 1) / TW;
    m.nt = m.ntx * ((nrows + TH - 1) / TH);
    m.chunk = (m.nt + 7) / 8;
    return m;
}
__device__ __forceinline__ bool tile_of_block(const tile_map &m, int &x0, int &ty0)
{
    const int t = (int)(blockIdx.x & 7u) * m.chunk + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= m.chunk || t >= m.nt) return false;
    const int ty = t / m.ntx;
    x0 = (t - ty * m.ntx) * TW;
    ty0 = ty * TH;
    return true;
}

template <typename E> struct vec4;
template <> struct vec4<float> { typedef float4 type; };
template <> struct vec4<uint8_t> { typedef uint32_t type; };

// tile[(TH + 2R)][TSTRIDE]: tile row r holds input row border(yb - R + r), tile column c input column x0 - TPAD + c
// (columns further than R outside the image are never read by the stencil and are left zero)
template <typename E, int R>
__device__ __forceinline__ void load_tile(E *__restrict__ tile, const E *__restrict__ x, int Hin, int W, int mode, int x0, int yb)
{
    typedef typename vec4<E>::type V;
    constexpr int NR = TH + 2 * R, SLOTS = TSTRIDE / 4;
    const bool vec = (W & 3) == 0 && ((uintptr_t)x & (sizeof(V) - 1)) == 0;
    for (int s = threadIdx.x; s < NR * SLOTS; s += 256) {
        const int r = s / SLOTS, g = s - r * SLOTS;
        const int gc = x0 - TPAD + 4 * g;
        const E *row = x + (size_t)border_idx(yb - R + r, Hin, mode) * W;
        V v;
        if (vec && gc >= 0 && gc + 3 < W) {
            v = *reinterpret_cast<const V *>(row + gc);
        } else {
            E e[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = gc + q;
                e[q] = (c < -R || c >= W + R) ? (E)0 : row[border_idx(c, W, mode)];
            }
            memcpy(&v, e, sizeof(V));
        }
        *reinterpret_cast<V *>(tile + r * TSTRIDE + 4 * g) = v;
    }
}

__device__ __forceinline__ void read12(const float *p, float (&v)[12])
{
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1], c = reinterpret_cast<const float4 *>(p)[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}
__device__ __forceinline__ void read12(const uint8_t *p, int (&v)[12])
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    const uint32_t a = w[0], b = w[1], c = w[2];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        v[q] = (a >> (8 * q)) & 255u;
        v[4 + q] = (b >> (8 * q)) & 255u;
        v[8 + q] = (c >> (8 * q)) & 255u;
    }
}

__device__ __forceinline__ void store4(float *out, int W, int orow, int col, const float (&o)[4], bool vec)
{
    float *p = out + (size_t)orow * W + col;
    if (vec) *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
    else
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (col + q < W) p[q] = o[q];
}
__device__ __forceinline__ void store4(uint8_t *out, int W, int orow, int col, const int (&o)[4], bool vec)
{
    uint8_t *p = out + (size_t)orow * W + col;
    if (vec) *reinterpret_cast<uint32_t *>(p) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
    else
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (col + q < W) p[q] = (uint8_t)o[q];
}

// ---- box mean (optionally of x*x), float64 sums: rows left-to-right, then rows top-to-bottom ----
template <int K, bool SQ>
__global__ __launch_bounds__(256) void k6_box(win_planes pl, int Hin, int W, int y0, int nrows, int mode, tile_map tm)
{
    constexpr int R = K / 2;
    __shared__ __align__(16) float tile[(TH + 2 * R) * TSTRIDE];
    const float *x = (const float *)pl.in[blockIdx.z];
    float *out = (float *)pl.out[blockIdx.z];
    int x0, ty0;
    if (!tile_of_block(tm, x0, ty0)) return;
    load_tile<float, R>(tile, x, Hin, W, mode, x0, y0 + ty0);
    __syncthreads();
    const int cx = (threadIdx.x & 63) * CPT, ry = (threadIdx.x >> 6) * RPT;
    if (x0 + cx >= W || ty0 + ry >= nrows) return;
    const bool vec = (W & 3) == 0 && ((uintptr_t)out & 15) == 0;
    double ring[K][CPT];
#pragma unroll
    for (int j = 0; j < RPT + 2 * R; j++) {
        float v[12];
        read12(tile + (ry + j) * TSTRIDE + cx, v);
        if (SQ)
#pragma unroll
            for (int e = 0; e < 12; e++) v[e] = v[e] * v[e];
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            double rs = (double)v[TPAD - R + c];
#pragma unroll
            for (int d = 1; d < K; d++) rs = rs + (double)v[TPAD - R + c + d];
            ring[j % K][c] = rs;
        }
        if (j >= 2 * R) {
            const int i = j - 2 * R, orow = ty0 + ry + i;
            if (orow < nrows) {
                float o[CPT];
#pragma unroll
                for (int c = 0; c < CPT; c++) {
                    double acc = ring[i % K][c];
#pragma unroll
                    for (int dy = 1; dy < K; dy++) acc = acc + ring[(i + dy) % K][c];
                    o[c] = (float)(acc * (1.0 / (double)(K * K)));
                }
                store4(out, W, orow, x0 + cx, o, vec);
            }
        }
    }
}

// variance_scale_k / std_dev_scale_k: max(blur(x*x) - blur(x)^2, 0) [sqrt], BORDER_REFLECT_101 (indices.py:537-548)
template <int K, bool VAR>
__global__ __launch_bounds__(256) void k6_std(const float *__restrict__ x, int Hin, int W, int y0, int nrows, float *__restrict__ out, tile_map tm)
{
    constexpr int R = K / 2;
    __shared__ __align__(16) float tile[(TH + 2 * R) * TSTRIDE];
    int x0, ty0;
    if (!tile_of_block(tm, x0, ty0)) return;
    load_tile<float, R>(tile, x, Hin, W, 1, x0, y0 + ty0);
    __syncthreads();
    const int cx = (threadIdx.x & 63) * CPT, ry = (threadIdx.x >> 6) * RPT;
    if (x0 + cx >= W || ty0 + ry >= nrows) return;
    const bool vec = (W & 3) == 0 && ((uintptr_t)out & 15) == 0;
    double ring1[K][CPT], ring2[K][CPT];
    const double sc = 1.0 / (double)(K * K);
#pragma unroll
    for (int j = 0; j < RPT + 2 * R; j++) {
        float v[12], vv[12];
        read12(tile + (ry + j) * TSTRIDE + cx, v);
#pragma unroll
        for (int e = 0; e < 12; e++) vv[e] = v[e] * v[e];
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            double r1 = (double)v[TPAD - R + c], r2 = (double)vv[TPAD - R + c];
#pragma unroll
            for (int d = 1; d < K; d++) {
                r1 = r1 + (double)v[TPAD - R + c + d];
                r2 = r2 + (double)vv[TPAD - R + c + d];
            }
            ring1[j % K][c] = r1;
            ring2[j % K][c] = r2;
        }
        if (j >= 2 * R) {
            const int i = j - 2 * R, orow = ty0 + ry + i;
            if (orow < nrows) {
                float o[CPT];
#pragma unroll
                for (int c = 0; c < CPT; c++) {
                    double a1 = ring1[i % K][c], a2 = ring2[i % K][c];
#pragma unroll
                    for (int dy = 1; dy < K; dy++) {
                        a1 = a1 + ring1[(i + dy) % K][c];
                        a2 = a2 + ring2[(i + dy) % K][c];
                    }
                    const float mean = (float)(a1 * sc), mean_sq = (float)(a2 * sc);
                    const float mm = mean * mean;
                    float var = mean_sq - mm;
                    if (var < 0.f) var = 0.f;
                    o[c] = VAR ? var : sqrtf(var);
                }
                store4(out, W, orow, x0 + cx, o, vec);
            }
        }
    }
}

// erode (MODE 0: min), dilate (MODE 1: max), gradient (MODE 2: max - min) with a K x K rectangle
template <int K, int MODE>
__global__ __launch_bounds__(256) void k7_morph(const uint8_t *__restrict__ q, int Hin, int W, int y0, int nrows, uint8_t *__restrict__ out, tile_map tm)
{
    constexpr int R = K / 2;
    __shared__ __align__(16) uint8_t tile[(TH + 2 * R) * TSTRIDE];
    int x0, ty0;
    if (!tile_of_block(tm, x0, ty0)) return;
    load_tile<uint8_t, R>(tile, q, Hin, W, BORDER_REPLICATE, x0, y0 + ty0);
    __syncthreads();
    const int cx = (threadIdx.x & 63) * CPT, ry = (threadIdx.x >> 6) * RPT;
    if (x0 + cx >= W || ty0 + ry >= nrows) return;
    const bool vec = (W & 3) == 0 && ((uintptr_t)out & 3) == 0;
    int rmin[K][CPT], rmax[K][CPT];
#pragma unroll
    for (int j = 0; j < RPT + 2 * R; j++) {
        int v[12];
        read12(tile + (ry + j) * TSTRIDE + cx, v);
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            int mn = v[TPAD - R + c], mx = mn;
#pragma unroll
            for (int d = 1; d < K; d++) {
                const int t = v[TPAD - R + c + d];
                if (MODE != 1) mn = t < mn ? t : mn;
                if (MODE != 0) mx = t > mx ? t : mx;
            }
            rmin[j % K][c] = mn;
            rmax[j % K][c] = mx;
        }
        if (j >= 2 * R) {
            const int i = j - 2 * R, orow = ty0 + ry + i;
            if (orow < nrows) {
                int o[CPT];
#pragma unroll
                for (int c = 0; c < CPT; c++) {
                    int mn = rmin[i % K][c], mx = rmax[i % K][c];
#pragma unroll
                    for (int dy = 1; dy < K; dy++) {
                        const int a = rmin[(i + dy) % K][c], b = rmax[(i + dy) % K][c];
                        if (MODE != 1) mn = a < mn ? a : mn;
                        if (MODE != 0) mx = b > mx ? b : mx;
                    }
                    o[c] = MODE == 0 ? mn : (MODE == 1 ? mx : mx - mn);
                }
                store4(out, W, orow, x0 + cx, o, vec);
            }
        }
    }
}

// 3x3 Sobel magnitude (KIND 0) and cross Laplacian (KIND 1) of a uint8 plane, BORDER_REFLECT_101, value / 255 in float32.
// PASS 0 reduces the extrema of the rows [y0, y0 + nrows) into mm[0..1] (ordered keys, common.h) and writes nothing;
// PASS 1 recomputes the stencil and writes (value - sub) / den — the plane is read twice (1 B/px each) and written
// once instead of being written, re-read and re-written as float32.
template <int KIND, int PASS>
__global__ __launch_bounds__(256) void k8_filter(const uint8_t *__restrict__ q, int Hin, int W, int y0, int nrows, float *__restrict__ out,
                                                 float sub, float den, uint32_t *__restrict__ mm, tile_map tm)
{
    constexpr int R = 1;
    __shared__ __align__(16) uint8_t tile[(TH + 2 * R) * TSTRIDE];
    int x0, ty0;
    if (!tile_of_block(tm, x0, ty0)) return;   // whole workgroup: uniform
    load_tile<uint8_t, R>(tile, q, Hin, W, 1, x0, y0 + ty0);
    __syncthreads();
    const int cx = (threadIdx.x & 63) * CPT, ry = (threadIdx.x >> 6) * RPT;
    float lmn = INFINITY, lmx = -INFINITY;
    if (x0 + cx < W && ty0 + ry < nrows) {
        const bool vec = (W & 3) == 0 && ((uintptr_t)out & 15) == 0;
        int ra[3][CPT], rb[3][CPT];  // Sobel: horizontal difference / smoothing of a row; Laplacian: left + right / centre
#pragma unroll
        for (int j = 0; j < RPT + 2; j++) {
            int v[12];
            read12(tile + (ry + j) * TSTRIDE + cx, v);
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                const int l = v[3 + c], m = v[4 + c], r = v[5 + c];
                ra[j % 3][c] = KIND == 0 ? r - l : l + r;
                rb[j % 3][c] = KIND == 0 ? l + 2 * m + r : m;
            }
            if (j >= 2) {
                const int i = j - 2, orow = ty0 + ry + i;
                if (orow < nrows) {
                    float o[CPT];
#pragma unroll
                    for (int c = 0; c < CPT; c++) {
                        float val;
                        if (KIND == 0) {
                            const int gx = ra[i % 3][c] + 2 * ra[(i + 1) % 3][c] + ra[(i + 2) % 3][c];
                            const int gy = rb[(i + 2) % 3][c] - rb[i % 3][c];
                            const float sx = (float)gx / 255.0f, sy = (float)gy / 255.0f;
                            const float s2 = sx * sx + sy * sy;
                            val = sqrtf(s2);
                        } else {
                            const int s = rb[i % 3][c] + rb[(i + 2) % 3][c] + ra[(i + 1) % 3][c] - 4 * rb[(i + 1) % 3][c];
                            val = (float)s / 255.0f;
                        }
                        if (PASS == 0) {
                            if (x0 + cx + c < W) { lmn = fminf(lmn, val); lmx = fmaxf(lmx, val); }
                        } else {
                            const float dlt = val - sub;
                            o[c] = KIND == 0 ? val / den : dlt / den;
                        }
                    }
                    if (PASS == 1) store4(out, W, orow, x0 + cx, o, vec);
                }
            }
        }
    }
    if (PASS == 0) {
        // one commit per WORKGROUP (a per-wave test of the global slot makes 4 M waves read one address: 2.8 ms at
        // 16384^2); lanes without pixels carry +inf / -inf and never win
        __shared__ float smn[4], smx[4];
        lmn = wave_min(lmn);
        lmx = wave_max(lmx);
        if (lane_id() == 0) { smn[threadIdx.x >> 6] = lmn; smx[threadIdx.x >> 6] = lmx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3])), mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
            const uint32_t kmn = mm_key(mn), kmx = mm_key(mx);
            uint32_t *rs = mm + 16 * (blockIdx.x % RSSEG_MM_REPL);   // 64 replicas, one line each
            if (kmn < __builtin_nontemporal_load(&rs[0])) atomicMin(&rs[0], kmn);
            if (kmx > __builtin_nontemporal_load(&rs[1])) atomicMax(&rs[1], kmx);
        }
    }
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
struct resize_planes {
    const float *src[WIN_MAXP];
    float *dst[WIN_MAXP];
};

// A workgroup owns a strip of RS_W = 256 destination columns and R consecutive rows of it, one row per iteration: the
// horizontal taps and weights of a thread's pixel column are computed once instead of per pixel (the per-pixel form spent ~70
// vector instructions per pixel, a third of them the float64 coordinate arithmetic), the source row an output row shares with
// the next one is still in the CU's L1 / the XCD's L2, and a strip column stays on one XCD (blockIdx % 8 = tx % 8 when
// 8 | gx).  256 columns and not 64: source rows are not 128-byte aligned (16378 floats), so a 64-pixel segment touches three
// cache lines where two hold its data and the line at each end is fetched again by the neighbouring strip's XCD — 6.8 B/px
// read for 4 (PMC, r03); a 1 KB segment touches nine for eight.
#define RS_W 256
#define RS_AHEAD 8   // source rows requested together: 16 four-byte loads in flight per lane
// r04: the walk goes over SOURCE rows.  h(y) = src[y][sx] * a0 + src[y][sx1] * a1 is formed once per source row and lane
// (a destination row's lower row is the next destination row's upper row when the scale is near 1: 2 loads per pixel, not
// 4), and the taps of the next RS_AHEAD source rows are requested back to back before any of them is used — the r03 form
// had ONE destination row in flight per wave (four loads, wait, store: a memory round trip per row).  Every destination
// row py with floor(fy(py)) == y is then written as h(y) * (1 - fy) + h(y + 1) * fy with rows clamped to the image: the
// operations and operands of resize_px, so the same bits.
template <bool MM>
__global__ __launch_bounds__(256) void k5_resize(resize_planes pl, int sh, int sw, int dh, int dw, double scale_x, double scale_y, int src_row0,
                                                 int sh_local, int dst_row0, int dh_local, int gx, int R, int plane, uint32_t *__restrict__ mm)
{
    const float *__restrict__ src = pl.src[plane];
    float *__restrict__ dst = pl.dst[plane];
    float mn = INFINITY, mx = -INFINITY;
    const int tx = (int)(blockIdx.x % (unsigned)gx), by = (int)(blockIdx.x / (unsigned)gx);
    const int px = tx * RS_W + (int)threadIdx.x;
    if (px < dw) {
        // horizontal taps (the arithmetic of resize_px, hoisted)
        float fx = (float)(((double)px + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx = fx - (float)sx;
        if (sx < 0) { sx = 0; fx = 0.f; }
        if (sx >= sw - 1) { sx = sw - 1; fx = 0.f; }
        const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
        const float a0 = 1.0f - fx, a1 = fx;
        const int row_end = min((by + 1) * R, dh_local);
        int pyl = by * R;
        // vertical position of a destination row: floor and fraction, as resize_px computes them
        auto vpos = [&](int row_local, float &frac) -> int {
            float fy = (float)(((double)(row_local + dst_row0) + 0.5) * scale_y - 0.5);
            const int sy = (int)floorf(fy);
            frac = fy - (float)sy;
            return sy;
        };
        // a source row's address: clamped to the image (the border rule) and to the rows this stripe holds (only rows
        // requested ahead of need can fall outside; they are never used)
        const int lo = src_row0, hi = src_row0 + sh_local - 1;
        auto rowp = [&](int y) -> const float * {
            y = y < 0 ? 0 : (y > sh - 1 ? sh - 1 : y);
            y = y < lo ? lo : (y > hi ? hi : y);
            return src + (size_t)(y - src_row0) * sw;
        };
        float fcur;
        int sy_cur = pyl < row_end ? vpos(pyl, fcur) : 0;
        int ys = sy_cur;                                   // the source row h_cur belongs to
        float h_cur;
        {
            const float *r = rowp(ys);
            h_cur = r[sx] * a0 + r[sx1] * a1;
        }
        while (pyl < row_end) {
            float t0[RS_AHEAD], t1[RS_AHEAD];
#pragma unroll
            for (int i = 0; i < RS_AHEAD; i++) {
                const float *r = rowp(ys + 1 + i);
                t0[i] = r[sx];
                t1[i] = r[sx1];
            }
#pragma unroll
            for (int i = 0; i < RS_AHEAD; i++) {
                const float h_next = t0[i] * a0 + t1[i] * a1;
                while (pyl < row_end && sy_cur == ys + i) {     // wave-uniform: every destination row between the two source rows
                    const float u0 = h_cur * (1.0f - fcur), u1 = h_next * fcur;
                    const float v = u0 + u1;
                    dst[(size_t)pyl * dw + px] = v;
                    if (MM) {
                        const float z = v != v ? 0.f : v;
                        mn = fminf(mn, z);
                        mx = fmaxf(mx, z);
                    }
                    pyl++;
                    if (pyl < row_end) sy_cur = vpos(pyl, fcur);
                }
                h_cur = h_next;
            }
            ys += RS_AHEAD;
            if (pyl < row_end && sy_cur > ys) {             // a gap (downsampling, or a stripe that starts further on): jump
                ys = sy_cur;
                const float *r = rowp(ys);
                h_cur = r[sx] * a0 + r[sx1] * a1;
            }
        }
    }
    if (MM) mm_commit_wg(mm + 2 * plane, mn, mx);
}

static dim3 grid2d(int H, int W) { return dim3((W + WG_X - 1) / WG_X, (H + WG_Y - 1) / WG_Y); }

// rows form: the plane holds Hin rows; rows [y0, y1) are produced.  edges bit 0 / bit 1: row 0 / row Hin - 1 is the
// image's first / last row (the border rule applies there); otherwise it is a halo row of a stripe and must be out of
// the stencil's reach
static int rows_check(rsseg_ctx *ctx, const char *what, const void *in, const void *out, int Hin, int W, int y0, int y1, int R, int edges)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!in || !out || Hin < 1 || W < 1 || y0 < 0 || y1 > Hin || y0 > y1) return rs_fail(ctx, RSSEG_ERR_INVALID, "%s: bad arguments", what);
    if (in == out) return rs_fail(ctx, RSSEG_ERR_INVALID, "%s: in-place not supported", what);
    if (y0 < y1 && ((!(edges & 1) && y0 < R) || (!(edges & 2) && y1 + R > Hin)))
        return rs_fail(ctx, RSSEG_ERR_INVALID, "%s: rows [%d,%d) of a %d-row stripe need %d halo rows on a side that is not an image edge", what, y0, y1, Hin, R);
    return RSSEG_OK;
}

extern "C" int rsseg_box_mean_rows_f32(rsseg_ctx *ctx, const float *const *d_x, int nplanes, int Hin, int W, int y0, int y1, int edges, int k,
                                       int border, int square, float *const *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || !d_out || nplanes < 1 || nplanes > WIN_MAXP || (border != 0 && border != 1)) return rs_fail(ctx, RSSEG_ERR_INVALID, "box_mean: bad arguments");
    if (k != 3 && k != 5 && k != 7 && k != 9) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "box_mean: kernel size %d not in {3,5,7,9}", k);
    win_planes pl;
    memset(&pl, 0, sizeof(pl));
    for (int p = 0; p < nplanes; p++) {
        RSCHK(rows_check(ctx, "box_mean", d_x[p], d_out[p], Hin, W, y0, y1, k / 2, edges));
        pl.in[p] = d_x[p];
        pl.out[p] = d_out[p];
    }
    if (y0 == y1) return RSSEG_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, nplanes > 1 ? "ctxmean" : "box");  // "ctxmean": several planes per launch (add_spatial_context)
        const tile_map tm = make_tile_map(y1 - y0, W);
        const dim3 g(8 * tm.chunk, 1, nplanes);
#define BOX_GO(KV)                                                                                                          \
    case KV:                                                                                                                \
        if (square) hipLaunchKernelGGL((k6_box<KV, true>), g, dim3(256), 0, ctx->stream, pl, Hin, W, y0, y1 - y0, border, tm); \
        else hipLaunchKernelGGL((k6_box<KV, false>), g, dim3(256), 0, ctx->stream, pl, Hin, W, y0, y1 - y0, border, tm);      \
        break;
        switch (k) { BOX_GO(3) BOX_GO(5) BOX_GO(7) BOX_GO(9) }
#undef BOX_GO
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_box_mean_f32(rsseg_ctx *ctx, const float *d_x, int H, int W, int k, int border, int square, float *d_out)
{
    return rsseg_box_mean_rows_f32(ctx, &d_x, 1, H, W, 0, H, 3, k, border, square, &d_out);
}

extern "C" int rsseg_local_std_rows_f32(rsseg_ctx *ctx, const float *d_x, int Hin, int W, int y0, int y1, int edges, int k, int variance,
                                        float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (k != 3 && k != 5 && k != 7) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "local_std: kernel size %d not in {3,5,7}", k);
    RSCHK(rows_check(ctx, "local_std", d_x, d_out, Hin, W, y0, y1, k / 2, edges));
    if (y0 == y1) return RSSEG_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        prof_scope ps(ctx, "box");
        const tile_map tm = make_tile_map(y1 - y0, W);
        const dim3 g(8 * tm.chunk);
#define STD_GO(KV)                                                                                                    \
    case KV:                                                                                                          \
        if (variance) hipLaunchKernelGGL((k6_std<KV, true>), g, dim3(256), 0, ctx->stream, d_x, Hin, W, y0, y1 - y0, d_out, tm); \
        else hipLaunchKernelGGL((k6_std<KV, false>), g, dim3(256), 0, ctx->stream, d_x, Hin, W, y0, y1 - y0, d_out, tm);         \
        break;
        switch (k) { STD_GO(3) STD_GO(5) STD_GO(7) }
#undef STD_GO
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_local_std_f32(rsseg_ctx *ctx, const float *d_x, int H, int W, int k, float *d_out)
{
    return rsseg_local_std_rows_f32(ctx, d_x, H, W, 0, H, 3, k, 0, d_out);
}

extern "C" int rsseg_local_var_f32(rsseg_ctx *ctx, const float *d_x, int H, int W, int k, float *d_out)
{
    return rsseg_local_std_rows_f32(ctx, d_x, H, W, 0, H, 3, k, 1, d_out);
}

template <int MODE> static int morph_launch(rsseg_ctx *ctx, const uint8_t *in, int Hin, int W, int y0, int y1, int k, uint8_t *out)
{
    prof_scope ps(ctx, "morph");
    const tile_map tm = make_tile_map(y1 - y0, W);
    const dim3 g(8 * tm.chunk);
    switch (k) {
    case 3: hipLaunchKernelGGL((k7_morph<3, MODE>), g, dim3(256), 0, ctx->stream, in, Hin, W, y0, y1 - y0, out, tm); break;
    case 5: hipLaunchKernelGGL((k7_morph<5, MODE>), g, dim3(256), 0, ctx->stream, in, Hin, W, y0, y1 - y0, out, tm); break;
    case 7: hipLaunchKernelGGL((k7_morph<7, MODE>), g, dim3(256), 0, ctx->stream, in, Hin, W, y0, y1 - y0, out, tm); break;
    }
    return RSSEG_OK;
}

extern "C" int rsseg_morph_rows_u8(rsseg_ctx *ctx, const uint8_t *d_q, int Hin, int W, int y0, int y1, int edges, int k, int op, uint8_t *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (k != 3 && k != 5 && k != 7) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "morph: kernel size %d not in {3,5,7}", k);
    const int R = k / 2;
    const bool two = op == RSSEG_MORPH_OPEN || op == RSSEG_MORPH_CLOSE;
    if (op < RSSEG_MORPH_ERODE || op > RSSEG_MORPH_GRADIENT) return rs_fail(ctx, RSSEG_ERR_INVALID, "morph: unknown operation %d", op);
    RSCHK(rows_check(ctx, "morph", d_q, d_out, Hin, W, y0, y1, two ? 2 * R : R, edges));
    if (y0 == y1) return RSSEG_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (op == RSSEG_MORPH_ERODE) RSCHK(morph_launch<0>(ctx, d_q, Hin, W, y0, y1, k, d_out));
    else if (op == RSSEG_MORPH_DILATE) RSCHK(morph_launch<1>(ctx, d_q, Hin, W, y0, y1, k, d_out));
    else if (op == RSSEG_MORPH_GRADIENT) RSCHK(morph_launch<2>(ctx, d_q, Hin, W, y0, y1, k, d_out));
    else {
        // first pass on rows [t0, t1) = [y0 - R, y1 + R) clipped to the plane (a clipped side is an image edge by
        // rows_check), second pass on that intermediate: its rows outside [y0 - R, y1 + R) are never tapped
        const int t0 = std::max(y0 - R, 0), t1 = std::min(y1 + R, Hin);
        RSCHK(ws_reserve(ctx, (size_t)(t1 - t0) * W + 64));
        uint8_t *tmp = (uint8_t *)ctx->d_ws;
        if (op == RSSEG_MORPH_OPEN) {
            RSCHK(morph_launch<0>(ctx, d_q, Hin, W, t0, t1, k, tmp));
            RSCHK(morph_launch<1>(ctx, tmp, t1 - t0, W, y0 - t0, y1 - t0, k, d_out));
        } else {
            RSCHK(morph_launch<1>(ctx, d_q, Hin, W, t0, t1, k, tmp));
            RSCHK(morph_launch<0>(ctx, tmp, t1 - t0, W, y0 - t0, y1 - t0, k, d_out));
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_morph_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int k, int op, uint8_t *d_out)
{
    return rsseg_morph_rows_u8(ctx, d_q, H, W, 0, H, 3, k, op, d_out);
}

extern "C" int rsseg_morph_gradient_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int k, uint8_t *d_out)
{
    return rsseg_morph_rows_u8(ctx, d_q, H, W, 0, H, 3, k, RSSEG_MORPH_GRADIENT, d_out);
}

// Sobel magnitude / Laplacian: extrema pass, all-reduce of the two scalars, normalising write pass
template <int KIND>
static int filter_rows(rsseg_ctx *ctx, const char *what, const uint8_t *d_q, int Hin, int W, int y0, int y1, int edges, float *d_out)
{
    RSCHK(rows_check(ctx, what, d_q, d_out, Hin, W, y0, y1, 1, edges));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    RSCHK(ws_reserve(ctx, 64 * RSSEG_MM_REPL));
    RSCHK(pin_reserve(ctx, 64 * RSSEG_MM_REPL));
    uint32_t *d_keys = (uint32_t *)ctx->d_ws;   // [RSSEG_MM_REPL] lines of {min key, max key, ...}
    const tile_map tm = make_tile_map(std::max(y1 - y0, 1), W);
    const dim3 g(8 * tm.chunk);
    double mm[2] = {-INFINITY, -INFINITY};  // {-(min), max}: a rank without rows contributes nothing to the MAX-reduce
    if (y1 > y0) {
        HIPCHK(ctx, hipMemsetAsync(d_keys, 0, 64 * RSSEG_MM_REPL, ctx->stream));
        HIPCHK(ctx, hipMemset2DAsync(d_keys, 64, 0xff, 4, RSSEG_MM_REPL, ctx->stream));
        {
            prof_scope ps(ctx, "filt_max");
            hipLaunchKernelGGL((k8_filter<KIND, 0>), g, dim3(256), 0, ctx->stream, d_q, Hin, W, y0, y1 - y0, (float *)nullptr, 0.f, 1.f, d_keys, tm);
        }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_keys, 64 * RSSEG_MM_REPL, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, rs_sync(ctx));
        uint32_t kmn = 0xffffffffu, kmx = 0;
        for (int r = 0; r < RSSEG_MM_REPL; r++) {
            kmn = std::min(kmn, ((const uint32_t *)ctx->h_pin)[16 * r]);
            kmx = std::max(kmx, ((const uint32_t *)ctx->h_pin)[16 * r + 1]);
        }
        auto unkey = [](uint32_t key) {
            uint32_t u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
            float f;
            memcpy(&f, &u, 4);
            return (double)f;
        };
        mm[0] = -unkey(kmn);
        mm[1] = unkey(kmx);
    }
    RSCHK(comm_allreduce_host(ctx, mm, 2, RSSEG_F64, RSSEG_MAX));
    if (y1 == y0) return RSSEG_OK;
    volatile float fmn = (float)(-mm[0]), fmx = (float)mm[1];
    float sub, den;
    if (KIND == 0) {  // sobel_mag / (sobel_mag.max() + 1e-10), float32 (indices.py:480)
        volatile float d = fmx + 1e-10f;
        sub = 0.f;
        den = d;
    } else {          // (l - min) / (max - min + 1e-10), float32 throughout (indices.py:474)
        volatile float range = fmx - fmn;
        volatile float d = range + 1e-10f;
        sub = fmn;
        den = d;
    }
    {
        prof_scope ps(ctx, "filt_write");
        hipLaunchKernelGGL((k8_filter<KIND, 1>), g, dim3(256), 0, ctx->stream, d_q, Hin, W, y0, y1 - y0, d_out, sub, den, (uint32_t *)nullptr, tm);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}

extern "C" int rsseg_sobel_mag_rows_u8(rsseg_ctx *ctx, const uint8_t *d_q, int Hin, int W, int y0, int y1, int edges, float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    return filter_rows<0>(ctx, "sobel_mag", d_q, Hin, W, y0, y1, edges, d_out);
}

extern "C" int rsseg_sobel_mag_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, float *d_out)
{
    return rsseg_sobel_mag_rows_u8(ctx, d_q, H, W, 0, H, 3, d_out);
}

extern "C" int rsseg_laplacian_norm_rows_u8(rsseg_ctx *ctx, const uint8_t *d_q, int Hin, int W, int y0, int y1, int edges, float *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    return filter_rows<1>(ctx, "laplacian", d_q, Hin, W, y0, y1, edges, d_out);
}

extern "C" int rsseg_laplacian_norm_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, float *d_out)
{
    return rsseg_laplacian_norm_rows_u8(ctx, d_q, H, W, 0, H, 3, d_out);
}

static int resize_rows(rsseg_ctx *ctx, const float *const *d_src, int nplanes, int sh_local, int sw, int src_row0, int sh, float *const *d_dst,
                       int dh_local, int dw, int dst_row0, int dh)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_src || !d_dst || nplanes < 1 || nplanes > WIN_MAXP || sh < 1 || sw < 1 || dh < 1 || dw < 1 || sh_local < 1 || dh_local < 1 || src_row0 < 0 ||
        dst_row0 < 0 || src_row0 + sh_local > sh || dst_row0 + dh_local > dh)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "resize: bad arguments");
    resize_planes pl;
    memset(&pl, 0, sizeof(pl));
    for (int p = 0; p < nplanes; p++) {
        if (!d_src[p] || !d_dst[p]) return rs_fail(ctx, RSSEG_ERR_INVALID, "resize: plane %d is null", p);
        pl.src[p] = d_src[p];
        pl.dst[p] = d_dst[p];
    }
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
    RSCHK(mm_begin(ctx, nplanes));
    // one launch per plane (five source maps walked at once thrash the L2: measured 5.8 against 3.1 ms at 16384^2), but
    // ONE extrema read-back and synchronisation for all of them
    for (int p = 0; p < nplanes; p++) {
        prof_scope ps(ctx, "resize");
        // about 8192 workgroups: every 256-column strip gets ny of them, each with R consecutive rows
        const int gx = (dw + RS_W - 1) / RS_W;
        const int R = (int)std::max<int64_t>(1, ceil_div64(dh_local, std::max(1, 8192 / gx)));
        const int ny = (int)std::max<int64_t>(1, ceil_div64(dh_local, R));
        const dim3 pg((unsigned)((int64_t)gx * ny));
        if (ctx->mm_collect)
            hipLaunchKernelGGL(k5_resize<true>, pg, dim3(256), 0, ctx->stream, pl, sh, sw, dh, dw, scale_x, scale_y, src_row0, sh_local, dst_row0, dh_local, gx, R, p,
                               ctx->d_mm);
        else
            hipLaunchKernelGGL(k5_resize<false>, pg, dim3(256), 0, ctx->stream, pl, sh, sw, dh, dw, scale_x, scale_y, src_row0, sh_local, dst_row0, dh_local, gx, R, p,
                               (uint32_t *)nullptr);
    }
    HIPCHK(ctx, hipGetLastError());
    RSCHK(mm_end(ctx, nplanes));
    return ctx->mm_collect ? RSSEG_OK : stream_sync(ctx);   // the extrema read-back has already waited for the stream
}

extern "C" int rsseg_resize_bilinear_f32(rsseg_ctx *ctx, const float *d_src, int sh, int sw, float *d_dst, int dh, int dw)
{
    return resize_rows(ctx, &d_src, 1, sh, sw, 0, sh, &d_dst, dh, dw, 0, dh);
}

extern "C" int rsseg_resize_bilinear_rows_f32(rsseg_ctx *ctx, const float *d_src, int sh_local, int sw, int src_row0, int sh,
                                              float *d_dst, int dh_local, int dw, int dst_row0, int dh)
{
    return resize_rows(ctx, &d_src, 1, sh_local, sw, src_row0, sh, &d_dst, dh_local, dw, dst_row0, dh);
}

extern "C" int rsseg_resize_bilinear_rows_multi_f32(rsseg_ctx *ctx, const float *const *d_src, int nplanes, int sh_local, int sw, int src_row0, int sh,
                                                    float *const *d_dst, int dh_local, int dw, int dst_row0, int dh)
{
    return resize_rows(ctx, d_src, nplanes, sh_local, sw, src_row0, sh, d_dst, dh_local, dw, dst_row0, dh);
}
