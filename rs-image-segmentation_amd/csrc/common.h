// Internal definitions shared by the gfx950 kernels of librsseg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/rsseg.h"

#define WAVE 64

struct prof_entry {
    double ms = 0.0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct forest_dev {
    void *d_nodes = nullptr;    // packed 8-byte nodes (k11_forest.hip)
    void *d_leafval = nullptr;  // double[n_nodes_total][n_classes]
    void *d_treeoff = nullptr;  // int64 classes[n_classes], then rf_tree[n_trees]
    void *d_groups = nullptr;   // rf_group[n_groups]: the LDS plan (k11_forest.hip); none when a tree exceeds the LDS area
    int n_groups = 0;
    int n_trees = 0, n_classes = 0, n_features = 0, max_depth = 0;
    int64_t n_nodes = 0;
    int64_t classes[64];
};

struct rsseg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool async_mode = false;  // entry points return after enqueueing (no trailing stream sync)
    char err[512] = {0};
    // communication
    int rank = 0, world = 1;
    std::vector<std::pair<const void *, int64_t>> byte_valued;   // (plane, pixels) the last select saw hold only the integers 0..255 (a hint: users verify)
    bool comm_on = false;   // reductions go through the hook: world > 1, or a hook installed on a single rank (identity reduction; how a one-GPU box exercises the RCCL path)
    rsseg_allreduce_fn allreduce = nullptr;
    void *comm_user = nullptr;
    char *d_comm = nullptr;
    size_t comm_bytes = 0;
    char *h_comm = nullptr;   // pinned staging of the communication buffer (comm_bytes)
    void *rccl_comm = nullptr;   // ncclComm_t when the library drives RCCL itself (rsseg_ctx_set_comm_rccl)
    bool own_comm_buf = false;   // d_comm was allocated by the library
    // workspace (device) and pinned host staging
    char *d_ws = nullptr;
    size_t ws_bytes = 0;
    char *h_pin = nullptr;
    size_t pin_bytes = 0;
    // profiling
    bool prof_on = false;
    std::map<std::string, prof_entry> prof;
    std::vector<hipEvent_t> event_pool;
    forest_dev forest;
    // optional per-plane extrema of the planes the last producing call wrote (rsseg_ctx_collect_minmax)
    bool mm_collect = false;
    uint32_t *d_mm = nullptr;  // [RSSEG_MM_REPL][RSSEG_MM_PLANES][2] ordered keys {min, max}
    int mm_count = 0;
    double mm_min[16], mm_max[16];
    // host synchronisations (hipStreamSynchronize / blocking copies) the library has made on this context (rsseg_ctx_host_syncs)
    long long host_syncs = 0;
};
// every wait of the host for the context's stream goes through here, so that it is counted
static inline hipError_t rs_sync(rsseg_ctx *ctx)
{
    ctx->host_syncs++;
    return hipStreamSynchronize(ctx->stream);
}
// moves the last `count` recorded launches of family `from` to family `to` (speculative launches that turned out to be no-ops)
void prof_retag(rsseg_ctx *ctx, const char *from, int count, const char *to);
#define RSSEG_MM_PLANES 16
#define RSSEG_MM_REPL 64   // replicas of the slot table, one 64-byte line each: a workgroup commits to replica blockIdx.x % 64
int mm_begin(rsseg_ctx *ctx, int nplanes);   // reset the device slots before a producing launch (no-op when off)
int mm_end(rsseg_ctx *ctx, int nplanes);     // read them back into ctx->mm_min / mm_max (after the launch)

int rs_fail(rsseg_ctx *ctx, int code, const char *fmt, ...);
#define HIPCHK(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return rs_fail(ctx, RSSEG_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,    \
                           hipGetErrorString(e_));                                             \
    } while (0)
#define RSCHK(expr)                  \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != RSSEG_OK) return rc_; \
    } while (0)

// workspace: returns a device pointer valid until the next ws_reserve with a larger size
int ws_reserve(rsseg_ctx *ctx, size_t bytes);
int pin_reserve(rsseg_ctx *ctx, size_t bytes);
// all-reduce of a small host array through the device comm buffer (no-op when world == 1)
int comm_allreduce_host(rsseg_ctx *ctx, void *host, int64_t count, int dtype, int op);
int stream_sync(rsseg_ctx *ctx);

// scoped kernel timer (HIP events on ctx->stream) — active only when profiling is on
struct prof_scope {
    rsseg_ctx *ctx;
    prof_entry *e = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    prof_scope(rsseg_ctx *c, const char *name);
    ~prof_scope();
};

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
// float32 (hi - lo + 1e-10) of robust_normalize (indices.py:44)
static inline float norm_den(float lo, float hi)
{
    volatile float d = hi - lo;
    volatile float e = d + 1e-10f;
    return e;
}

#ifdef __HIPCC__
// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// robust_normalize of one value: np.clip keeps NaN; (clipped - lo) / (hi - lo + 1e-10), float32 (indices.py:41-44)
__device__ __forceinline__ float norm1(float x, float lo, float hi, float den)
{
    float c = x < lo ? lo : x;
    c = c > hi ? hi : c;
    return (c - lo) / den;
}

template <typename V>
__device__ __forceinline__ V wave_sum(V v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- running extrema of a produced plane (NaN counts as 0, as the KMeans scaler sees it) -----------------------
// monotone map float -> uint32 (as in k1_select.hip); atomicMin / atomicMax on the keys
__device__ __forceinline__ uint32_t mm_key(float x)
{
    if (x != x) x = 0.f;
    uint32_t u = __float_as_uint(x);
    if (u == 0x80000000u) u = 0;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// per-wave reduction, then at most one atomic per wave and only when it improves the current global value
__device__ __forceinline__ void mm_commit(uint32_t *slot, float mn, float mx)
{
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane_id() == 0) {
        const uint32_t kmn = mm_key(mn), kmx = mm_key(mx);
        if (kmn < __builtin_nontemporal_load(&slot[0])) atomicMin(&slot[0], kmn);
        if (kmx > __builtin_nontemporal_load(&slot[1])) atomicMax(&slot[1], kmx);
    }
}

// the same once per WORKGROUP (every thread of the block must call it): a per-wave test makes tens of thousands of waves
// read one address at the end of a kernel, ~0.3 ms whatever the raster size — a first-order term for the small stripes
// of a sharded raster
__device__ __forceinline__ void mm_commit_wg(uint32_t *slot, float mn, float mx)
{
    __shared__ float s_mn[16], s_mx[16];
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane_id() == 0) { s_mn[w] = mn; s_mx[w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; i++) { mn = fminf(mn, s_mn[i]); mx = fmaxf(mx, s_mx[i]); }
        const uint32_t kmn = mm_key(mn), kmx = mm_key(mx);
        uint32_t *rs = slot + (size_t)(blockIdx.x % RSSEG_MM_REPL) * (2 * RSSEG_MM_PLANES);   // this workgroup's replica
        if (kmn < __builtin_nontemporal_load(&rs[0])) atomicMin(&rs[0], kmn);
        if (kmx > __builtin_nontemporal_load(&rs[1])) atomicMax(&rs[1], kmx);
    }
    __syncthreads();
}

// streaming (read-once) 16-byte loads: non-temporal, so that 7-30 GB of planes per pass do not push the small tables and the
// other streams' lines out of the caches
typedef float rs_f4v __attribute__((ext_vector_type(4)));
typedef unsigned int rs_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream_f4(const void *plane, int64_t i)
{
    const rs_f4v v = __builtin_nontemporal_load(reinterpret_cast<const rs_f4v *>(plane) + i);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint32_t ld_stream_u32(const void *plane, int64_t i)
{
    return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(plane) + i);
}

// ---- the seven spectral indices of one pixel (indices.py:62-69, 86-93, 109-112, 128-135, 150-156, 171-177, 194-201) ------------
// shared by k2_indices and the fused index + PCA-projection kernel of k3_pca.hip; one IEEE float32 operation per NumPy operation
__device__ __forceinline__ float clip11(float v)
{
    v = v < -1.0f ? -1.0f : v;
    return v > 1.0f ? 1.0f : v;
}
__device__ __forceinline__ float ratio_index(float num, float den)
{
    // zeros_like; out[mask] = num/den with mask = den > 0.001 (False for NaN); clip to [-1, 1]
    float v = den > 0.001f ? num / den : 0.0f;
    return clip11(v);
}
struct evi_coef_t {
    float L, C1, C2, G;   // calculate_evi's coefficients (defaults 1, 6, 7.5, 2.5), as float32 like NumPy's weak scalars
};
// nb: normalised blue, green, red, nir, swir1;  o: ndvi, evi, msavi, ndwi, mndwi, ndbi, bsi
__device__ __forceinline__ void indices_pixel(const evi_coef_t &e, const float nb[5], float o[7])
{
    const float blue = nb[0], green = nb[1], red = nb[2], nir = nb[3], swir = nb[4];
    const float nmr = nir - red;
    o[0] = ratio_index(nmr, nir + red);
    {   // indices.py:86-93  nir + C1*red - C2*blue + L ;  G*(nir-red)/den
        float den = nir + e.C1 * red;
        den = den - e.C2 * blue;
        den = den + e.L;
        o[1] = ratio_index(e.G * nmr, den);
    }
    {   // indices.py:109-112  (a - sqrt(a**2 - 8*(nir-red))) / 2
        const float a = 2.0f * nir + 1.0f;
        float r = a * a - 8.0f * nmr;
        float m = (a - sqrtf(r)) / 2.0f;
        o[2] = clip11(m);  // NaN propagates like np.clip
    }
    o[3] = ratio_index(green - nir, green + nir);
    o[4] = ratio_index(green - swir, green + swir);
    o[5] = ratio_index(swir - nir, swir + nir);
    {
        const float a = swir + red, b = nir + blue;
        o[6] = ratio_index(a - b, a + b);
    }
}

// round-to-nearest-even fixed point, quantum 2^-40, exact for |x| < 2^11:
// bits(fma(x, 2^40, 1.5*2^52)) - bits(1.5*2^52) == llrint(x * 2^40)
#define FX_MAGIC 6755399441055744.0 /* 1.5 * 2^52 */
__device__ __forceinline__ long long to_fixed40(double x)
{
    double d = fma(x, 1099511627776.0, FX_MAGIC);
    return __double_as_longlong(d) - __double_as_longlong(FX_MAGIC);
}
#endif
