// K1 — exact order statistics of a float32 plane by 3-pass radix select (11 + 11 + 10 key bits).
//
// Replaces the introselect partition inside np.percentile (reference modules/features/indices.py:38-39)
// and np.nanmedian / np.nanpercentile of sklearn's RobustScaler (indices.py:230-231).  All requested
// ranks are resolved together: each pass reads the plane once (4 B/px, coalesced 16 B per lane),
// builds LDS-private histograms per workgroup (one 2048-bin table per distinct key prefix still
// alive) and flushes the non-zero bins to HBM with 64-bit atomics.  HBM-bound: 12 B/px for any
// number of ranks <= RSSEG_MAX_RANKS.
#include <mutex>

#include "common.h"

#define SEL_BINS 2048
#define SEL_THREADS 256

__device__ __forceinline__ uint32_t f32_key(float x, bool &is_nan)
{
    uint32_t u = __float_as_uint(x);
    is_nan = (u & 0x7fffffffu) > 0x7f800000u;
    if (u == 0x80000000u) u = 0;  // -0.0 sorts with +0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// pass 3 (tried FIRST): small-integer planes.  The reference's preprocessed tiles are uint8 digital numbers stored as float32
//         (preprocessing.py:117-118, 144): when every value of a plane is an integer in [0, 2048) or NaN, ONE pass with
//         bin = (int)value resolves every rank exactly (value = bin) — one read of the plane and one all-reduce instead of
//         three.  The first value that does not qualify raises a flag in global memory and every workgroup stops at its next
//         batch, so planes of general floats pay ~1 % of a pass before the three radix passes below take over.
// pass 0: bin = key >> 21 (one table).
// pass 1: prefix = key >> 21 (11 bits), bin = (key >> 10) & 2047, one 2048-bin table per live prefix.
// pass 2: prefix = key >> 10 (22 bits), bin = key & 1023.
// Which table an element feeds is found with O(1) LDS byte lookups, not a scan of the live prefixes:
// tab1[key >> 21] = slot of the 11-bit prefix (255 = not wanted); pass 2 adds tab2[slot1][(key >> 10) & 2047].
#define SEL_NONE 255u
// Counter / lookup position of bin b inside a 2048-entry table.  Integer-valued planes put all their keys on
// multiples of 64 (the low mantissa bits are zero), i.e. on ONE LDS bank; folding bits 6..10 into bits 0..4 spreads
// them over the banks.  An involution: swz(swz(b)) == b.
__host__ __device__ __forceinline__ uint32_t swz(uint32_t b) { return b ^ (b >> 6); }

// THREADS: 256, or 1024 so that the one set of tables a CU has room for (up to 82 KB) is shared by 16 waves.
// BATCH: the LDS lookups of the 4*UNR values of a lane are issued back to back (one wait) instead of value by value.
// (Merging the lanes of a wave that hit the same counter before the atomic was measured and does not pay.)
// Pass 0 has LDS to spare and the most crowded counters (sign, exponent and two mantissa bits: a coherent image
// region is ONE bin), so its table is replicated P0_COPIES times, copy = lane & 7 at an odd word stride.
#define P0_COPIES 8
#define P0_STRIDE (SEL_BINS + 1)
template <int PASS, int THREADS, int UNR>
__global__ __launch_bounds__(THREADS) void k1_hist(const float *__restrict__ x, int64_t n,
                                                       const uint32_t *__restrict__ prefixes, int nprefix,
                                                       unsigned long long *__restrict__ hist,
                                                       unsigned long long *__restrict__ nan_count)
{
    extern __shared__ uint32_t lh[];  // [ntab][SEL_BINS] counters, then the byte lookup tables
    const int ntab = (PASS == 0 || PASS == 3) ? 1 : nprefix;
    const int nb = (PASS == 0 || PASS == 3) ? P0_COPIES * P0_STRIDE : ntab * SEL_BINS;
    const uint32_t my_copy = (threadIdx.x & (P0_COPIES - 1)) * P0_STRIDE;
    uint8_t *tab1 = reinterpret_cast<uint8_t *>(lh + nb);  // [2048]
    uint8_t *tab2 = tab1 + SEL_BINS;                       // [n1][2048], pass 2 only
    __shared__ uint32_t p1list[RSSEG_MAX_RANKS];
    __shared__ int n1s;
    for (int i = threadIdx.x; i < nb; i += THREADS) lh[i] = 0;
    if (PASS != 0 && PASS != 3) {
        for (int i = threadIdx.x; i < SEL_BINS; i += THREADS) tab1[i] = SEL_NONE;
        if (threadIdx.x == 0) {
            int n1 = 0;  // distinct 11-bit prefixes
            for (int j = 0; j < nprefix; j++) {
                const uint32_t p1 = PASS == 1 ? prefixes[j] : (prefixes[j] >> 11);
                bool seen = false;
                for (int t = 0; t < n1; t++) seen = seen || p1list[t] == p1;
                if (!seen) p1list[n1++] = p1;
            }
            n1s = n1;
        }
        __syncthreads();
        if (PASS == 2)
            for (int i = threadIdx.x; i < n1s * SEL_BINS; i += THREADS) tab2[i] = SEL_NONE;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int t = 0; t < n1s; t++) tab1[p1list[t]] = (uint8_t)t;
            if (PASS == 1) {
                for (int j = 0; j < nprefix; j++) tab1[prefixes[j]] = (uint8_t)j;  // slot = table index
            } else {
                for (int j = 0; j < nprefix; j++) {
                    const uint32_t p1 = prefixes[j] >> 11, mid = prefixes[j] & 2047u;
                    tab2[(size_t)tab1[p1] * SEL_BINS + swz(mid)] = (uint8_t)j;
                }
            }
        }
    }
    __syncthreads();
    uint32_t my_nan = 0, my_bad = 0;
    const int64_t n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    auto handle = [&](float v) {
        bool isn;
        uint32_t k = f32_key(v, isn);
        if (isn) my_nan++;
        if (PASS == 3) {
            const bool ok = !isn && v >= 0.f && v < (float)SEL_BINS && truncf(v) == v;   // -0.0 counts as 0, like the radix key
            if (!isn && !ok) my_bad++;
            if (ok) atomicAdd(&lh[my_copy + swz((uint32_t)(int)v)], 1u);
            return;
        }
        bool want = !isn;
        uint32_t idx = 0;
        if (PASS == 0) {
            idx = my_copy + swz(k >> 21);
        } else {
            const uint32_t s1 = tab1[k >> 21];
            want = want && s1 != SEL_NONE;
            if (PASS == 1) {
                idx = (want ? s1 : 0u) * SEL_BINS + swz((k >> 10) & 2047u);
            } else {
                const uint32_t s2 = want ? tab2[(size_t)s1 * SEL_BINS + swz((k >> 10) & 2047u)] : SEL_NONE;
                want = want && s2 != SEL_NONE;
                idx = (want ? s2 : 0u) * SEL_BINS + swz(k & 1023u);
            }
        }
        if (want) atomicAdd(&lh[idx], 1u);
    };
    {
        const int64_t stride = (int64_t)gridDim.x * THREADS;
        int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
        for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {  // UNR 16-byte loads in flight per lane
            // PASS 3 gives up as soon as any workgroup met a value that is not a small integer.  The flag is requested
            // TOGETHER with the batch's data (r04; before, the data loads waited for the flag: two dependent memory round
            // trips per batch) and looked at after the batch — a stale flag costs one more batch, never a result
            unsigned long long stop = 0;
            if (PASS == 3) {
                if (my_bad) __builtin_nontemporal_store(1ull, nan_count + 2);
                stop = __builtin_nontemporal_load(nan_count + 2);
            }
            float4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) v[u] = ld_stream_f4(x, i + u * stride);
#pragma unroll
            for (int u = 0; u < UNR; u++) { handle(v[u].x); handle(v[u].y); handle(v[u].z); handle(v[u].w); }
            if (PASS == 3 && stop) { i = n4; break; }   // nothing of this plane's histogram is used any more
        }
        for (; i < n4; i += stride) {
            float4 v = x4[i];
            handle(v.x); handle(v.y); handle(v.z); handle(v.w);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) handle(x[(n4 << 2) + threadIdx.x]);
    __syncthreads();
    if (PASS == 0 || PASS == 3) {
        for (int i = threadIdx.x; i < SEL_BINS; i += THREADS) {
            uint32_t c = 0;
#pragma unroll
            for (int cp = 0; cp < P0_COPIES; cp++) c += lh[cp * P0_STRIDE + i];
            if (c) atomicAdd(&hist[swz(i)], (unsigned long long)c);
        }
    } else {
        for (int i = threadIdx.x; i < nb; i += THREADS) {
            uint32_t c = lh[i];
            if (c) atomicAdd(&hist[(i & ~(SEL_BINS - 1)) + swz(i & (SEL_BINS - 1))], (unsigned long long)c);
        }
    }
    if (PASS == 0 || PASS == 3) {
        uint32_t t = wave_sum(my_nan);
        if (lane_id() == 0 && t) atomicAdd(nan_count, (unsigned long long)t);
    }
    if (PASS == 3) {
        uint32_t t = wave_sum(my_bad);
        if (lane_id() == 0 && t) atomicAdd(nan_count + 1, (unsigned long long)t);
    }
}

// 8-bit planes (the TM tiles the reference reads are uint8 digital numbers, preprocessing.py:117-118): bin = value, 256 bins,
// 16 pixels per 16-byte load.  Every lane adds into one of U8_COPIES private copies of the table (odd word stride: a wave's
// lanes that carry the same value — coherent image regions — land on different banks).
#define U8_COPIES 32
#define U8_STRIDE 257
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k1_hist_u8(const uint8_t *__restrict__ x, int64_t n, unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t lh[U8_COPIES * U8_STRIDE];
    for (int i = threadIdx.x; i < U8_COPIES * U8_STRIDE; i += THREADS) lh[i] = 0;
    __syncthreads();
    uint32_t *mine = lh + (threadIdx.x & (U8_COPIES - 1)) * U8_STRIDE;
    const int64_t n16 = n >> 4;
    const uint4 *x16 = reinterpret_cast<const uint4 *>(x);
    auto word = [&](uint32_t w) {
        atomicAdd(&mine[w & 255u], 1u);
        atomicAdd(&mine[(w >> 8) & 255u], 1u);
        atomicAdd(&mine[(w >> 16) & 255u], 1u);
        atomicAdd(&mine[w >> 24], 1u);
    };
    const int64_t stride = (int64_t)gridDim.x * THREADS;
    int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    for (; i + stride < n16; i += 2 * stride) {   // two 16-byte loads in flight per lane
        const uint4 a = x16[i], b = x16[i + stride];
        word(a.x); word(a.y); word(a.z); word(a.w);
        word(b.x); word(b.y); word(b.z); word(b.w);
    }
    for (; i < n16; i += stride) {
        const uint4 a = x16[i];
        word(a.x); word(a.y); word(a.z); word(a.w);
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < (n & 15)) atomicAdd(&mine[x[(n16 << 4) + threadIdx.x]], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < 256; b += THREADS) {
        uint32_t c = 0;
#pragma unroll 8
        for (int cp = 0; cp < U8_COPIES; cp++) c += lh[cp * U8_STRIDE + b];
        if (c) atomicAdd(&hist[b], (unsigned long long)c);
    }
}

static float key_to_f32(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// All planes advance pass by pass together: ONE device-to-host copy, ONE stream synchronisation and ONE all-reduce
// per pass for the whole group (3 per group instead of 3 per plane).
#define SEL_MAX_PLANES 8
static int order_stats_core(rsseg_ctx *ctx, const float *const *d_planes, int P, int64_t n_local, const int64_t *ranks, int nranks,
                            float *out_values, int64_t *n_nan_out)
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->byte_valued.clear();
    // per plane: 8 words (NaN counter first) then up to RSSEG_MAX_RANKS tables of 2048 bins; only the tables a pass
    // uses travel to the host and through the all-reduce
    const size_t hist_elems = 8 + (size_t)RSSEG_MAX_RANKS * SEL_BINS;
    const size_t hist_bytes = hist_elems * sizeof(unsigned long long);
    RSCHK(ws_reserve(ctx, (size_t)P * (hist_bytes + 256)));
    RSCHK(pin_reserve(ctx, (size_t)P * hist_bytes));
    unsigned long long *d_hist_all = (unsigned long long *)ctx->d_ws;
    uint32_t *d_pre_all = (uint32_t *)(ctx->d_ws + (size_t)P * hist_bytes);  // [P][64]
    long long *h_hist_all = (long long *)ctx->h_pin;
    if (ctx->comm_on && (size_t)P * hist_bytes > ctx->comm_bytes)
        return rs_fail(ctx, RSSEG_ERR_COMM, "order_stats: %d planes may need a %zu-byte communication buffer", P, (size_t)P * hist_bytes);

    static bool attr_done[64] = {false};  // hipFuncSetAttribute is per device
    static std::mutex attr_mu;            // contexts of several threads (one per rank in the threaded tests) may arrive together
    const int SEL_BATCH = 8;  // live prefixes per launch: 8 * (8 KB counters + 2 KB lookup) + 2 KB of LDS
    std::unique_lock<std::mutex> attr_lock(attr_mu);
    if (!attr_done[ctx->device & 63]) {
        const int l1 = SEL_BATCH * SEL_BINS * 4 + SEL_BINS, l2 = SEL_BATCH * SEL_BINS * 5 + SEL_BINS, l0 = P0_COPIES * P0_STRIDE * 4;
#define SEL_ATTR(TH)                                                                                                       \
    HIPCHK(ctx, hipFuncSetAttribute((const void *)k1_hist<0, TH, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, l0));    \
    HIPCHK(ctx, hipFuncSetAttribute((const void *)k1_hist<1, TH, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, l1));    \
    HIPCHK(ctx, hipFuncSetAttribute((const void *)k1_hist<2, TH, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, l2));    \
    HIPCHK(ctx, hipFuncSetAttribute((const void *)k1_hist<3, TH, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, l0));
        SEL_ATTR(1024)
        attr_done[ctx->device & 63] = true;
    }
    attr_lock.unlock();
    const int threads = 1024;
    // 512 workgroups = the two per CU that are resident anyway (66 KB of LDS each): every workgroup clears and flushes its
    // tables once.  profiles/r04_k1_sweep.json: 0.174 ms per 16384^2 plane (6.2 TB/s) against 0.200 ms with 2048 workgroups
    int grid_cap = 512;
    if (const char *e = getenv("RSSEG_K1_GRID")) grid_cap = std::max(1, atoi(e));   // experiments (profiles/r04_k1_sweep.py)
    const int grid = (int)std::min<int64_t>(grid_cap, std::max<int64_t>(1, ceil_div64(n_local >> 2, threads)));
    auto launch = [&](const float *d_x, int pass, size_t lds, const uint32_t *pre, int npre, unsigned long long *hb, unsigned long long *d_nan) {
#define SEL_GO(PS, TH) hipLaunchKernelGGL((k1_hist<PS, TH, 4>), dim3(grid), dim3(TH), lds, ctx->stream, d_x, n_local, pre, npre, hb, d_nan)
#define SEL_PASS(TH)                         \
    do {                                     \
        if (pass == 0) SEL_GO(0, TH);        \
        else if (pass == 1) SEL_GO(1, TH);   \
        else if (pass == 2) SEL_GO(2, TH);   \
        else SEL_GO(3, TH);                  \
    } while (0)
        SEL_PASS(1024);
    };
    struct plane_state {
        int64_t rem[RSSEG_MAX_RANKS];
        uint32_t prefix[RSSEG_MAX_RANKS];
        bool is_nan_rank[RSSEG_MAX_RANKS];
        int64_t n_global, n_nan;
        uint32_t dp[RSSEG_MAX_RANKS];
        int ndp, slot[RSSEG_MAX_RANKS];
    };
    std::vector<plane_state> st((size_t)P);
    std::vector<uint32_t> h_pre((size_t)P * 64);

    {   // ---- small-integer fast path (pass 3): one read of every plane, one all-reduce ----
        const size_t used = 8 + SEL_BINS;
        HIPCHK(ctx, hipMemsetAsync(d_hist_all, 0, (size_t)P * hist_bytes, ctx->stream));
        for (int p = 0; p < P; p++) {
            prof_scope ps(ctx, "select");
            unsigned long long *d_nan = d_hist_all + (size_t)p * hist_elems;
            launch(d_planes[p], 3, (size_t)P0_COPIES * P0_STRIDE * sizeof(uint32_t), d_pre_all, 1, d_nan + 8, d_nan);
        }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpy2DAsync(h_hist_all, used * 8, d_hist_all, hist_bytes, used * 8, (size_t)P, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, rs_sync(ctx));
        RSCHK(comm_allreduce_host(ctx, h_hist_all, (int64_t)((size_t)P * used), RSSEG_I64, RSSEG_SUM));
        bool small_ints = true;
        for (int p = 0; p < P; p++) small_ints = small_ints && h_hist_all[(size_t)p * used + 1] == 0 && h_hist_all[(size_t)p * used + 2] == 0;
        if (small_ints) {
            for (int p = 0; p < P; p++) {
                const long long *h_nanp = h_hist_all + (size_t)p * used, *h = h_nanp + 8;
                const int64_t *rk = ranks + (size_t)p * nranks;
                {   // a plane of 8-bit digital numbers (no NaN, nothing above 255): later passes over it may use 256-entry tables
                    bool bytes = h_nanp[0] == 0;
                    for (int b = 256; b < SEL_BINS && bytes; b++) bytes = h[b] == 0;
                    if (bytes) ctx->byte_valued.emplace_back((const void *)d_planes[p], n_local);
                }
                int64_t n_global = h_nanp[0];
                for (int b = 0; b < SEL_BINS; b++) n_global += h[b];
                for (int r = 0; r < nranks; r++) {
                    if (rk[r] < 0 || rk[r] >= n_global)
                        return rs_fail(ctx, RSSEG_ERR_INVALID, "order_stats: rank %lld outside [0,%lld)", (long long)rk[r], (long long)n_global);
                    float v = __builtin_nanf("");
                    if (rk[r] < n_global - h_nanp[0]) {   // NaNs sort last
                        int64_t acc = 0;
                        int b = 0;
                        for (; b < SEL_BINS; b++) {
                            if (rk[r] < acc + h[b]) break;
                            acc += h[b];
                        }
                        v = (float)b;
                    }
                    out_values[(size_t)p * nranks + r] = v;
                }
                if (n_nan_out) n_nan_out[p] = h_nanp[0];
            }
            return RSSEG_OK;
        }
    }
    for (int pass = 0; pass < 3; pass++) {
        int live = 0;
        for (int p = 0; p < P; p++) {
            plane_state &S = st[p];
            S.ndp = 0;
            if (pass == 0) {
                S.ndp = 1;
                for (int r = 0; r < nranks; r++) S.slot[r] = 0;
            } else {
                for (int r = 0; r < nranks; r++) {
                    S.slot[r] = -1;
                    if (S.is_nan_rank[r]) continue;
                    for (int j = 0; j < S.ndp; j++)
                        if (S.dp[j] == S.prefix[r]) S.slot[r] = j;
                    if (S.slot[r] < 0) {
                        S.dp[S.ndp] = S.prefix[r];
                        S.slot[r] = S.ndp++;
                    }
                }
                for (int j = 0; j < S.ndp; j++) h_pre[(size_t)p * 64 + j] = S.dp[j];
            }
            live += S.ndp;
        }
        if (live == 0) break;
        if (pass > 0) {
            HIPCHK(ctx, hipMemcpyAsync(d_pre_all, h_pre.data(), sizeof(uint32_t) * h_pre.size(), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, rs_sync(ctx));  // h_pre is rewritten in the next pass
        }
        HIPCHK(ctx, hipMemsetAsync(d_hist_all, 0, (size_t)P * hist_bytes, ctx->stream));
        for (int p = 0; p < P; p++) {
            plane_state &S = st[p];
            unsigned long long *d_nan = d_hist_all + (size_t)p * hist_elems;
            unsigned long long *d_hist = d_nan + 8;
            const uint32_t *d_pre = d_pre_all + (size_t)p * 64;
            for (int b0 = 0; b0 < S.ndp; b0 += SEL_BATCH) {
                prof_scope ps(ctx, "select");
                const int nb_ = std::min(SEL_BATCH, S.ndp - b0);
                // counters + tab1 (+ tab2: at most nb_ distinct 11-bit prefixes)
                const size_t lds = pass == 0 ? (size_t)P0_COPIES * P0_STRIDE * sizeof(uint32_t)
                                             : (size_t)nb_ * SEL_BINS * sizeof(uint32_t) + SEL_BINS + (pass == 2 ? (size_t)nb_ * SEL_BINS : 0);
                launch(d_planes[p], pass, lds, pass == 0 ? d_pre : d_pre + b0, pass == 0 ? 1 : nb_, d_hist + (size_t)b0 * SEL_BINS, d_nan);
            }
        }
        HIPCHK(ctx, hipGetLastError());
        int max_tab = 1;
        for (int p = 0; p < P; p++) max_tab = std::max(max_tab, st[p].ndp);
        const size_t used_elems = 8 + (size_t)max_tab * SEL_BINS;  // per plane, compact on the host
        HIPCHK(ctx, hipMemcpy2DAsync(h_hist_all, used_elems * 8, d_hist_all, hist_bytes, used_elems * 8, (size_t)P, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, rs_sync(ctx));
        RSCHK(comm_allreduce_host(ctx, h_hist_all, (int64_t)((size_t)P * used_elems), RSSEG_I64, RSSEG_SUM));
        for (int p = 0; p < P; p++) {
            plane_state &S = st[p];
            const long long *h_nanp = h_hist_all + (size_t)p * used_elems;
            const long long *h_hist = h_nanp + 8;
            const int64_t *rk = ranks + (size_t)p * nranks;
            if (pass == 0) {
                S.n_nan = h_nanp[0];
                S.n_global = S.n_nan;
                for (int b = 0; b < SEL_BINS; b++) S.n_global += h_hist[b];
                for (int r = 0; r < nranks; r++) {
                    if (rk[r] < 0 || rk[r] >= S.n_global)
                        return rs_fail(ctx, RSSEG_ERR_INVALID, "order_stats: rank %lld outside [0,%lld)", (long long)rk[r], (long long)S.n_global);
                    S.is_nan_rank[r] = rk[r] >= S.n_global - S.n_nan;
                    S.rem[r] = rk[r];
                }
            }
            const int nbins = pass == 2 ? 1024 : SEL_BINS;
            for (int r = 0; r < nranks; r++) {
                if (S.is_nan_rank[r]) continue;
                const long long *h = h_hist + (size_t)S.slot[r] * SEL_BINS;
                int64_t acc = 0;
                int b = 0;
                for (; b < nbins; b++) {
                    if (S.rem[r] < acc + h[b]) break;
                    acc += h[b];
                }
                if (b == nbins) return rs_fail(ctx, RSSEG_ERR_HIP, "order_stats: inconsistent histogram (pass %d)", pass);
                S.rem[r] -= acc;
                S.prefix[r] = pass == 0 ? (uint32_t)b : ((S.prefix[r] << (pass == 1 ? 11 : 10)) | (uint32_t)b);
            }
        }
    }
    for (int p = 0; p < P; p++) {
        for (int r = 0; r < nranks; r++)
            out_values[(size_t)p * nranks + r] = st[p].is_nan_rank[r] ? __builtin_nanf("") : key_to_f32(st[p].prefix[r]);
        if (n_nan_out) n_nan_out[p] = st[p].n_nan;
    }
    return RSSEG_OK;
}

extern "C" int rsseg_order_stats_multi_u8(rsseg_ctx *ctx, const uint8_t *const *d_planes, int nplanes, int64_t n_local, const int64_t *ranks,
                                          int nranks, float *out_values, int64_t *n_nan_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_planes || nplanes < 1 || nplanes > SEL_MAX_PLANES || n_local < 0 || !ranks || !out_values || nranks < 1 || nranks > RSSEG_MAX_RANKS)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "order_stats_multi_u8: bad arguments (planes=%d n=%lld nranks=%d)", nplanes, (long long)n_local, nranks);
    for (int p = 0; p < nplanes; p++)
        if (!d_planes[p] || ((uintptr_t)d_planes[p] & 15) != 0)
            return rs_fail(ctx, RSSEG_ERR_INVALID, "order_stats_multi_u8: plane %d null or not 16-byte aligned", p);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int P = nplanes;
    const size_t used = 256;
    RSCHK(ws_reserve(ctx, (size_t)P * used * 8));
    RSCHK(pin_reserve(ctx, (size_t)P * used * 8));
    unsigned long long *d_hist = (unsigned long long *)ctx->d_ws;
    long long *h_hist = (long long *)ctx->h_pin;
    HIPCHK(ctx, hipMemsetAsync(d_hist, 0, (size_t)P * used * 8, ctx->stream));
    const int grid = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ceil_div64(n_local >> 4, 1024)));
    for (int p = 0; p < P; p++) {
        prof_scope ps(ctx, "select");
        hipLaunchKernelGGL((k1_hist_u8<1024>), dim3(grid), dim3(1024), 0, ctx->stream, d_planes[p], n_local, d_hist + (size_t)p * used);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(h_hist, d_hist, (size_t)P * used * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, rs_sync(ctx));
    RSCHK(comm_allreduce_host(ctx, h_hist, (int64_t)((size_t)P * used), RSSEG_I64, RSSEG_SUM));
    for (int p = 0; p < P; p++) {
        const long long *h = h_hist + (size_t)p * used;
        const int64_t *rk = ranks + (size_t)p * nranks;
        int64_t n_global = 0;
        for (int b = 0; b < 256; b++) n_global += h[b];
        for (int r = 0; r < nranks; r++) {
            if (rk[r] < 0 || rk[r] >= n_global)
                return rs_fail(ctx, RSSEG_ERR_INVALID, "order_stats: rank %lld outside [0,%lld)", (long long)rk[r], (long long)n_global);
            int64_t acc = 0;
            int b = 0;
            for (; b < 256; b++) {
                if (rk[r] < acc + h[b]) break;
                acc += h[b];
            }
            out_values[(size_t)p * nranks + r] = (float)b;
        }
        if (n_nan_out) n_nan_out[p] = 0;
    }
    return RSSEG_OK;
}

extern "C" int rsseg_order_stats_f32(rsseg_ctx *ctx, const float *d_x, int64_t n_local, const int64_t *ranks, int nranks,
                                     float *out_values, int64_t *n_nan_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_x || n_local < 0 || !ranks || !out_values || nranks < 1 || nranks > RSSEG_MAX_RANKS)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "order_stats: bad arguments (n=%lld nranks=%d)", (long long)n_local, nranks);
    if (((uintptr_t)d_x & 15) != 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "order_stats: plane must be 16-byte aligned");
    return order_stats_core(ctx, &d_x, 1, n_local, ranks, nranks, out_values, n_nan_out);
}

extern "C" int rsseg_order_stats_multi_f32(rsseg_ctx *ctx, const float *const *d_planes, int nplanes, int64_t n_local,
                                           const int64_t *ranks, int nranks, float *out_values, int64_t *n_nan_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_planes || nplanes < 1 || nplanes > SEL_MAX_PLANES || n_local < 0 || !ranks || !out_values || nranks < 1 || nranks > RSSEG_MAX_RANKS)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "order_stats_multi: bad arguments (planes=%d n=%lld nranks=%d)", nplanes, (long long)n_local, nranks);
    for (int p = 0; p < nplanes; p++)
        if (!d_planes[p] || ((uintptr_t)d_planes[p] & 15) != 0)
            return rs_fail(ctx, RSSEG_ERR_INVALID, "order_stats_multi: plane %d null or not 16-byte aligned", p);
    return order_stats_core(ctx, d_planes, nplanes, n_local, ranks, nranks, out_values, n_nan_out);
}
