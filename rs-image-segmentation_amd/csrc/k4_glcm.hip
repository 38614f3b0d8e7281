// K4 — GLCM texture windows (contrast, dissimilarity, homogeneity, energy, correlation; mean over the
// angles 0/45/90/135 degrees at distance 1; symmetric, normalised co-occurrence).
//
// Replaces the Python double loop of calculate_glcm_features (reference modules/features/indices.py:
// 283-305: graycomatrix + 5 x graycoprops per window; scikit-image semantics restated in oracle/oracle.c).
//
// The co-occurrence matrix is never formed.  All five properties follow from exact integer
// statistics of the window's pixel pairs (a, b), per angle:
//     np = #pairs, S1 = sum|a-b|, S2 = sum(a-b)^2, Hq = sum round(2^52/(1+(a-b)^2)),
//     M1 = sum(a+b), M2 = sum(a^2+b^2), Mx = sum 2ab,
//     A  = sum_ij (G_ij+G_ji)^2 = 2*(np + 2*E_all) + 2*(D + 2*E_diag)
//          E_all  = #{p<q : unordered(a_p,b_p) == unordered(a_q,b_q)},  D = #{p : a_p == b_p},
//          E_diag = #{p<q : a_p == b_p == a_q == b_q}
// and the float64 formulas at the end are those of oracle.c (mode 1), so results are bit-identical.
//
// Two kernels:
//   k4_glcm_thread<WIN>  one thread per window, window in registers, pair statistics by direct
//                        comparison (WIN <= 7; the dense step-1 case of BASELINE config 3).
//                        Integer-VALU-bound: ~WIN^4 compare-accumulates per window and angle.
//   k4_glcm_wg           one workgroup per window with an LDS co-occurrence histogram (any window
//                        size, levels <= 64; the reference's 21x21 / step 21 default).
#include <utility>

#include "common.h"

__constant__ long long c_glcm_hq[256];

static const int H_DR[4] = {0, 1, 1, 1};
static const int H_DC[4] = {1, 1, 0, -1};

struct glcm_out {
    float *p[5];  // contrast, dissimilarity, homogeneity, energy, correlation
};

struct glcm_stats {
    long long np, S1, S2, Hq, M1, M2, Mx, A;
};

__device__ __forceinline__ void glcm_props(const glcm_stats &s, double &pc, double &pd, double &ph, double &pe, double &pr)
{
    const double npd = (double)s.np, tot = (double)(2 * s.np);
    pc = (double)s.S2 / npd;
    pd = (double)s.S1 / npd;
    ph = ((double)s.Hq * (1.0 / 4503599627370496.0)) / npd;
    pe = sqrt((double)s.A / (tot * tot));
    const long long den = s.M2 * (2 * s.np) - s.M1 * s.M1, num = s.Mx * (2 * s.np) - s.M1 * s.M1;
    pr = den == 0 ? 1.0 : (double)num / (double)den;
}

// ---- compile-time machinery: every register array below is indexed by constants only ----------
template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Batcher odd-even merge sort network for P elements (comparators touching padded slots pruned)
template <int P> struct sort_net {
    int a[P * 12], b[P * 12];
    int n;
};
template <int P> constexpr sort_net<P> make_sort_net()
{
    sort_net<P> s{};
    int m = 1;
    while (m < P) m *= 2;
    int n = 0;
    for (int p = 1; p < m; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j + k < m; j += 2 * k)
                for (int i = 0; i < k && i + j + k < m; i++)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && i + j + k < P) {
                        s.a[n] = i + j;
                        s.b[n] = i + j + k;
                        n++;
                    }
    s.n = n;
    return s;
}

template <int P> struct net_holder {
    static constexpr sort_net<P> net = make_sort_net<P>();
};

template <int WIN, int A> struct angle_geom {
    static constexpr int DR = A == 0 ? 0 : 1;
    static constexpr int DC = A == 0 ? 1 : (A == 1 ? 1 : (A == 2 ? 0 : -1));
    static constexpr int R1 = DR > 0 ? WIN - 1 : WIN;
    static constexpr int C0 = DC < 0 ? 1 : 0;
    static constexpr int C1 = DC > 0 ? WIN - 1 : WIN;
    static constexpr int PW = C1 - C0;
    static constexpr int P = R1 * PW;
};

// pair statistics of one angle; key = diag<<16 | lo<<8 | hi, sorted, equal runs counted
// window rows packed 4 pixels per register: w[r][0] = px 0..3, w[r][1] = px 4..7
template <int R, int C> __device__ __forceinline__ int px_at(const unsigned (&w)[8][2])
{
    return (int)((w[R][C >> 2] >> (8 * (C & 3))) & 0xffu);
}

template <int WIN, int A>
__device__ __forceinline__ void glcm_angle(const unsigned (&w)[8][2], const long long *__restrict__ hq, glcm_stats &s)
{
    using G = angle_geom<WIN, A>;
    constexpr int P = G::P;
    unsigned key[P];
    unsigned S1 = 0;
    int M1 = 0, M2 = 0, Mx = 0;
    long long Hq = 0;
    static_for<P>([&](auto I) {
        constexpr int p = I;
        constexpr int r = p / G::PW, c = G::C0 + p % G::PW;
        const int x = px_at<r, c>(w), y = px_at<r + G::DR, c + G::DC>(w);
        const int lo = x < y ? x : y, hi = x < y ? y : x;
        const unsigned d = (unsigned)(hi - lo);
        key[p] = ((d == 0 ? 1u : 0u) << 16) | ((unsigned)lo << 8) | (unsigned)hi;
        S1 += d;
        Hq += hq[d];
        M1 += x + y;
        M2 += x * x + y * y;
        Mx += 2 * x * y;
        // keep at most 8 LUT reads in flight: unconstrained, the scheduler issues all P of them up front
        // and their 2P result registers push the kernel to one wave per SIMD
        if constexpr (p % 8 == 7) __builtin_amdgcn_sched_barrier(0);
    });
    static_for<net_holder<P>::net.n>([&](auto I) {
        constexpr int ia = net_holder<P>::net.a[I], ib = net_holder<P>::net.b[I];
        const unsigned ka = key[ia], kb = key[ib];
        key[ia] = ka < kb ? ka : kb;
        key[ib] = ka < kb ? kb : ka;
    });
    // E2 = sum over equal runs of w * c(c-1)/2 with w = 2 on the diagonal, 1 elsewhere
    unsigned E2 = 0, t = 0, D = key[0] >> 16;
    static_for<P - 1>([&](auto I) {
        constexpr int i = I + 1;
        const unsigned w = 1u + (key[i] >> 16);
        t = key[i] == key[i - 1] ? t + w : 0u;
        E2 += t;
        D += key[i] >> 16;
    });
    s.np = P;
    s.S1 = S1;
    s.S2 = (long long)M2 - (long long)Mx;  // sum (a-b)^2 = sum(a^2+b^2) - sum 2ab
    s.Hq = Hq;
    s.M1 = M1;
    s.M2 = M2;
    s.Mx = Mx;
    s.A = 2ll * (P + (int)D) + 4ll * (long long)E2;
}

// compiler fence: the packed window is redefined (as far as the compiler can tell) at the top of every
// angle iteration, so the four angle bodies cannot be hoisted out of the loop or merged
template <int WIN> __device__ __forceinline__ void opaque_window(unsigned (&w)[8][2])
{
#if defined(__HIP_DEVICE_COMPILE__)
    static_for<WIN>([&](auto I) {
        unsigned &a = w[I][0];
        unsigned &b = w[I][1];
        asm volatile("" : "+v"(a), "+v"(b));
    });
#endif
}

template <int WIN>
__global__ __launch_bounds__(256) void k4_glcm_thread(const uint8_t *__restrict__ q, int H, int W, int step, int oh, int ow,
                                                      glcm_out out)
{
    __shared__ long long hq[256];
    hq[threadIdx.x] = c_glcm_hq[threadIdx.x];
    __syncthreads();
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= ow || oy >= oh) return;
    unsigned w[8][2];
    {
        const uint8_t *wp = q + (size_t)(oy * step) * W + (size_t)ox * step;
        static_for<WIN>([&](auto I) {
            constexpr int r = I;
            unsigned lo = 0, hi = 0;
            static_for<WIN>([&](auto J) {
                constexpr int c = J;
                const unsigned b = wp[(size_t)r * W + c];
                if constexpr (c < 4) lo |= b << (8 * c);
                else hi |= b << (8 * (c - 4));
            });
            w[r][0] = lo;
            w[r][1] = hi;
        });
    }
    // The four angles run through a RUNTIME loop on purpose: unrolled, the compiler merges their common
    // sub-expressions and the combined live ranges (4 x ~42 keys + 49 pixels) spill to scratch.
    double sc = 0, sd = 0, sh = 0, se = 0, sr = 0;
#pragma nounroll
    for (int a = 0; a < 4; a++) {
        opaque_window<WIN>(w);  // the angle bodies are loop-invariant: without this they are all hoisted
        glcm_stats s;
        switch (a) {
        case 0: glcm_angle<WIN, 0>(w, hq, s); break;
        case 1: glcm_angle<WIN, 1>(w, hq, s); break;
        case 2: glcm_angle<WIN, 2>(w, hq, s); break;
        default: glcm_angle<WIN, 3>(w, hq, s); break;
        }
        double c, d, h, e, r;
        glcm_props(s, c, d, h, e, r);
        // (((p0 + p1) + p2) + p3): adding to an exact 0.0 first does not change p0
        sc = sc + c; sd = sd + d; sh = sh + h; se = se + e; sr = sr + r;
    }
    const size_t o = (size_t)oy * ow + ox;
    if (out.p[0]) out.p[0][o] = (float)(sc / 4.0);
    if (out.p[1]) out.p[1][o] = (float)(sd / 4.0);
    if (out.p[2]) out.p[2][o] = (float)(sh / 4.0);
    if (out.p[3]) out.p[3][o] = (float)(se / 4.0);
    if (out.p[4]) out.p[4][o] = (float)(sr / 4.0);
}

// one workgroup per window; LDS histogram of ordered cells [levels][levels]
__global__ __launch_bounds__(256) void k4_glcm_wg(const uint8_t *__restrict__ q, int H, int W, int levels, int win, int step,
                                                  int oh, int ow, glcm_out out)
{
    extern __shared__ unsigned int hist[];  // levels*levels
    __shared__ long long red[4][8];
    __shared__ double props[4][5];
    const int ox = blockIdx.x, oy = blockIdx.y;
    const uint8_t *wp = q + (size_t)(oy * step) * W + (size_t)ox * step;
    const int LL = levels * levels;
    for (int a = 0; a < 4; a++) {
        const int dr = a == 0 ? 0 : 1, dc = a == 0 ? 1 : (a == 1 ? 1 : (a == 2 ? 0 : -1));
        const int r1 = dr > 0 ? win - dr : win, c0 = dc < 0 ? -dc : 0, c1 = dc > 0 ? win - dc : win;
        const int pw = c1 - c0, P = r1 * pw;
        for (int i = threadIdx.x; i < LL; i += 256) hist[i] = 0;
        __syncthreads();
        long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // np S1 S2 Hq M1 M2 Mx A
        for (int p = threadIdx.x; p < P; p += 256) {
            const int r = p / pw, c = c0 + p % pw;
            const int x = wp[(size_t)r * W + c], y = wp[(size_t)(r + dr) * W + (c + dc)];
            if (x >= levels || y >= levels) continue;
            atomicAdd(&hist[x * levels + y], 1u);
            const int d = x > y ? x - y : y - x;
            st[0] += 1; st[1] += d; st[2] += d * d; st[3] += c_glcm_hq[d];
            st[4] += x + y; st[5] += x * x + y * y; st[6] += 2 * x * y;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < LL; i += 256) {
            const int x = i / levels, y = i % levels;
            const long long g = (long long)hist[i] + (long long)hist[y * levels + x];
            st[7] += g * g;
        }
#pragma unroll
        for (int t = 0; t < 8; t++) {
            long long s = wave_sum(st[t]);
            if (lane_id() == 0) red[threadIdx.x >> 6][t] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            glcm_stats s;
            long long *sp = &s.np;
            for (int t = 0; t < 8; t++) sp[t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
            if (s.np == 0) {
                props[a][0] = props[a][1] = props[a][2] = props[a][3] = 0.0;
                props[a][4] = 1.0;
            } else {
                glcm_props(s, props[a][0], props[a][1], props[a][2], props[a][3], props[a][4]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 5 && out.p[threadIdx.x]) {
        const int t = threadIdx.x;
        out.p[t][(size_t)oy * ow + ox] = (float)((((props[0][t] + props[1][t]) + props[2][t]) + props[3][t]) / 4.0);
    }
}

static bool g_hq_ready[64] = {false};

extern "C" int rsseg_glcm_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int levels, int win, int step,
                             float *const *d_props)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_q || !d_props || H < 1 || W < 1 || levels < 2 || levels > 256 || win < 2 || win > H || win > W || step < 1)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "glcm: bad arguments (H=%d W=%d levels=%d win=%d step=%d)", H, W, levels, win, step);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!g_hq_ready[ctx->device & 63]) {
        long long lut[256];
        for (int d = 0; d < 256; d++) lut[d] = llrint(4503599627370496.0 / (1.0 + (double)d * (double)d));
        HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_glcm_hq), lut, sizeof(lut)));
        g_hq_ready[ctx->device & 63] = true;
    }
    const int oh = (H - win) / step + 1, ow = (W - win) / step + 1;
    glcm_out out;
    for (int i = 0; i < 5; i++) out.p[i] = d_props[i];
    {
        prof_scope ps(ctx, "glcm");
        const dim3 tg((ow + 63) / 64, (oh + 3) / 4);
        if (win == 7)
            hipLaunchKernelGGL(k4_glcm_thread<7>, tg, dim3(256), 0, ctx->stream, d_q, H, W, step, oh, ow, out);
        else if (win == 5)
            hipLaunchKernelGGL(k4_glcm_thread<5>, tg, dim3(256), 0, ctx->stream, d_q, H, W, step, oh, ow, out);
        else if (win == 3)
            hipLaunchKernelGGL(k4_glcm_thread<3>, tg, dim3(256), 0, ctx->stream, d_q, H, W, step, oh, ow, out);
        else {
            if (levels > 64) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "glcm: levels > 64 with window %d not supported", win);
            if (ow > 2147483647 || oh > 65535) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "glcm: output map too tall for the workgroup-per-window kernel");
            hipLaunchKernelGGL(k4_glcm_wg, dim3(ow, oh), dim3(256), sizeof(unsigned int) * levels * levels, ctx->stream, d_q, H, W,
                               levels, win, step, oh, ow, out);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}
