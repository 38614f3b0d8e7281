// K4 — GLCM texture windows (contrast, dissimilarity, homogeneity, energy, correlation; mean over the
// angles 0/45/90/135 degrees at distance 1; symmetric, normalised co-occurrence).
//
// Replaces the Python double loop of calculate_glcm_features (reference modules/features/indices.py:
// 283-305: graycomatrix + 5 x graycoprops per window; scikit-image semantics restated in oracle/oracle.c).
//
// The co-occurrence matrix is never formed.  All five properties follow from exact integer
// statistics of the window's pixel pairs (a, b), per angle:
//     np = #pairs, S1 = sum|a-b|, XY = sum a*b, M1 = sum(a+b), M2 = sum(a^2+b^2),  S2 = M2 - 2 XY,
//     Hq = sum round(2^52/(1+(a-b)^2)),
//     A  = sum_ij (G_ij+G_ji)^2 = 2*(np + D) + 4*E2,  D = #{p : a_p == b_p},
//          E2 = sum over runs of equal unordered keys of w*c(c-1)/2, w = 2 on the diagonal else 1
// and the float64 formulas at the end are those of oracle.c (mode 1), so results are bit-identical.
// Angles 0/90 (np = w(w-1)) and 45/135 (np = (w-1)^2) are combined over common denominators, which
// leaves 8 float64 divisions and 4 square roots per window.
//
// Two kernels:
//   k4_glcm_thread<WIN>  one thread per window (WIN <= 7, levels <= 64; the dense step-1 case of BASELINE
//        config 3).  The window lives in 2*WIN registers, 4 pixels per register: pair moments come from
//        v_sad_u8 / v_dot4_u32_u8 on whole rows; the unordered pair keys (13 bits) of TWO angles share a
//        register and go through one Batcher network of v_pk_min_u16 / v_pk_max_u16, then a packed
//        run-length pass.  Integer-VALU-bound, not HBM-bound (1 B/px in, 20 B/px out).
//   k4_glcm_wg           one workgroup per window with an LDS co-occurrence histogram (any window
//        size, levels <= 64; the reference's 21x21 / step 21 default).
#include <utility>

#include <mutex>

#include "common.h"

__constant__ long long c_glcm_hq[256];
__device__ long long g_glcm_hq2[1024];  // pair sums hq[dA] + hq[dB] at [dB * 32 + dA] (levels <= 32)

static const int H_DR[4] = {0, 1, 1, 1};
static const int H_DC[4] = {1, 1, 0, -1};

struct glcm_out {
    float *p[5];  // contrast, dissimilarity, homogeneity, energy, correlation
};

struct glcm_stats {
    long long np, S1, S2, Hq, M1, M2, Mx, A;
};

// correlation of one angle from exact integers (oracle.c mode 1)
__device__ __forceinline__ double glcm_corr(long long np, long long M1, long long M2, long long Mx)
{
    const long long den = M2 * (2 * np) - M1 * M1, num = Mx * (2 * np) - M1 * M1;
    return den == 0 ? 1.0 : (double)num / (double)den;
}

// group sums: g0 = angles 0 and 90 degrees (na pairs each), g1 = 45 and 135 degrees (nb pairs each)
struct glcm_group {
    long long S1, S2, Hq;
    double sq;  // sqrt(A_a) + sqrt(A_b)
};

// RN(a / b) for a divisor known on the host: with y = RN(1/b), q = RN(a*y) is a faithful quotient, the fma residual
// r = a - q*b is exact and RN(q + r*y) is the correctly rounded quotient (Markstein) - three instructions instead of
// the ~12-instruction IEEE sequence.  The host checks the precondition (significand of b not all ones).
__device__ __forceinline__ double div_const(double a, double b, double y)
{
    const double q = a * y;
    const double r = fma(-q, b, a);
    return fma(r, y, q);
}

struct glcm_consts {
    double den4, den8, rden4, rden8;  // 4*na*nb, 8*na*nb and their correctly rounded reciprocals
};

__device__ __forceinline__ void glcm_finish(const glcm_group &g0, const glcm_group &g1, long long na, long long nb, double r0,
                                            double r1, double r2, double r3, size_t o, const glcm_out &out, const glcm_consts &gc)
{
    const double dna = (double)na, dnb = (double)nb;
    if (out.p[0]) out.p[0][o] = (float)div_const((double)(g0.S2 * nb + g1.S2 * na), gc.den4, gc.rden4);
    if (out.p[1]) out.p[1][o] = (float)div_const((double)(g0.S1 * nb + g1.S1 * na), gc.den4, gc.rden4);
    if (out.p[2]) {
        const double t1 = (double)g1.Hq * dna;
        const double num = fma((double)g0.Hq, dnb, t1);
        out.p[2][o] = (float)(div_const(num, gc.den4, gc.rden4) * (1.0 / 4503599627370496.0));
    }
    if (out.p[3]) {
        const double t1 = g1.sq * dna;
        const double num = fma(g0.sq, dnb, t1);
        out.p[3][o] = (float)div_const(num, gc.den8, gc.rden8);
    }
    if (out.p[4]) out.p[4][o] = (float)((((r0 + r1) + r2) + r3) * 0.25);
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

// Sorting network for P elements: Batcher's merge exchange for an arbitrary count (Knuth, TAOCP 5.2.2, Algorithm M) —
// 309 comparators at P = 42, 241 at P = 36 (the power-of-two odd-even merge sort with the padded slots pruned: 327, 268)
template <int P> struct sort_net {
    int a[P * 12], b[P * 12];
    int n;
};
template <int P> constexpr sort_net<P> make_sort_net()
{
    sort_net<P> s{};
    int t = 0;
    while ((1 << t) < P) t++;
    int n = 0;
    for (int p = t > 0 ? 1 << (t - 1) : 0; p > 0; p /= 2) {
        int q = 1 << (t - 1), r = 0, d = p;
        while (d > 0) {
            for (int i = 0; i + d < P; i++)
                if ((i & p) == r) {
                    s.a[n] = i;
                    s.b[n] = i + d;
                    n++;
                }
            d = q - p;
            q /= 2;
            r = p;
        }
    }
    s.n = n;
    return s;
}
template <int P> struct net_holder {
    static constexpr sort_net<P> net = make_sort_net<P>();
};

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
// v_pk_min_u16 the optimiser cannot see through: min(x, 1) written in C++ is canonicalised to a compare + select,
// which has no packed form and costs four instructions per register instead of one.
__device__ __forceinline__ unsigned pk_min_opaque(unsigned a, unsigned b)
{
    unsigned r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
#endif
    return r;
}
// max(a - b, 0) on both 16-bit halves (v_pk_sub_u16 with the clamp bit): 1 - d saturates to [d == 0]
__device__ __forceinline__ unsigned pk_sub_sat(unsigned a, unsigned b)
{
    unsigned r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
#endif
    return r;
}
// a * b on both 16-bit halves (v_pk_mul_lo_u16)
__device__ __forceinline__ unsigned pk_mul(unsigned a, unsigned b)
{
    unsigned r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
#endif
    return r;
}
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (us2)(__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (us2)(__builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b)));
}


__device__ __forceinline__ void pin32(unsigned &v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
}
__device__ __forceinline__ void pin64(long long &v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
}


// pair moments of one angle from whole packed rows: S1 = sum|a-b|, XY = sum ab, M2 = sum a^2+b^2, M1 = sum a+b
template <int WIN, int DR, int DC>
__device__ __forceinline__ void row_moments(const unsigned (&w)[8][2], unsigned &S1, unsigned &XY)
{
    constexpr int NB = DC == 0 ? WIN : WIN - 1;  // bytes taking part per row
    constexpr unsigned KLO = NB >= 4 ? 0xffffffffu : ((1u << (8 * (NB & 3))) - 1u);
    constexpr unsigned KHI = NB <= 4 ? 0u : ((NB >= 8) ? 0xffffffffu : ((1u << (8 * (NB - 4))) - 1u));
    constexpr int R1 = DR > 0 ? WIN - 1 : WIN;
    S1 = XY = 0;
    static_for<R1>([&](auto I) {
        constexpr int r = I;
        const unsigned a0 = w[r][0], a1 = w[r][1], b0 = w[r + DR][0], b1 = w[r + DR][1];
        unsigned Alo, Ahi, Blo, Bhi;
        if constexpr (DC == 1) {         // (c, c+1): A = bytes 0..WIN-2, B = bytes 1..WIN-1
            Alo = a0 & KLO; Ahi = a1 & KHI;
            Blo = __builtin_amdgcn_alignbyte(b1, b0, 1); Bhi = b1 >> 8;
        } else if constexpr (DC == 0) {
            Alo = a0; Ahi = a1; Blo = b0; Bhi = b1;
        } else {                         // (c, c-1): A = bytes 1..WIN-1 of row r, B = bytes 0..WIN-2 of row r+1
            Alo = __builtin_amdgcn_alignbyte(a1, a0, 1); Ahi = a1 >> 8;
            Blo = b0 & KLO; Bhi = b1 & KHI;
        }
        S1 = __builtin_amdgcn_sad_u8(Alo, Blo, S1);
        XY = __builtin_amdgcn_udot4(Alo, Blo, XY, false);
        if constexpr (WIN > 4) {
            S1 = __builtin_amdgcn_sad_u8(Ahi, Bhi, S1);
            XY = __builtin_amdgcn_udot4(Ahi, Bhi, XY, false);
        }
    });
}

// M1 = sum(a+b) and M2 = sum(a^2+b^2) over the pairs of each angle, from sums the four angles share: with T the
// window total, R0/RL the first/last row, C0/CL the first/last column and the corners (L = WIN-1),
//   0 deg   (r,c)-(r,c+1):    2T - C0 - CL
//   90 deg  (r,c)-(r+1,c):    2T - R0 - RL
//   45 deg  (r,c)-(r+1,c+1):  2T - R0 - RL - C0 - CL + w00 + wLL
//   135 deg (r,c)-(r+1,c-1):  2T - R0 - RL - C0 - CL + w0L + wL0
// and the same with squares (a pixel is the first member of a pair unless it lies in the last row/column the angle
// excludes, the second member unless it lies in the first).  m1[] / m2[] are indexed 0, 45, 90, 135 degrees.
template <int WIN> __device__ __forceinline__ void window_m1m2(const unsigned (&w)[8][2], unsigned (&m1)[4], unsigned (&m2)[4])
{
    constexpr int L = WIN - 1;
    unsigned rs[WIN], rq[WIN];
    static_for<WIN>([&](auto I) {
        constexpr int r = I;
        rs[r] = __builtin_amdgcn_sad_u8(w[r][0], 0u, 0u);
        rq[r] = __builtin_amdgcn_udot4(w[r][0], w[r][0], 0u, false);
        if constexpr (WIN > 4) {
            rs[r] = __builtin_amdgcn_sad_u8(w[r][1], 0u, rs[r]);
            rq[r] = __builtin_amdgcn_udot4(w[r][1], w[r][1], rq[r], false);
        }
    });
    unsigned T = 0, T2 = 0;
    static_for<WIN>([&](auto I) { T += rs[I]; T2 += rq[I]; });
    // first / last column gathered into packed registers (rows 0..3, rows 4..)
    constexpr unsigned s0 = 0x0c0c0400u;                                         // byte 0 of both operands
    constexpr unsigned sl = 0x0c0c0000u | ((4u + (L & 3)) << 8) | (unsigned)(L & 3);  // byte L&3 of both operands
    auto column = [&](auto lastc, unsigned &lo, unsigned &hi) {
        constexpr bool LAST = decltype(lastc)::value;
        constexpr int k = LAST ? (L >> 2) : 0;
        constexpr unsigned sel = LAST ? sl : s0;
        // __builtin_amdgcn_perm(hi_src, lo_src, sel): selector bytes 0..3 pick from lo_src, 4..7 from hi_src
        const unsigned p01 = __builtin_amdgcn_perm(w[1 < WIN ? 1 : 0][k], w[0][k], sel);
        const unsigned p23 = WIN > 2 ? __builtin_amdgcn_perm(w[3 < WIN ? 3 : 0][k], w[2 < WIN ? 2 : 0][k], sel) : 0u;
        lo = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
        if constexpr (WIN == 3) lo &= 0x00ffffffu;
        hi = 0u;
        if constexpr (WIN > 4) {
            const unsigned p45 = __builtin_amdgcn_perm(w[5 < WIN ? 5 : 0][k], w[4][k], sel);
            const unsigned p6 = WIN > 6 ? __builtin_amdgcn_perm(0u, w[6 < WIN ? 6 : 0][k], LAST ? (0x0c0c0c00u | (unsigned)(L & 3)) : 0x0c0c0c00u) : 0u;
            hi = __builtin_amdgcn_perm(p6, p45, 0x05040100u);
            if constexpr (WIN == 5) hi &= 0x000000ffu;
        }
    };
    unsigned c0lo, c0hi, cllo, clhi;
    column(std::false_type{}, c0lo, c0hi);
    column(std::true_type{}, cllo, clhi);
    unsigned C0 = __builtin_amdgcn_sad_u8(c0lo, 0u, 0u), CL = __builtin_amdgcn_sad_u8(cllo, 0u, 0u);
    unsigned C0q = __builtin_amdgcn_udot4(c0lo, c0lo, 0u, false), CLq = __builtin_amdgcn_udot4(cllo, cllo, 0u, false);
    if constexpr (WIN > 4) {
        C0 = __builtin_amdgcn_sad_u8(c0hi, 0u, C0);
        CL = __builtin_amdgcn_sad_u8(clhi, 0u, CL);
        C0q = __builtin_amdgcn_udot4(c0hi, c0hi, C0q, false);
        CLq = __builtin_amdgcn_udot4(clhi, clhi, CLq, false);
    }
    const unsigned w00 = c0lo & 0xffu, w0L = cllo & 0xffu;
    const unsigned wL0 = WIN > 4 ? (c0hi >> (8 * (L - 4))) & 0xffu : (c0lo >> (8 * L)) & 0xffu;
    const unsigned wLL = WIN > 4 ? (clhi >> (8 * (L - 4))) & 0xffu : (cllo >> (8 * L)) & 0xffu;
    const unsigned R = rs[0] + rs[L], Rq = rq[0] + rq[L], C = C0 + CL, Cq = C0q + CLq;
    m1[0] = 2 * T - C;
    m1[2] = 2 * T - R;
    m1[1] = 2 * T - R - C + w00 + wLL;
    m1[3] = 2 * T - R - C + w0L + wL0;
    m2[0] = 2 * T2 - Cq;
    m2[2] = 2 * T2 - Rq;
    m2[1] = 2 * T2 - Rq - Cq + w00 * w00 + wLL * wLL;
    m2[3] = 2 * T2 - Rq - Cq + w0L * w0L + wL0 * wL0;
}

// One angle group: G = 0 -> angles 0 (0,1) and 90 (1,0) degrees, G = 1 -> 45 (1,1) and 135 (1,-1).
// Both angles of a group have the same pair count P, so their keys share registers (low / high half).
// The window holds the pixels PRE-SCALED by 2^SH (SH = 3 for levels <= 32, 2 for levels <= 64: still one byte), so
// that the scaled |a-b| gives the byte offset into the Hq table without a shift and all sums are exact multiples
// that are shifted back at the end.  Key = lo' << 8 | hi' | diag (primed = scaled: the low SH bits of a scaled
// value are zero, so bit 0 is free for the flag [a == b]): one v_lshl_or and one v_or on both halves at once.
struct glcm_raw {  // what one angle group leaves behind (S1 / XY still carry the 2^SH pre-scaling)
    unsigned S1, XYa, XYb;
    long long Hq;
    double sq;
};
template <int WIN, int G, int SH>
__device__ __forceinline__ void glcm_group_stats(const unsigned (&w)[8][2], const long long *__restrict__ hq, glcm_raw &g)
{
    constexpr int P = G == 0 ? WIN * (WIN - 1) : (WIN - 1) * (WIN - 1);
    unsigned K[P];
    long long HqA = 0, HqB = 0;  // SH == 3: hq is the 32 x 32 table of PAIR sums hq[dB][dA] = Hq(dA) + Hq(dB): one read per register
    static_for<P>([&](auto I) {
        constexpr int p = I;
        // pixel positions of pair p in angle A (low half) and angle B (high half)
        constexpr int rxa = G == 0 ? p / (WIN - 1) : p / (WIN - 1), cxa = G == 0 ? p % (WIN - 1) : p % (WIN - 1);
        constexpr int rya = G == 0 ? rxa : rxa + 1, cya = cxa + 1;
        constexpr int rxb = G == 0 ? p / WIN : p / (WIN - 1), cxb = G == 0 ? p % WIN : p % (WIN - 1) + 1;
        constexpr int ryb = rxb + 1, cyb = G == 0 ? cxb : cxb - 1;
        // v_perm_b32: result byte 0 <- angle-A pixel, byte 2 <- angle-B pixel, bytes 1 and 3 <- 0
        constexpr unsigned selx = (unsigned)(cxa & 3) | (0x0cu << 8) | ((unsigned)(4 + (cxb & 3)) << 16) | (0x0cu << 24);
        constexpr unsigned sely = (unsigned)(cya & 3) | (0x0cu << 8) | ((unsigned)(4 + (cyb & 3)) << 16) | (0x0cu << 24);
        const unsigned x = __builtin_amdgcn_perm(w[rxb][cxb >> 2], w[rxa][cxa >> 2], selx);
        const unsigned y = __builtin_amdgcn_perm(w[ryb][cyb >> 2], w[rya][cya >> 2], sely);
        const unsigned lo = pk_min(x, y), hi = pk_max(x, y);
        const unsigned d = pk_sub(hi, lo);
        const unsigned one = 0x00010001u;
        const unsigned diag = pk_sub_sat(one, d);  // [a == b] in both halves
        K[p] = ((lo << 8) | hi) | diag;            // each half stays below 2^16: the 32-bit shift does not cross
        pin32(K[p]);  // materialise the packed key now (short live ranges)
        if constexpr (SH == 3) {
            // d = 8*dA | (8*dB) << 16  ->  byte offset 8 * (dA + 32 * dB)
            // byte offset 8 * (dA + 32 * dB) = lo16(d) * 1 + hi16(d) * 32 in one v_dot2_u32_u16
            const unsigned off = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, d), (us2){1, 32}, 0u, false);
            HqA += *reinterpret_cast<const long long *>(reinterpret_cast<const char *>(hq) + off);
        } else {
            HqA += *reinterpret_cast<const long long *>(reinterpret_cast<const char *>(hq) + ((d & 0xffffu) << (3 - SH)));
            HqB += *reinterpret_cast<const long long *>(reinterpret_cast<const char *>(hq) + ((d >> 16) << (3 - SH)));
        }
        if constexpr (p % 6 == 5) {  // every 6 pairs: up to 12 LUT reads (24 registers) in flight, not 2P
            pin64(HqA);
            pin64(HqB);
        }
    });
    __builtin_amdgcn_sched_barrier(0);  // phase boundaries keep the phases' live ranges from overlapping
    unsigned S1a, XYa, S1b, XYb;
    if constexpr (G == 0) {
        row_moments<WIN, 0, 1>(w, S1a, XYa);
        row_moments<WIN, 1, 0>(w, S1b, XYb);
    } else {
        row_moments<WIN, 1, 1>(w, S1a, XYa);
        row_moments<WIN, 1, -1>(w, S1b, XYb);
    }
    __builtin_amdgcn_sched_barrier(0);
    // sort both halves at once
    static_for<net_holder<P>::net.n>([&](auto I) {
        constexpr int ia = net_holder<P>::net.a[I], ib = net_holder<P>::net.b[I];
        const unsigned ka = K[ia], kb = K[ib];
        K[ia] = pk_min(ka, kb);
        K[ib] = pk_max(ka, kb);
    });
    __builtin_amdgcn_sched_barrier(0);
    // packed run-length: t = equal-to-previous ? t + w : 0 (w = 2 on the diagonal, else 1); E2 += t; D += diag.
    // All fields stay far below 2^16, so plain 32-bit adds (v_add3_u32) serve both halves.
    const unsigned one = 0x00010001u;
    unsigned E2 = 0, t = 0, D = K[0] & one;
    static_for<P - 1>([&](auto I) {
        constexpr int i = I + 1;
        const unsigned diag = K[i] & one;
        const unsigned eq = pk_sub_sat(one, K[i] ^ K[i - 1]);     // 1 where equal, 0 where different
        t = pk_mul(t + one + diag, eq);
        E2 += t;
        D += diag;
    });
    const long long Aa = 2ll * (P + (int)(D & 0xffffu)) + 4ll * (long long)(E2 & 0xffffu);
    const long long Ab = 2ll * (P + (int)(D >> 16)) + 4ll * (long long)(E2 >> 16);
    g.S1 = S1a + S1b;
    g.XYa = XYa;
    g.XYb = XYb;
    g.Hq = HqA + HqB;
    g.sq = sqrt((double)Aa) + sqrt((double)Ab);
}

// compiler fence: the packed window is redefined (as far as the compiler can tell) at the top of every
// group iteration, so the two group bodies cannot be hoisted out of the loop or merged (their combined
// live ranges would not fit the register budget)
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

template <int WIN, int SH>
__global__ __launch_bounds__(256) void k4_glcm_thread(const uint8_t *__restrict__ q, int H, int W, int step, int oh, int ow,
                                                      glcm_out out, glcm_consts gc)
{
    __shared__ long long hq[SH == 3 ? 1024 : 256];
    if constexpr (SH == 3) {
        for (int i = threadIdx.x; i < 1024; i += 256) hq[i] = g_glcm_hq2[i];
    } else {
        hq[threadIdx.x] = c_glcm_hq[threadIdx.x];
    }
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
                if constexpr (c < 4) lo |= b << (8 * c + SH);  // pre-scaled by 2^SH (see glcm_group_stats)
                else hi |= b << (8 * (c - 4) + SH);
            });
            w[r][0] = lo;
            w[r][1] = hi;
        });
    }
    glcm_raw q0, q1;
#pragma nounroll
    for (int g = 0; g < 2; g++) {
        opaque_window<WIN>(w);
        if (g == 0) glcm_group_stats<WIN, 0, SH>(w, hq, q0);
        else glcm_group_stats<WIN, 1, SH>(w, hq, q1);
    }
    // M1 / M2 of the four angles (computed after the groups: nothing of it has to stay live across them), then the
    // 2^SH pre-scaling is undone (exact: every term is a multiple)
    opaque_window<WIN>(w);
    unsigned m1[4], m2[4];
    window_m1m2<WIN>(w, m1, m2);
    constexpr long long NA = (long long)WIN * (WIN - 1), NB = (long long)(WIN - 1) * (WIN - 1);
    const long long xy0 = q0.XYa >> (2 * SH), xy90 = q0.XYb >> (2 * SH), xy45 = q1.XYa >> (2 * SH), xy135 = q1.XYb >> (2 * SH);
    const long long M20 = m2[0] >> (2 * SH), M245 = m2[1] >> (2 * SH), M290 = m2[2] >> (2 * SH), M2135 = m2[3] >> (2 * SH);
    glcm_group g0, g1;
    g0.S1 = q0.S1 >> SH;
    g0.S2 = (M20 - 2ll * xy0) + (M290 - 2ll * xy90);
    g0.Hq = q0.Hq;
    g0.sq = q0.sq;
    g1.S1 = q1.S1 >> SH;
    g1.S2 = (M245 - 2ll * xy45) + (M2135 - 2ll * xy135);
    g1.Hq = q1.Hq;
    g1.sq = q1.sq;
    const double r0 = glcm_corr(NA, m1[0] >> SH, M20, 2ll * xy0), r1 = glcm_corr(NB, m1[1] >> SH, M245, 2ll * xy45);
    const double r2 = glcm_corr(NA, m1[2] >> SH, M290, 2ll * xy90), r3 = glcm_corr(NB, m1[3] >> SH, M2135, 2ll * xy135);
    glcm_finish(g0, g1, NA, NB, r0, r1, r2, r3, (size_t)oy * ow + ox, out, gc);
}


// ------------------------------------------------------------------------------------------------
// k4_glcm_pair — the dense case (window 7, step 1, levels <= 32) with TWO horizontally adjacent windows per thread.
// Windows x and x+1 share six of their seven columns: 35 of the 42 pair keys of the 0-degree angle, 36 of 42 (90),
// 30 of 36 (45 and 135).  The shared keys are built and sorted ONCE; each window then sorts its own 7 (6) keys and
// merges them into the shared run with Batcher's odd-even (m, n)-merging network (Knuth 5.3.4) before the same
// run-length pass as k4_glcm_thread:   per window 404 compare-exchanges instead of 550, 46 packed keys built instead
// of 78, 46 Hq table reads instead of 78.  The 7 x 8 patch of both windows fits the 14 registers the 7 x 7 window
// already took.  Where an angle has one key fewer than its register partner (35 / 36, 6 / 7) the free half holds a
// pad: an even value above every real key (no diagonal flag, equal to nothing), built from a pixel paired with
// itself so that its table read adds exactly Hq(0) = 2^52, which is subtracted again.
// Same integer statistics, same float64 finish: bit-identical to k4_glcm_thread (and to oracle.c mode 1).
// ------------------------------------------------------------------------------------------------
template <int M, int N> struct merge_net {
    int a[(M + N) * 8], b[(M + N) * 8];
    int n;
    int order[M + N];  // register indices in ascending order of their values after the network
};
struct merge_emit {
    int *a, *b, *n;
};
// merges the sorted runs held in registers x[0..m) and y[0..n): comparators appended to e, ascending order to out
constexpr void oem_build(const int *x, int m, const int *y, int n, int *out, merge_emit e)
{
    if (m == 0) { for (int i = 0; i < n; i++) out[i] = y[i]; return; }
    if (n == 0) { for (int i = 0; i < m; i++) out[i] = x[i]; return; }
    if (m == 1 && n == 1) {
        e.a[*e.n] = x[0]; e.b[*e.n] = y[0]; (*e.n)++;
        out[0] = x[0]; out[1] = y[0];
        return;
    }
    int xe[64] = {}, xo[64] = {}, ye[64] = {}, yo[64] = {}, v[128] = {}, w[128] = {};
    int me = 0, mo = 0, ne = 0, no = 0;
    for (int i = 0; i < m; i++) { if (i & 1) xo[mo++] = x[i]; else xe[me++] = x[i]; }
    for (int i = 0; i < n; i++) { if (i & 1) yo[no++] = y[i]; else ye[ne++] = y[i]; }
    oem_build(xe, me, ye, ne, v, e);
    oem_build(xo, mo, yo, no, w, e);
    const int lv = me + ne, lw = mo + no;
    int k = 0;
    out[k++] = v[0];
    for (int i = 0; i < lw; i++) {
        if (i + 1 < lv) {
            e.a[*e.n] = w[i]; e.b[*e.n] = v[i + 1]; (*e.n)++;
            out[k++] = w[i];
            out[k++] = v[i + 1];
        } else {
            out[k++] = w[i];
        }
    }
    for (int i = lw + 1; i < lv; i++) out[k++] = v[i];
}
template <int M, int N> constexpr merge_net<M, N> make_merge_net()
{
    merge_net<M, N> s{};
    int x[M] = {}, y[N] = {};
    for (int i = 0; i < M; i++) x[i] = i;
    for (int i = 0; i < N; i++) y[i] = M + i;
    int n = 0;
    oem_build(x, M, y, N, s.order, merge_emit{s.a, s.b, &n});
    s.n = n;
    return s;
}
// zero-one principle restricted to merging: every pair of sorted 0/1 runs must come out sorted
template <int M, int N> constexpr bool merge_net_ok(const merge_net<M, N> &s)
{
    for (int za = 0; za <= M; za++)
        for (int zb = 0; zb <= N; zb++) {
            int r[M + N] = {};
            for (int i = 0; i < M; i++) r[i] = i >= za;
            for (int i = 0; i < N; i++) r[M + i] = i >= zb;
            for (int c = 0; c < s.n; c++) {
                const int lo = r[s.a[c]] < r[s.b[c]] ? r[s.a[c]] : r[s.b[c]], hi = r[s.a[c]] + r[s.b[c]] - lo;
                r[s.a[c]] = lo;
                r[s.b[c]] = hi;
            }
            for (int i = 1; i < M + N; i++)
                if (r[s.order[i - 1]] > r[s.order[i]]) return false;
        }
    return true;
}
template <int M, int N> struct merge_holder {
    static constexpr merge_net<M, N> net = make_merge_net<M, N>();
    static_assert(merge_net_ok(net), "odd-even merging network does not merge");
};

#define GP_PAD_LO 0x0000fffeu
#define GP_PAD_HI 0xfffc0000u

// pixel positions (patch row, patch column 0..7) of entry p of a key set: SET 0 = shared by both windows,
// 1 = only window A (patch columns 0..6), 2 = only window B (columns 1..7).  half 0 = low 16 bits (angle 0 / 45 degrees),
// half 1 = high 16 bits (90 / 135).  pad: the entry does not exist for this half.
struct gp_pos {
    int rx, cx, ry, cy;
    bool pad;
};
template <int G, int SET> __host__ __device__ constexpr int gp_count() { return G == 0 ? (SET == 0 ? 36 : 7) : (SET == 0 ? 30 : 6); }
template <int G, int SET> __host__ __device__ constexpr gp_pos gp_where(int p, int half)
{
    if (G == 0) {
        if (half == 0) {  // 0 degrees: (r, c)-(r, c+1); pair columns 0 | 1..5 | 6
            if (SET == 0) return p < 35 ? gp_pos{p / 5, 1 + p % 5, p / 5, 2 + p % 5, false} : gp_pos{0, 0, 0, 0, true};
            return SET == 1 ? gp_pos{p, 0, p, 1, false} : gp_pos{p, 6, p, 7, false};
        }
        // 90 degrees: (r, c)-(r+1, c); columns 0 | 1..6 | 7
        if (SET == 0) return gp_pos{p / 6, 1 + p % 6, p / 6 + 1, 1 + p % 6, false};
        if (p >= 6) return gp_pos{0, 0, 0, 0, true};
        return SET == 1 ? gp_pos{p, 0, p + 1, 0, false} : gp_pos{p, 7, p + 1, 7, false};
    }
    if (half == 0) {  // 45 degrees: (r, c)-(r+1, c+1); first columns 0 | 1..5 | 6
        if (SET == 0) return gp_pos{p / 5, 1 + p % 5, p / 5 + 1, 2 + p % 5, false};
        return SET == 1 ? gp_pos{p, 0, p + 1, 1, false} : gp_pos{p, 6, p + 1, 7, false};
    }
    // 135 degrees: (r, c)-(r+1, c-1); first columns 1 | 2..6 | 7
    if (SET == 0) return gp_pos{p / 5, 2 + p % 5, p / 5 + 1, 1 + p % 5, false};
    return SET == 1 ? gp_pos{p, 1, p + 1, 0, false} : gp_pos{p, 7, p + 1, 6, false};
}
template <int G, int SET> __host__ __device__ constexpr int gp_pads()
{
    int n = 0;
    for (int p = 0; p < gp_count<G, SET>(); p++) n += (gp_where<G, SET>(p, 0).pad ? 1 : 0) + (gp_where<G, SET>(p, 1).pad ? 1 : 0);
    return n;
}

// builds the packed keys of one set into K[OFF ..) and returns the sum of their Hq table reads (pads included)
template <int G, int SET, int OFF, int NK>
__device__ __forceinline__ long long gp_build(const unsigned (&P)[8][2], const long long *__restrict__ hq, unsigned (&K)[NK])
{
    long long Hq = 0;
    static_for<gp_count<G, SET>()>([&](auto I) {
        constexpr int p = I;
        constexpr gp_pos A = gp_where<G, SET>(p, 0), B = gp_where<G, SET>(p, 1);
        constexpr unsigned selx = (unsigned)(A.cx & 3) | (0x0cu << 8) | ((unsigned)(4 + (B.cx & 3)) << 16) | (0x0cu << 24);
        constexpr unsigned sely = (unsigned)(A.cy & 3) | (0x0cu << 8) | ((unsigned)(4 + (B.cy & 3)) << 16) | (0x0cu << 24);
        const unsigned x = __builtin_amdgcn_perm(P[B.rx][B.cx >> 2], P[A.rx][A.cx >> 2], selx);
        const unsigned y = __builtin_amdgcn_perm(P[B.ry][B.cy >> 2], P[A.ry][A.cy >> 2], sely);
        const unsigned lo = pk_min(x, y), hi = pk_max(x, y);
        const unsigned d = pk_sub(hi, lo);
        const unsigned one = 0x00010001u;
        const unsigned diag = pk_sub_sat(one, d);
        unsigned k = ((lo << 8) | hi) | diag;
        if constexpr (A.pad) k = (k & 0xffff0000u) | GP_PAD_LO;
        if constexpr (B.pad) k = (k & 0x0000ffffu) | GP_PAD_HI;
        K[OFF + p] = k;
        pin32(K[OFF + p]);
        const unsigned off = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, d), (us2){1, 32}, 0u, false);
        Hq += *reinterpret_cast<const long long *>(reinterpret_cast<const char *>(hq) + off);
        if constexpr (p % 6 == 5) pin64(Hq);
    });
    return Hq;
}

// E2 and D (both halves packed) of the keys in K taken in the order ORD
template <int NK, typename NET> __device__ __forceinline__ void gp_runlength(const unsigned (&K)[NK], unsigned &E2, unsigned &D)
{
    const unsigned one = 0x00010001u;
    E2 = 0;
    unsigned t = 0;
    D = K[NET::net.order[0]] & one;
    static_for<NK - 1>([&](auto I) {
        constexpr int i = NET::net.order[I + 1], j = NET::net.order[I];
        const unsigned diag = K[i] & one;
        const unsigned eq = pk_sub_sat(one, K[i] ^ K[j]);   // 1 where the keys are equal, 0 where they differ
        t = pk_mul(t + one + diag, eq);
        E2 += t;
        D += diag;
    });
}

template <int NS> __device__ __forceinline__ void opaque_patch(unsigned (&P)[8][2])
{
#if defined(__HIP_DEVICE_COMPILE__)
    static_for<NS>([&](auto I) {
        unsigned &a = P[I][0];
        unsigned &b = P[I][1];
        asm volatile("" : "+v"(a), "+v"(b));
    });
#endif
}

// one angle group of both windows: E2 / D per window from the shared sorted run, Hq per window
template <int G>
__device__ __forceinline__ void gp_group(const unsigned (&P)[8][2], const long long *__restrict__ hq, long long &HqA, long long &HqB,
                                         double &sqA, double &sqB)
{
    constexpr int NS = gp_count<G, 0>(), NO = gp_count<G, 1>(), NK = NS + NO;
    constexpr int PAIRS = G == 0 ? 42 : 36;
    using MNET = merge_holder<NS, NO>;
    unsigned KA[NK], KB[NK];
    const long long HqS = gp_build<G, 0, 0, NK>(P, hq, KA);
    __builtin_amdgcn_sched_barrier(0);
    static_for<net_holder<NS>::net.n>([&](auto I) {
        constexpr int ia = net_holder<NS>::net.a[I], ib = net_holder<NS>::net.b[I];
        const unsigned ka = KA[ia], kb = KA[ib];
        KA[ia] = pk_min(ka, kb);
        KA[ib] = pk_max(ka, kb);
    });
    static_for<NS>([&](auto I) { KB[I] = KA[I]; });
    __builtin_amdgcn_sched_barrier(0);
    auto window = [&](auto set_t, unsigned (&K)[NK], long long &Hq, double &sq) {
        constexpr int SET = decltype(set_t)::value;
        Hq = HqS + gp_build<G, SET, NS, NK>(P, hq, K) - (long long)(gp_pads<G, 0>() + gp_pads<G, SET>()) * 4503599627370496ll;
        static_for<net_holder<NO>::net.n>([&](auto I) {
            constexpr int ia = NS + net_holder<NO>::net.a[I], ib = NS + net_holder<NO>::net.b[I];
            const unsigned ka = K[ia], kb = K[ib];
            K[ia] = pk_min(ka, kb);
            K[ib] = pk_max(ka, kb);
        });
        static_for<MNET::net.n>([&](auto I) {
            constexpr int ia = MNET::net.a[I], ib = MNET::net.b[I];
            const unsigned ka = K[ia], kb = K[ib];
            K[ia] = pk_min(ka, kb);
            K[ib] = pk_max(ka, kb);
        });
        unsigned E2, D;
        gp_runlength<NK, MNET>(K, E2, D);
        const long long Aa = 2ll * (PAIRS + (int)(D & 0xffffu)) + 4ll * (long long)(E2 & 0xffffu);
        const long long Ab = 2ll * (PAIRS + (int)(D >> 16)) + 4ll * (long long)(E2 >> 16);
        sq = sqrt((double)Aa) + sqrt((double)Ab);
    };
    window(std::integral_constant<int, 1>{}, KA, HqA, sqA);
    __builtin_amdgcn_sched_barrier(0);
    window(std::integral_constant<int, 2>{}, KB, HqB, sqB);
}

// everything of one window that does not involve the key sort: pair moments, M1 / M2, the float64 finish
__device__ __forceinline__ void gp_finish(unsigned (&w)[8][2], long long Hq0, double sq0, long long Hq1, double sq1, size_t o,
                                          const glcm_out &out, const glcm_consts &gc)
{
    constexpr int WIN = 7, SH = 3;
    unsigned S1a, XYa, S1b, XYb, S1c, XYc, S1d, XYd;
    row_moments<WIN, 0, 1>(w, S1a, XYa);
    row_moments<WIN, 1, 0>(w, S1b, XYb);
    row_moments<WIN, 1, 1>(w, S1c, XYc);
    row_moments<WIN, 1, -1>(w, S1d, XYd);
    unsigned m1[4], m2[4];
    window_m1m2<WIN>(w, m1, m2);
    constexpr long long NA = (long long)WIN * (WIN - 1), NB = (long long)(WIN - 1) * (WIN - 1);
    const long long xy0 = XYa >> (2 * SH), xy90 = XYb >> (2 * SH), xy45 = XYc >> (2 * SH), xy135 = XYd >> (2 * SH);
    const long long M20 = m2[0] >> (2 * SH), M245 = m2[1] >> (2 * SH), M290 = m2[2] >> (2 * SH), M2135 = m2[3] >> (2 * SH);
    glcm_group g0, g1;
    g0.S1 = (S1a + S1b) >> SH;
    g0.S2 = (M20 - 2ll * xy0) + (M290 - 2ll * xy90);
    g0.Hq = Hq0;
    g0.sq = sq0;
    g1.S1 = (S1c + S1d) >> SH;
    g1.S2 = (M245 - 2ll * xy45) + (M2135 - 2ll * xy135);
    g1.Hq = Hq1;
    g1.sq = sq1;
    const double r0 = glcm_corr(NA, m1[0] >> SH, M20, 2ll * xy0), r1 = glcm_corr(NB, m1[1] >> SH, M245, 2ll * xy45);
    const double r2 = glcm_corr(NA, m1[2] >> SH, M290, 2ll * xy90), r3 = glcm_corr(NB, m1[3] >> SH, M2135, 2ll * xy135);
    glcm_finish(g0, g1, NA, NB, r0, r1, r2, r3, o, out, gc);
}

__global__ __launch_bounds__(256) void k4_glcm_pair(const uint8_t *__restrict__ q, int H, int W, int oh, int ow, glcm_out out,
                                                    glcm_consts gc)
{
    constexpr int SH = 3;
    __shared__ long long hq[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) hq[i] = g_glcm_hq2[i];
    __syncthreads();
    const int ox = 2 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= ow || oy >= oh) return;
    const bool hasB = ox + 1 < ow;  // an odd map width leaves the last thread of a row with one window
    unsigned P[8][2];
    {
        const uint8_t *wp = q + (size_t)oy * W + (size_t)ox;
        static_for<7>([&](auto I) {
            constexpr int r = I;
            unsigned lo = 0, hi = 0;
            static_for<8>([&](auto J) {
                constexpr int c = J;
                const unsigned b = (c < 7 || hasB) ? wp[(size_t)r * W + c] : 0u;
                if constexpr (c < 4) lo |= b << (8 * c + SH);
                else hi |= b << (8 * (c - 4) + SH);
            });
            P[r][0] = lo;
            P[r][1] = hi;
        });
        P[7][0] = P[7][1] = 0;
    }
    long long Hq[2][2];
    double sq[2][2];
#pragma nounroll
    for (int g = 0; g < 2; g++) {
        opaque_patch<7>(P);
        if (g == 0) gp_group<0>(P, hq, Hq[0][0], Hq[1][0], sq[0][0], sq[1][0]);
        else gp_group<1>(P, hq, Hq[0][1], Hq[1][1], sq[0][1], sq[1][1]);
    }
    const size_t o = (size_t)oy * ow + ox;
    unsigned w[8][2];
    opaque_patch<7>(P);
    static_for<7>([&](auto I) {
        w[I][0] = P[I][0];
        w[I][1] = P[I][1] & 0x00ffffffu;
    });
    w[7][0] = w[7][1] = 0;
    gp_finish(w, Hq[0][0], sq[0][0], Hq[0][1], sq[0][1], o, out, gc);
    if (hasB) {
        opaque_patch<7>(P);
        static_for<7>([&](auto I) {
            w[I][0] = __builtin_amdgcn_alignbyte(P[I][1], P[I][0], 1);
            w[I][1] = P[I][1] >> 8;
        });
        gp_finish(w, Hq[1][0], sq[1][0], Hq[1][1], sq[1][1], o + 1, out, gc);
    }
}

// ------------------------------------------------------------------------------------------------
// k4_glcm_quad (r04) — the dense case with a 2 x 2 BLOCK of windows per thread: A = (x, y), B = (x+1, y), C = (x, y+1),
// D = (x+1, y+1) on one 8 x 8 patch (16 registers).  The four windows share, per angle, a CORE of pairs (30 of the 42
// pairs of the 0 / 90 degree angles, 25 of the 36 of 45 / 135), two of them a STRIP of 5 more, and each adds its OWN 7
// (6): the core is built and sorted once per thread, core + strip merged once per two windows, and a window only sorts
// its own keys and merges them in — per window 375 compare-exchanges instead of the pair kernel's 446 and 32 packed keys
// built instead of 46.  The same integer statistics, the same float64 finish (gp_finish): bit-identical.
//   Packing.  A register carries two angles (low / high half).  For 45 / 135 degrees both halves have the same geometry.
//   For 0 / 90 degrees the geometry of one is the transpose of the other (0: row strips of 5, column strips of 7;
//   90: column strips of 5, row strips of 7), so a register "slot" carries the 0-degree keys of one window and the
//   90-degree keys of its TRANSPOSE partner (A|A, B|C, C|B, D|D): then the merged core + strip array of the top row strip
//   (0 degrees: A, B) and of the left column strip (90 degrees: A, C) is what slots 0 and 1 both start from, and the
//   bottom / right one what slots 2 and 3 start from.  The Hq sums of the sets that are split between windows are read
//   per half from the 32-entry table instead of the table of pair sums.
// ------------------------------------------------------------------------------------------------
// two-stage merging network on registers [0, NC + NT + NO): stage 1 merges the sorted runs [0, NC) and [NC, NC + NT)
// (the logical order afterwards is order1), stage 2 merges that run with the sorted run [NC + NT, NC + NT + NO)
template <int NC, int NT, int NO> struct quad_net {
    int a1[(NC + NT) * 8], b1[(NC + NT) * 8], n1;
    int order1[NC + NT];
    int a2[(NC + NT + NO) * 8], b2[(NC + NT + NO) * 8], n2;
    int order2[NC + NT + NO];
};
template <int NC, int NT, int NO> constexpr quad_net<NC, NT, NO> make_quad_net()
{
    quad_net<NC, NT, NO> s{};
    int x[NC] = {}, y[NT] = {}, z[NO] = {};
    for (int i = 0; i < NC; i++) x[i] = i;
    for (int i = 0; i < NT; i++) y[i] = NC + i;
    for (int i = 0; i < NO; i++) z[i] = NC + NT + i;
    int n = 0;
    oem_build(x, NC, y, NT, s.order1, merge_emit{s.a1, s.b1, &n});
    s.n1 = n;
    n = 0;
    oem_build(s.order1, NC + NT, z, NO, s.order2, merge_emit{s.a2, s.b2, &n});
    s.n2 = n;
    return s;
}
// zero-one principle, stage by stage (stage 2 starts from any sorted 0 / 1 run laid out in stage 1's order)
template <int NC, int NT, int NO> constexpr bool quad_net_ok(const quad_net<NC, NT, NO> &s)
{
    for (int za = 0; za <= NC; za++)
        for (int zb = 0; zb <= NT; zb++) {
            int r[NC + NT] = {};
            for (int i = 0; i < NC; i++) r[i] = i >= za;
            for (int i = 0; i < NT; i++) r[NC + i] = i >= zb;
            for (int c = 0; c < s.n1; c++) {
                const int lo = r[s.a1[c]] < r[s.b1[c]] ? r[s.a1[c]] : r[s.b1[c]], hi = r[s.a1[c]] + r[s.b1[c]] - lo;
                r[s.a1[c]] = lo;
                r[s.b1[c]] = hi;
            }
            for (int i = 1; i < NC + NT; i++)
                if (r[s.order1[i - 1]] > r[s.order1[i]]) return false;
        }
    for (int za = 0; za <= NC + NT; za++)
        for (int zc = 0; zc <= NO; zc++) {
            int r[NC + NT + NO] = {};
            for (int i = 0; i < NC + NT; i++) r[s.order1[i]] = i >= za;
            for (int i = 0; i < NO; i++) r[NC + NT + i] = i >= zc;
            for (int c = 0; c < s.n2; c++) {
                const int lo = r[s.a2[c]] < r[s.b2[c]] ? r[s.a2[c]] : r[s.b2[c]], hi = r[s.a2[c]] + r[s.b2[c]] - lo;
                r[s.a2[c]] = lo;
                r[s.b2[c]] = hi;
            }
            for (int i = 1; i < NC + NT + NO; i++)
                if (r[s.order2[i - 1]] > r[s.order2[i]]) return false;
        }
    return true;
}
template <int NC, int NT, int NO> struct quad_holder {
    static constexpr quad_net<NC, NT, NO> net = make_quad_net<NC, NT, NO>();
    static_assert(quad_net_ok(net), "two-stage merging network does not merge");
};

// sets of a group: 0 = core, 1 = strip of slots 0 / 1 (0 degrees: top row; 90: left column; 45 / 135: top row),
// 2 = strip of slots 2 / 3 (bottom row; right column), 3 + s = own keys of slot s
template <int G, int SET> __host__ __device__ constexpr int gq_count()
{
    return G == 0 ? (SET == 0 ? 30 : (SET <= 2 ? 5 : 7)) : (SET == 0 ? 25 : (SET <= 2 ? 5 : 6));
}
template <int G, int SET> __host__ __device__ constexpr gp_pos gq_where(int p, int half)
{
    if (G == 0 && half == 0) {   // 0 degrees: (r, c)-(r, c + 1), pair column c in 0..6
        if (SET == 0) return gp_pos{1 + p / 5, 1 + p % 5, 1 + p / 5, 2 + p % 5, false};
        if (SET == 1) return gp_pos{0, 1 + p, 0, 2 + p, false};            // top row strip (A, B)
        if (SET == 2) return gp_pos{7, 1 + p, 7, 2 + p, false};            // bottom row strip (C, D)
        if (SET == 3) return gp_pos{p, 0, p, 1, false};                    // slot 0: A's left column
        if (SET == 4) return gp_pos{p, 6, p, 7, false};                    // slot 1: B's right column
        if (SET == 5) return gp_pos{1 + p, 0, 1 + p, 1, false};            // slot 2: C's left column
        return gp_pos{1 + p, 6, 1 + p, 7, false};                          // slot 3: D's right column
    }
    if (G == 0) {                // 90 degrees: (r, c)-(r + 1, c), pair row r in 0..6
        if (SET == 0) return gp_pos{1 + p / 6, 1 + p % 6, 2 + p / 6, 1 + p % 6, false};
        if (SET == 1) return gp_pos{1 + p, 0, 2 + p, 0, false};            // left column strip (A, C)
        if (SET == 2) return gp_pos{1 + p, 7, 2 + p, 7, false};            // right column strip (B, D)
        if (SET == 3) return gp_pos{0, p, 1, p, false};                    // slot 0: A's top row
        if (SET == 4) return gp_pos{6, p, 7, p, false};                    // slot 1: C's bottom row
        if (SET == 5) return gp_pos{0, 1 + p, 1, 1 + p, false};            // slot 2: B's top row
        return gp_pos{6, 1 + p, 7, 1 + p, false};                          // slot 3: D's bottom row
    }
    // 45 degrees: (r, c)-(r + 1, c + 1); 135 degrees: (r, c + 1)-(r + 1, c); pair row r and pair column c in 0..6
    const int sh = half == 0 ? 0 : 1;
    int r = 0, c = 0;
    if (SET == 0) { r = 1 + p / 5; c = 1 + p % 5; }
    else if (SET == 1) { r = 0; c = 1 + p; }
    else if (SET == 2) { r = 6; c = 1 + p; }
    else if (SET == 3) { r = p; c = 0; }
    else if (SET == 4) { r = p; c = 6; }
    else if (SET == 5) { r = 1 + p; c = 0; }
    else { r = 1 + p; c = 6; }
    return gp_pos{r, c + sh, r + 1, c + 1 - sh, false};
}

// builds the packed keys of one set into K[OFF ..).  SPLIT: the Hq table reads are made per half (HqLo / HqHi: the two
// halves end up in different windows); otherwise one read of the pair-sum table (added to HqLo)
template <int G, int SET, int OFF, int NK, bool SPLIT>
__device__ __forceinline__ void gq_build(const unsigned (&P)[8][2], const long long *__restrict__ hq2, const long long *__restrict__ hq1,
                                         unsigned (&K)[NK], long long &HqLo, long long &HqHi)
{
    static_for<gq_count<G, SET>()>([&](auto I) {
        constexpr int p = I;
        constexpr gp_pos A = gq_where<G, SET>(p, 0), B = gq_where<G, SET>(p, 1);
        constexpr unsigned selx = (unsigned)(A.cx & 3) | (0x0cu << 8) | ((unsigned)(4 + (B.cx & 3)) << 16) | (0x0cu << 24);
        constexpr unsigned sely = (unsigned)(A.cy & 3) | (0x0cu << 8) | ((unsigned)(4 + (B.cy & 3)) << 16) | (0x0cu << 24);
        const unsigned x = __builtin_amdgcn_perm(P[B.rx][B.cx >> 2], P[A.rx][A.cx >> 2], selx);
        const unsigned y = __builtin_amdgcn_perm(P[B.ry][B.cy >> 2], P[A.ry][A.cy >> 2], sely);
        const unsigned lo = pk_min(x, y), hi = pk_max(x, y);
        const unsigned d = pk_sub(hi, lo);
        const unsigned one = 0x00010001u;
        const unsigned diag = pk_sub_sat(one, d);
        K[OFF + p] = ((lo << 8) | hi) | diag;
        pin32(K[OFF + p]);
        if constexpr (SPLIT) {   // d = 8 * dA | (8 * dB) << 16: byte offsets into the 32-entry table
            HqLo += *reinterpret_cast<const long long *>(reinterpret_cast<const char *>(hq1) + (d & 0xffffu));
            HqHi += *reinterpret_cast<const long long *>(reinterpret_cast<const char *>(hq1) + (d >> 16));
        } else {
            const unsigned off = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, d), (us2){1, 32}, 0u, false);
            HqLo += *reinterpret_cast<const long long *>(reinterpret_cast<const char *>(hq2) + off);
        }
        // table reads in flight: 6 pair-sum reads (core, built while few keys are live) or 2 x 2 per-half reads (8 registers:
        // the own-key sets are built at the kernel's register peak)
        if constexpr (SPLIT ? p % 2 == 1 : p % 6 == 5) { pin64(HqLo); pin64(HqHi); }
    });
}

// one angle group of the four windows.  ED[s] = D + 2 E2 of slot s (both halves packed: at most 42 + 2 * 1722 per half, so
// A = 2 (np + D) + 4 E2 = 2 (np + ED)); Hq[w] of window w (A, B, C, D)
template <int G>
__device__ __forceinline__ void gq_group(const unsigned (&P)[8][2], const long long *__restrict__ hq2, const long long *__restrict__ hq1,
                                         unsigned (&ED)[4], long long (&Hq)[4])
{
    constexpr int NC = gq_count<G, 0>(), NT = gq_count<G, 1>(), NO = gq_count<G, 3>(), NS = NC + NT, NK = NS + NO;
    using QN = quad_holder<NC, NT, NO>;
    unsigned KC[NK], KS[NK], K[NK];    // KC: the sorted core (later core + strip 2, then slot 3 in place); KS: core + strip 1, then slot 1 in place
    long long hqCore = 0, dummy = 0;
    gq_build<G, 0, 0, NK, false>(P, hq2, hq1, KC, hqCore, dummy);
    __builtin_amdgcn_sched_barrier(0);
    static_for<net_holder<NC>::net.n>([&](auto I) {
        constexpr int ia = net_holder<NC>::net.a[I], ib = net_holder<NC>::net.b[I];
        const unsigned ka = KC[ia], kb = KC[ib];
        KC[ia] = pk_min(ka, kb);
        KC[ib] = pk_max(ka, kb);
    });
    __builtin_amdgcn_sched_barrier(0);
    // Hq of a window = core + its strip halves + its own halves.  Group 0 packs the 0-degree keys of window w with the
    // 90-degree keys of its transpose partner: the low half of slot s belongs to window s, the high half to window
    // part(s) = A, C, B, D; the low half of strip HB to windows 2 HB, 2 HB + 1 (a row strip), the high half to the windows of
    // column HB (A, C / B, D).  Group 1: both halves of everything belong to the slot's own window / row of windows.
    Hq[0] = Hq[1] = Hq[2] = Hq[3] = hqCore;
    // core + strip HB, merged, in S (whose first NC registers hold the sorted core)
    auto add_strip = [&](auto hb_t, unsigned (&S)[NK]) {
        constexpr int HB = decltype(hb_t)::value;
        long long lo = 0, hi = 0;
        gq_build<G, 1 + HB, NC, NK, true>(P, hq2, hq1, S, lo, hi);
        Hq[2 * HB] += lo;
        Hq[2 * HB + 1] += lo;
        if constexpr (G == 0) { Hq[HB] += hi; Hq[HB + 2] += hi; }
        else { Hq[2 * HB] += hi; Hq[2 * HB + 1] += hi; }
        static_for<net_holder<NT>::net.n>([&](auto I) {
            constexpr int ia = NC + net_holder<NT>::net.a[I], ib = NC + net_holder<NT>::net.b[I];
            const unsigned ka = S[ia], kb = S[ib];
            S[ia] = pk_min(ka, kb);
            S[ib] = pk_max(ka, kb);
        });
        static_for<QN::net.n1>([&](auto I) {
            constexpr int ia = QN::net.a1[I], ib = QN::net.b1[I];
            const unsigned ka = S[ia], kb = S[ib];
            S[ia] = pk_min(ka, kb);
            S[ib] = pk_max(ka, kb);
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    // slot S on the array X whose first NS registers hold core + strip (logical order: order1)
    auto slot = [&](auto s_t, unsigned (&X)[NK]) {
        constexpr int S = decltype(s_t)::value;
        {
            long long lo = 0, hi = 0;
            gq_build<G, 3 + S, NS, NK, true>(P, hq2, hq1, X, lo, hi);
            constexpr int PART = G == 0 ? (S == 1 ? 2 : (S == 2 ? 1 : S)) : S;
            Hq[S] += lo;
            Hq[PART] += hi;
        }
        static_for<net_holder<NO>::net.n>([&](auto I) {
            constexpr int ia = NS + net_holder<NO>::net.a[I], ib = NS + net_holder<NO>::net.b[I];
            const unsigned ka = X[ia], kb = X[ib];
            X[ia] = pk_min(ka, kb);
            X[ib] = pk_max(ka, kb);
        });
        static_for<QN::net.n2>([&](auto I) {
            constexpr int ia = QN::net.a2[I], ib = QN::net.b2[I];
            const unsigned ka = X[ia], kb = X[ib];
            X[ia] = pk_min(ka, kb);
            X[ib] = pk_max(ka, kb);
        });
        const unsigned one = 0x00010001u;
        unsigned E2 = 0, t = 0, D = X[QN::net.order2[0]] & one;
        static_for<NK - 1>([&](auto I) {
            constexpr int i = QN::net.order2[I + 1], j = QN::net.order2[I];
            const unsigned diag = X[i] & one;
            const unsigned eq = pk_sub_sat(one, X[i] ^ X[j]);
            t = pk_mul(t + one + diag, eq);
            E2 += t;
            D += diag;
        });
        ED[S] = D + (E2 << 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // slots 0, 1 from core + strip 1 (a copy of the core: the core itself is needed again); the last user of an array works in place
    static_for<NC>([&](auto I) { KS[I] = KC[I]; });
    add_strip(std::integral_constant<int, 0>{}, KS);
    static_for<NS>([&](auto I) { K[I] = KS[I]; });
    slot(std::integral_constant<int, 0>{}, K);
    slot(std::integral_constant<int, 1>{}, KS);
    add_strip(std::integral_constant<int, 1>{}, KC);
    static_for<NS>([&](auto I) { K[I] = KC[I]; });
    slot(std::integral_constant<int, 2>{}, K);
    slot(std::integral_constant<int, 3>{}, KC);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k4_glcm_quad(const uint8_t *__restrict__ q, int H, int W, int oh, int ow, glcm_out out,
                                                    glcm_consts gc)
{
    constexpr int SH = 3;
    __shared__ long long hq2[1024];
    __shared__ long long hq1[32];
    for (int i = threadIdx.x; i < 1024; i += 256) hq2[i] = g_glcm_hq2[i];
    if (threadIdx.x < 32) hq1[threadIdx.x] = c_glcm_hq[threadIdx.x];
    __syncthreads();
    // The window coordinates are needed at the two ends of the kernel only.  Kept in vector registers they (or the thread id
    // they come from) were spilled to scratch memory at the 168-register budget: 3 dwords per thread, +2 B/px of HBM writes in
    // the PMC table.  So they are DERIVED twice from values that cost no vector register in between: the wave's index in
    // the workgroup (uniform: a scalar register) and the lane index (v_mbcnt).
    int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
    auto coords = [&](int &ox_, int &oy_) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(wave_s));        // opaque: not merged with the other derivation
#endif
        unsigned zero = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(zero));
#endif
        const int lane_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        ox_ = 2 * ((int)blockIdx.x * 64 + lane_);
        oy_ = 2 * ((int)blockIdx.y * 4 + wave_s);
    };
    int ox, oy;
    coords(ox, oy);
    if (ox >= ow || oy >= oh) return;
    bool hasX = ox + 1 < ow, hasY = oy + 1 < oh;   // an odd map width / height leaves the last column / row of threads with fewer windows
    unsigned P[8][2];
    {
        const uint8_t *wp = q + (size_t)oy * W + (size_t)ox;
        static_for<8>([&](auto I) {
            constexpr int r = I;
            unsigned lo = 0, hi = 0;
            static_for<8>([&](auto J) {
                constexpr int c = J;
                const unsigned b = ((c < 7 || hasX) && (r < 7 || hasY)) ? wp[(size_t)r * W + c] : 0u;
                if constexpr (c < 4) lo |= b << (8 * c + SH);
                else hi |= b << (8 * (c - 4) + SH);
            });
            P[r][0] = lo;
            P[r][1] = hi;
        });
    }
    unsigned ED[2][4];
    long long Hq[2][4];
#pragma nounroll
    for (int g = 0; g < 2; g++) {
        opaque_patch<8>(P);
        if (g == 0) gq_group<0>(P, hq2, hq1, ED[0], Hq[0]);
        else gq_group<1>(P, hq2, hq1, ED[1], Hq[1]);
    }
    // A = 2 (np + D) + 4 E2 per window and angle; group 0: window w's 0-degree statistics sit in the low half of slot w,
    // its 90-degree statistics in the high half of its transpose partner's slot (A 0, B 2, C 1, D 3)
    coords(ox, oy);
    hasX = ox + 1 < ow;
    hasY = oy + 1 < oh;
    auto root_sum = [&](int pairs, unsigned ed_lo, unsigned ed_hi) {
        const long long Aa = 2ll * (pairs + (int)(ed_lo & 0xffffu));
        const long long Ab = 2ll * (pairs + (int)(ed_hi >> 16));
        return sqrt((double)Aa) + sqrt((double)Ab);
    };
    // RSSEG_GLCM_COUNT_UNROLL (profiles/valu_hist.sh only): the loop unrolled, so that the STATIC instruction histogram of the
    // code object equals the executed one (the shipped kernel keeps the loop rolled: one copy of the finish)
#ifdef RSSEG_GLCM_COUNT_UNROLL
#pragma unroll
#else
#pragma nounroll
#endif
    for (int wdw = 0; wdw < 4; wdw++) {
        const int dx = wdw & 1, dy = wdw >> 1;
        if ((dx && !hasX) || (dy && !hasY)) continue;
        const int part = wdw == 1 ? 2 : (wdw == 2 ? 1 : wdw);
        unsigned e0l = 0, e0h = 0, e1 = 0;
        long long h0 = 0, h1 = 0;
        static_for<4>([&](auto I) {     // constant indices into the register arrays
            constexpr int s = I;
            if (wdw == s) { e0l = ED[0][s]; e1 = ED[1][s]; h0 = Hq[0][s]; h1 = Hq[1][s]; }
            if (part == s) e0h = ED[0][s];
        });
        const double sq0 = root_sum(42, e0l, e0h), sq1 = root_sum(36, e1, e1);
        unsigned w[8][2];
        opaque_patch<8>(P);
        static_for<7>([&](auto I) {
            constexpr int r = I;
            unsigned a0 = 0, a1 = 0;
            static_for<2>([&](auto DY) {    // rows r + dy with a constant index
                if (dy == DY) { a0 = P[r + DY][0]; a1 = P[r + DY][1]; }
            });
            w[r][0] = dx ? __builtin_amdgcn_alignbyte(a1, a0, 1) : a0;
            w[r][1] = dx ? (a1 >> 8) : (a1 & 0x00ffffffu);
        });
        w[7][0] = w[7][1] = 0;
        gp_finish(w, h0, sq0, h1, sq1, (size_t)(oy + dy) * ow + (ox + dx), out, gc);
    }
}

// one workgroup per window; LDS histogram of ordered cells [levels][levels]
// One WAVE per window (levels <= 32, any window size < 256): the four angles' co-occurrence counts live in four private
// 2 KB LDS tables of packed 16-bit counters (a window has fewer than 65536 pairs); no workgroup barrier anywhere — a wave's
// LDS operations execute in order, so its own atomics are complete before its reads.  This is the reference's default
// geometry (window 21, step 21: indices.py:248) — k4_glcm_wg spent most of its 21 us per window in 20 barriers.
__global__ __launch_bounds__(256) void k4_glcm_wave(const uint8_t *__restrict__ q, int H, int W, int levels, int win, int step, int oh,
                                                    int ow, glcm_out out, glcm_consts gc)
{
    __shared__ unsigned hist_all[4][4 * 512];  // [wave][angle][x * 32 + y packed two counters per dword]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long widx = (long long)blockIdx.x * 4 + wave;
    if (widx >= (long long)oh * ow) return;
    const int oy = (int)(widx / ow), ox = (int)(widx - (long long)oy * ow);
    const uint8_t *wp = q + (size_t)(oy * step) * W + (size_t)ox * step;
    unsigned *hist = hist_all[wave];
#pragma unroll
    for (int t = 0; t < 8; t++) reinterpret_cast<uint4 *>(hist)[t * 64 + lane] = make_uint4(0, 0, 0, 0);
    long long sums[4][6];  // per angle: S1 S2 Hq M1 M2 Mx (this lane's share)
    for (int a = 0; a < 4; a++) {
        const int dr = a == 0 ? 0 : 1, dc = a == 0 ? 1 : (a == 1 ? 1 : (a == 2 ? 0 : -1));
        const int r1 = dr > 0 ? win - dr : win, c0 = dc < 0 ? -dc : 0, c1 = dc > 0 ? win - dc : win;
        const int pw = c1 - c0, P = r1 * pw;
        int s1 = 0, s2 = 0, m1 = 0, m2 = 0, mx = 0;
        long long hq = 0;
        for (int p = lane; p < P; p += 64) {
            const int r = p / pw, c = c0 + p - r * pw;
            int x = wp[(size_t)r * W + c], y = wp[(size_t)(r + dr) * W + (c + dc)];
            x = x > 31 ? 31 : x;   // the quantiser guarantees < levels; never index outside the table
            y = y > 31 ? 31 : y;
            const int bin = x * 32 + y;
            atomicAdd(&hist[a * 512 + (bin >> 1)], 1u << (16 * (bin & 1)));
            const int d = x > y ? x - y : y - x;
            s1 += d; s2 += d * d; hq += c_glcm_hq[d];
            m1 += x + y; m2 += x * x + y * y; mx += 2 * x * y;
        }
        sums[a][0] = s1; sums[a][1] = s2; sums[a][2] = hq; sums[a][3] = m1; sums[a][4] = m2; sums[a][5] = mx;
    }
    long long A[4];
    for (int a = 0; a < 4; a++) {
        const unsigned *h = hist + a * 512;
        long long acc = 0;
        for (int t = lane; t < 1024; t += 64) {
            const int x = t >> 5, y = t & 31, u = y * 32 + x;
            const long long g = (long long)((h[t >> 1] >> (16 * (t & 1))) & 0xffffu) + (long long)((h[u >> 1] >> (16 * (u & 1))) & 0xffffu);
            acc += g * g;
        }
        A[a] = wave_sum(acc);
#pragma unroll
        for (int t = 0; t < 6; t++) sums[a][t] = wave_sum(sums[a][t]);
    }
    if (lane == 0) {
        const long long na = (long long)win * (win - 1), nb = (long long)(win - 1) * (win - 1);
        glcm_group g0, g1;
        g0.S1 = sums[0][0] + sums[2][0]; g0.S2 = sums[0][1] + sums[2][1]; g0.Hq = sums[0][2] + sums[2][2];
        g0.sq = sqrt((double)A[0]) + sqrt((double)A[2]);
        g1.S1 = sums[1][0] + sums[3][0]; g1.S2 = sums[1][1] + sums[3][1]; g1.Hq = sums[1][2] + sums[3][2];
        g1.sq = sqrt((double)A[1]) + sqrt((double)A[3]);
        double r[4];
        for (int a = 0; a < 4; a++) r[a] = glcm_corr((a & 1) ? nb : na, sums[a][3], sums[a][4], sums[a][5]);
        glcm_finish(g0, g1, na, nb, r[0], r[1], r[2], r[3], (size_t)oy * ow + ox, out, gc);
    }
}

__global__ __launch_bounds__(256) void k4_glcm_wg(const uint8_t *__restrict__ q, int H, int W, int levels, int win, int step,
                                                  int oh, int ow, glcm_out out, glcm_consts gc)
{
    extern __shared__ unsigned int hist[];  // levels*levels
    __shared__ long long red[4][8];
    __shared__ long long sst[4][8];
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
        if (threadIdx.x < 8) sst[a][threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // sst[a] = np S1 S2 Hq M1 M2 Mx A ; all pixels are < levels, so np depends on the geometry only
        glcm_group g0, g1;
        g0.S1 = sst[0][1] + sst[2][1]; g0.S2 = sst[0][2] + sst[2][2]; g0.Hq = sst[0][3] + sst[2][3];
        g0.sq = sqrt((double)sst[0][7]) + sqrt((double)sst[2][7]);
        g1.S1 = sst[1][1] + sst[3][1]; g1.S2 = sst[1][2] + sst[3][2]; g1.Hq = sst[1][3] + sst[3][3];
        g1.sq = sqrt((double)sst[1][7]) + sqrt((double)sst[3][7]);
        double r[4];
        for (int a = 0; a < 4; a++) r[a] = glcm_corr(sst[a][0], sst[a][4], sst[a][5], sst[a][6]);
        glcm_finish(g0, g1, (long long)win * (win - 1), (long long)(win - 1) * (win - 1), r[0], r[1], r[2], r[3],
                    (size_t)oy * ow + ox, out, gc);
    }
}

static bool g_hq_ready[64] = {false};   // per device: the homogeneity tables are in place
static std::mutex g_hq_mu;              // contexts of several threads may arrive together

extern "C" int rsseg_glcm_u8(rsseg_ctx *ctx, const uint8_t *d_q, int H, int W, int levels, int win, int step,
                             float *const *d_props)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!d_q || !d_props || H < 1 || W < 1 || levels < 2 || levels > 256 || win < 2 || win > H || win > W || step < 1)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "glcm: bad arguments (H=%d W=%d levels=%d win=%d step=%d)", H, W, levels, win, step);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::unique_lock<std::mutex> hq_lock(g_hq_mu);
    if (!g_hq_ready[ctx->device & 63]) {
        long long lut[256];
        for (int d = 0; d < 256; d++) lut[d] = llrint(4503599627370496.0 / (1.0 + (double)d * (double)d));
        HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_glcm_hq), lut, sizeof(lut)));
        long long lut2[1024];
        for (int i = 0; i < 1024; i++) lut2[i] = lut[i & 31] + lut[i >> 5];
        HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_glcm_hq2), lut2, sizeof(lut2)));
        g_hq_ready[ctx->device & 63] = true;
    }
    hq_lock.unlock();
    const int oh = (H - win) / step + 1, ow = (W - win) / step + 1;
    glcm_out out;
    for (int i = 0; i < 5; i++) out.p[i] = d_props[i];
    {
        prof_scope ps(ctx, "glcm");
        const dim3 tg((ow + 63) / 64, (oh + 3) / 4);
        if (levels > 64) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "glcm: levels > 64 not supported");
        glcm_consts gc;
        {
            const long long na = (long long)win * (win - 1), nb = (long long)(win - 1) * (win - 1);
            gc.den4 = (double)(4 * na * nb);
            gc.den8 = (double)(8 * na * nb);
            gc.rden4 = 1.0 / gc.den4;
            gc.rden8 = 1.0 / gc.den8;
            // div_const needs a divisor whose significand is not all ones: true for these small integers, checked anyway
            for (double dv : {gc.den4, gc.den8}) {
                uint64_t b;
                memcpy(&b, &dv, 8);
                if ((b & 0xfffffffffffffull) == 0xfffffffffffffull) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "glcm: window size %d not supported", win);
            }
        }
#define GLCM_THREAD(WN)                                                                                                           \
    do {                                                                                                                          \
        if (levels <= 32) hipLaunchKernelGGL((k4_glcm_thread<WN, 3>), tg, dim3(256), 0, ctx->stream, d_q, H, W, step, oh, ow, out, gc); \
        else hipLaunchKernelGGL((k4_glcm_thread<WN, 2>), tg, dim3(256), 0, ctx->stream, d_q, H, W, step, oh, ow, out, gc);          \
    } while (0)
        if (win == 7 && step == 1 && levels <= 32) {
            const char *kv = getenv("RSSEG_GLCM_DENSE");      // "pair": the r02 kernel (two windows per thread), for A/B runs
            if (kv && !strcmp(kv, "pair")) {
                const dim3 pg((ow + 127) / 128, (oh + 3) / 4);   // two adjacent windows per thread
                hipLaunchKernelGGL(k4_glcm_pair, pg, dim3(256), 0, ctx->stream, d_q, H, W, oh, ow, out, gc);
            } else {
                const dim3 pg((ow + 127) / 128, (oh + 7) / 8);   // a 2 x 2 block of windows per thread
                hipLaunchKernelGGL(k4_glcm_quad, pg, dim3(256), 0, ctx->stream, d_q, H, W, oh, ow, out, gc);
            }
        } else if (win == 7) GLCM_THREAD(7);
        else if (win == 5) GLCM_THREAD(5);
        else if (win == 3) GLCM_THREAD(3);
        else {
            if (levels <= 32 && win < 256) {
                hipLaunchKernelGGL(k4_glcm_wave, dim3((unsigned)ceil_div64((int64_t)oh * ow, 4)), dim3(256), 0, ctx->stream, d_q, H, W, levels, win, step, oh,
                                   ow, out, gc);
            } else if (ow > 2147483647 || oh > 65535) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "glcm: output map too tall for the workgroup-per-window kernel");
            else hipLaunchKernelGGL(k4_glcm_wg, dim3(ow, oh), dim3(256), sizeof(unsigned int) * levels * levels, ctx->stream, d_q, H, W,
                               levels, win, step, oh, ow, out, gc);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}
