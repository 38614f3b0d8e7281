/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * KMeans restatement, one instantiation per element type (included twice from
 * oracle.c with T = float and T = double).  It follows, step by step, what
 * `unsupervised_kmeans_classification` (reference modules/features/extract.py:568-577)
 * makes scikit-learn 1.7.2 do:
 *
 *   MinMaxScaler().fit_transform      sklearn/preprocessing/_data.py:508-522, 555-556
 *   KMeans.fit: X -= X.mean(0)        sklearn/cluster/_kmeans.py:1476-1481
 *   _tolerance                        sklearn/cluster/_kmeans.py:279-287
 *   _kmeans_plusplus                  sklearn/cluster/_kmeans.py:213-270
 *   _euclidean_distances(_upcast)     sklearn/metrics/pairwise.py:391-437, 582-644
 *   _kmeans_single_lloyd              sklearn/cluster/_kmeans.py:699-748
 *   lloyd_iter_chunked_dense          sklearn/cluster/_k_means_lloyd.pyx:29-218
 *   _relocate_empty_clusters_dense / _average_centers / _center_shift
 *                                     sklearn/cluster/_k_means_common.pyx:167-311
 *
 * Where scikit-learn performs a LONG reduction (over all N pixels) its result
 * depends on BLAS / OpenMP summation order and is not reproducible run to run
 * (SURVEY.md §7 "Hard parts").  Every such reduction is restated here as an
 * EXACT fixed-point sum (quantum 2^-40, 128-bit accumulator), which is
 * independent of order and partition; the short per-pixel arithmetic keeps
 * scikit-learn's operation order and rounding (fma chains in T).
 * DESIGN.md §"KMeans numerics" states the same rules for the HIP kernels.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, TSUF)

/* numpy pairwise sum for a short contiguous T vector (n <= 128):
 * numpy/_core/src/umath/loops_utils.h.src  @TYPE@_pairwise_sum */
static T FN(np_pairwise_sum)(const T *a, int n)
{
    if (n < 8) {
        T res = (T)0; /* numpy starts from -0.0; irrelevant for the values used here */
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

static inline T FN(fixed_to_T)(i128 s)
{
    return (T)((double)s * FX_INV);
}

/* scaled + centred value sklearn clusters: fl(fl(fl(x*scale)+min_) - mean) */
static inline T FN(load_scaled)(const T *plane, int64_t i, T scale, T minv)
{
    T x = plane[i];
    if (x != x) x = (T)0; /* extract.py:548-556 NaN -> 0 */
    T xs = x * scale;
    xs = xs + minv;
    return xs;
}

/* d2 between candidate row c (T values) and pixel row y, sklearn upcast formula */
static inline T FN(d2_upcast)(const T *c, double cc, const T *y, int F)
{
    double dot = 0.0, yy = 0.0;
    for (int f = 0; f < F; f++) {
        dot = fma((double)c[f], (double)y[f], dot);
        yy = fma((double)y[f], (double)y[f], yy);
    }
    double d = -2.0 * dot;
    d = d + cc;
    d = d + yy;
    T dt = (T)d;
    return dt > (T)0 ? dt : (T)0; /* np.maximum(distances, 0) */
}

/*
 * planes     F pointers to N contiguous T values (band/feature-planar, as the GPU holds them)
 * u0_index   RandomState(42).choice(N, p=uniform) already resolved by the caller
 * uniforms   (k-1) * L draws of RandomState.uniform, L = 2 + floor(ln k)
 * relocated  out: number of empty-cluster relocations performed (0 in practice)
 * returns 0, or -1 on bad arguments
 */
int FN(oracle_kmeans_fit)(const T *const *planes, int64_t N, int F, int k,
                          int64_t u0_index, const double *uniforms, int max_iter, double tol_in,
                          int32_t *labels, double *centers_out, int32_t *n_iter_out,
                          double *scale_out, double *min_out, double *mean_out, double *tol_out,
                          int64_t *init_indices_out, int32_t *relocated)
{
    if (N <= 0 || F <= 0 || F > MAXF || k <= 0 || k > MAXK) return -1;
    const int L = 2 + (int)log((double)k);
    T scale[MAXF], minv[MAXF], mean[MAXF];
    *relocated = 0;

    /* ---- MinMaxScaler.fit ---- */
    for (int f = 0; f < F; f++) {
        T mn = (T)INFINITY, mx = (T)-INFINITY;
        for (int64_t i = 0; i < N; i++) {
            T x = planes[f][i];
            if (x != x) x = (T)0;
            if (x < mn) mn = x;
            if (x > mx) mx = x;
        }
        T range = mx - mn;
        if (range < (T)10 * TEPS) range = (T)1; /* _handle_zeros_in_scale */
        scale[f] = (T)1 / range;
        T t = mn * scale[f];
        minv[f] = (T)0 - t;
    }
    /* ---- X.mean(axis=0), np.var(X, axis=0) ---- */
    T var[MAXF];
    const T Nt = (T)N;
    for (int f = 0; f < F; f++) {
        i128 s = 0;
        for (int64_t i = 0; i < N; i++)
            s += (i128)llrint((double)FN(load_scaled)(planes[f], i, scale[f], minv[f]) * FX_ONE);
        T st = FN(fixed_to_T)(s);
        mean[f] = st / Nt;
        i128 v = 0;
        for (int64_t i = 0; i < N; i++) {
            T d = FN(load_scaled)(planes[f], i, scale[f], minv[f]) - mean[f];
            T dd = d * d;
            v += (i128)llrint((double)dd * FX_ONE);
        }
        T vt = FN(fixed_to_T)(v);
        var[f] = vt / Nt;
    }
    T tol;
    {
        T m = FN(np_pairwise_sum)(var, F) / (T)F;
        tol = m * (T)tol_in;
    }
    for (int f = 0; f < F; f++) {
        scale_out[f] = (double)scale[f];
        min_out[f] = (double)minv[f];
        mean_out[f] = (double)mean[f];
    }
    *tol_out = (double)tol;

    /* materialise the centred matrix, pixel-major, for the CPU walk */
    T *X = (T *)malloc((size_t)N * F * sizeof(T));
    T *closest = (T *)malloc((size_t)N * sizeof(T));
    T *cand_min = (T *)malloc((size_t)N * sizeof(T) * L);
    uint8_t *lab = (uint8_t *)malloc((size_t)N), *lab_old = (uint8_t *)malloc((size_t)N);
    if (!X || !closest || !cand_min || !lab || !lab_old) return -2;
    for (int f = 0; f < F; f++)
        for (int64_t i = 0; i < N; i++)
            X[i * F + f] = FN(load_scaled)(planes[f], i, scale[f], minv[f]) - mean[f];

    /* ---- k-means++ ---- */
    T C[MAXK][MAXF], Cnew[MAXK][MAXF];
    {
        const T *c0 = X + u0_index * F;
        for (int f = 0; f < F; f++) C[0][f] = c0[f];
        init_indices_out[0] = u0_index;
        double cc = 0.0;
        for (int f = 0; f < F; f++) cc = fma((double)c0[f], (double)c0[f], cc);
        u128 pot = 0;
        for (int64_t i = 0; i < N; i++) {
            closest[i] = FN(d2_upcast)(c0, cc, X + i * F, F);
            pot += (u128)llrint((double)closest[i] * FX_ONE);
        }
        T current_pot = (T)((double)pot * FX_INV);
        for (int c = 1; c < k; c++) {
            int64_t cand[8];
            for (int l = 0; l < L; l++) {
                double r = uniforms[(c - 1) * L + l] * (double)current_pot;
                /* smallest idx with prefix(idx) >= r  (np.searchsorted side='left' on stable_cumsum) */
                long double rl = ceill((long double)r * (long double)FX_ONE);
                u128 target = rl <= 0 ? 0 : (u128)rl;
                u128 run = 0;
                int64_t idx = N - 1; /* np.clip(candidate_ids, None, N-1) */
                for (int64_t i = 0; i < N; i++) {
                    run += (u128)llrint((double)closest[i] * FX_ONE);
                    if (run >= target) { idx = i; break; }
                }
                cand[l] = idx;
            }
            int best = 0;
            T best_pot = (T)0;
            for (int l = 0; l < L; l++) {
                const T *cl = X + cand[l] * F;
                double cc2 = 0.0;
                for (int f = 0; f < F; f++) cc2 = fma((double)cl[f], (double)cl[f], cc2);
                u128 p = 0;
                T *dst = cand_min + (size_t)l * N;
                for (int64_t i = 0; i < N; i++) {
                    T d = FN(d2_upcast)(cl, cc2, X + i * F, F);
                    T m = closest[i] < d ? closest[i] : d; /* np.minimum */
                    dst[i] = m;
                    p += (u128)llrint((double)m * FX_ONE);
                }
                T pt = (T)((double)p * FX_INV);
                if (l == 0 || pt < best_pot) { best = l; best_pot = pt; }
            }
            current_pot = best_pot;
            memcpy(closest, cand_min + (size_t)best * N, (size_t)N * sizeof(T));
            for (int f = 0; f < F; f++) C[c][f] = X[cand[best] * F + f];
            init_indices_out[c] = cand[best];
        }
    }

    /* ---- Lloyd ---- */
    memset(lab_old, 0xFF, (size_t)N);
    int strict = 0, it = 0;
    i128 S[MAXK][MAXF];
    int64_t cnt[MAXK];
    for (it = 0; it < max_iter; it++) {
        T csq[MAXK];
        for (int j = 0; j < k; j++) {
            T a = (T)0;
            for (int f = 0; f < F; f++) a = TFMA(C[j][f], C[j][f], a);
            csq[j] = a;
        }
        memset(S, 0, sizeof(S));
        memset(cnt, 0, sizeof(cnt));
        for (int64_t i = 0; i < N; i++) {
            const T *x = X + i * F;
            int bl = 0;
            T bd = (T)0;
            for (int j = 0; j < k; j++) {
                T a = (T)0;
                for (int f = 0; f < F; f++) a = TFMA(x[f], C[j][f], a);
                T d = TFMA((T)-2, a, csq[j]);
                if (j == 0 || d < bd) { bd = d; bl = j; }
            }
            lab[i] = (uint8_t)bl;
            cnt[bl]++;
            for (int f = 0; f < F; f++) S[bl][f] += (i128)llrint((double)x[f] * FX_ONE);
        }
        /* _relocate_empty_clusters_dense: farthest points (ties -> lowest index) feed empty clusters */
        int n_empty = 0, empty[MAXK];
        for (int j = 0; j < k; j++) if (cnt[j] == 0) empty[n_empty++] = j;
        if (n_empty > 0) {
            /* distances to the OLD centre of the assigned label, T arithmetic */
            char *taken = (char *)calloc((size_t)N, 1);
            T dmax_all = (T)0;
            for (int e = 0; e < n_empty; e++) {
                int64_t far = -1;
                T fd = (T)-1;
                for (int64_t i = 0; i < N; i++) {
                    if (taken[i]) continue;
                    T dd[MAXF];
                    for (int f = 0; f < F; f++) {
                        T t = X[i * F + f] - C[lab[i]][f];
                        dd[f] = t * t;
                    }
                    T d = FN(np_pairwise_sum)(dd, F);
                    if (d > fd) { fd = d; far = i; }
                }
                if (e == 0) dmax_all = fd;
                if (dmax_all == (T)0) break; /* np.max(distances) == 0: relocation pointless */
                taken[far] = 1;
                int nj = empty[e], oj = lab[far];
                for (int f = 0; f < F; f++) {
                    i128 q = (i128)llrint((double)X[far * F + f] * FX_ONE);
                    S[oj][f] -= q;
                    S[nj][f] = q;
                }
                cnt[nj] = 1;
                cnt[oj] -= 1;
                (*relocated)++;
            }
            free(taken);
        }
        /* _average_centers */
        int argmax_w = 0;
        for (int j = 1; j < k; j++) if (cnt[j] > cnt[argmax_w]) argmax_w = j;
        for (int j = 0; j < k; j++) {
            if (cnt[j] > 0) {
                T w = (T)cnt[j];
                T alpha = (T)(1.0 / (double)w);
                for (int f = 0; f < F; f++) Cnew[j][f] = FN(fixed_to_T)(S[j][f]) * alpha;
            }
        }
        for (int j = 0; j < k; j++)
            if (cnt[j] <= 0)
                for (int f = 0; f < F; f++) Cnew[j][f] = Cnew[argmax_w][f];
        /* _center_shift via _euclidean_dense_dense(new, old) */
        T shift2[MAXK];
        for (int j = 0; j < k; j++) {
            const T *a = Cnew[j], *b = C[j];
            T result = (T)0;
            int n4 = F / 4, rem = F % 4;
            for (int g = 0; g < n4; g++) {
                T t0 = (a[0] - b[0]) * (a[0] - b[0]);
                T t1 = (a[1] - b[1]) * (a[1] - b[1]);
                T t2 = (a[2] - b[2]) * (a[2] - b[2]);
                T t3 = (a[3] - b[3]) * (a[3] - b[3]);
                T g4 = ((t0 + t1) + t2) + t3;
                result = result + g4;
                a += 4; b += 4;
            }
            for (int r = 0; r < rem; r++) {
                T t = (a[r] - b[r]) * (a[r] - b[r]);
                result = result + t;
            }
            T sh = TSQRT(result);
            shift2[j] = sh * sh;
        }
        memcpy(C, Cnew, sizeof(C));
        int same = memcmp(lab, lab_old, (size_t)N) == 0;
        if (same) { strict = 1; it++; break; }
        T tot = FN(np_pairwise_sum)(shift2, k);
        if (tot <= tol) { it++; break; }
        memcpy(lab_old, lab, (size_t)N);
    }
    if (!strict) {
        T csq[MAXK];
        for (int j = 0; j < k; j++) {
            T a = (T)0;
            for (int f = 0; f < F; f++) a = TFMA(C[j][f], C[j][f], a);
            csq[j] = a;
        }
        for (int64_t i = 0; i < N; i++) {
            const T *x = X + i * F;
            int bl = 0;
            T bd = (T)0;
            for (int j = 0; j < k; j++) {
                T a = (T)0;
                for (int f = 0; f < F; f++) a = TFMA(x[f], C[j][f], a);
                T d = TFMA((T)-2, a, csq[j]);
                if (j == 0 || d < bd) { bd = d; bl = j; }
            }
            lab[i] = (uint8_t)bl;
        }
    }
    for (int64_t i = 0; i < N; i++) labels[i] = lab[i];
    for (int j = 0; j < k; j++)
        for (int f = 0; f < F; f++) centers_out[j * F + f] = (double)C[j][f];
    *n_iter_out = it;
    free(X); free(closest); free(cand_min); free(lab); free(lab_old);
    return 0;
}

#undef FN
#undef CAT
#undef CAT_
