// K11 — random-forest inference: per-pixel walk of every tree, float64 vote accumulation in tree
// order, first-max argmax, classes_ lookup.
//
// Replaces predict_image (reference modules/supervised_classifiers.py:99-115) and
// supervised_classification_predict (modules/features/extract.py:690-719), i.e.
// RandomForestClassifier.predict: sklearn/ensemble/_forest.py:640 (float32 cast), 903-906, 948-962;
// Tree._apply_dense sklearn/tree/_tree.pyx:955-996.
//
// Forest layout in HBM: 16-byte nodes {float thr, int feature, int left, int right|missing<<31}.
// `X[i,f] <= threshold` compares a float32 feature with a float64 threshold; that is equivalent to
// comparing with the threshold rounded DOWN to float32, which is what thr holds.  Leaves have
// feature = -1 and left = row of the (n_leaves x n_classes) float64 value table.
// The pixel's features are staged once in LDS ([F][256] floats, bank = lane, conflict-free for any
// per-lane feature choice); nodes are read through L1/L2 (the forest is small against the 4 MiB L2
// for shallow forests, Infinity-Cache resident for depth-16 forests).  Gather-latency-bound, not
// HBM-bound: algorithmic HBM traffic is 4F B/px in + 8 B/px out.
#include <cmath>

#include "common.h"

struct __align__(16) rf_node {
    float thr;
    int feature;
    int left;
    unsigned right;  // bit 31: missing_go_to_left
};

#define RF_THREADS 256
#define RF_NCMAX 8

struct rf_planes {
    const float *p[RSSEG_MAX_FEATURES];
};

template <int NC>
__global__ __launch_bounds__(RF_THREADS) void k11_forest(rf_planes pl, int F, int64_t n, const rf_node *__restrict__ nodes,
                                                         const int *__restrict__ tree_off, int n_trees,
                                                         const double *__restrict__ leafval, int n_classes,
                                                         const long long *__restrict__ classes, long long *__restrict__ out)
{
    extern __shared__ float feat[];  // [F][RF_THREADS]
    const int64_t i = (int64_t)blockIdx.x * RF_THREADS + threadIdx.x;
    for (int f = 0; f < F; f++) feat[f * RF_THREADS + threadIdx.x] = i < n ? pl.p[f][i] : 0.f;
    // each lane reads back only what it wrote: no barrier needed
    if (i >= n) return;
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = 0.0;
    for (int t = 0; t < n_trees; t++) {
        const rf_node *tn = nodes + tree_off[t];
        rf_node nd = tn[0];
        while (nd.feature >= 0) {
            const float x = feat[nd.feature * RF_THREADS + threadIdx.x];
            const int right = (int)(nd.right & 0x7fffffffu);
            int next;
            if (x != x) next = (nd.right >> 31) ? nd.left : right;
            else next = x <= nd.thr ? nd.left : right;
            nd = tn[next];
        }
        const double *v = leafval + (size_t)nd.left * n_classes;
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (c < n_classes) acc[c] += v[c];
    }
    int best = 0;
    double bv = acc[0] / (double)n_trees;
#pragma unroll
    for (int c = 1; c < NC; c++)
        if (c < n_classes) {
            const double p = acc[c] / (double)n_trees;
            if (p > bv) { bv = p; best = c; }
        }
    out[i] = classes[best];
}

extern "C" int rsseg_forest_load(rsseg_ctx *ctx, int n_trees, const int64_t *tree_off, const int32_t *left, const int32_t *right,
                                 const int32_t *feature, const double *threshold, const uint8_t *missing_go_left,
                                 const double *value, int n_classes, const int64_t *classes, int n_features)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (n_trees < 1 || !tree_off || !left || !right || !feature || !threshold || !value || !classes)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_load: null argument");
    if (n_classes < 1 || n_classes > RF_NCMAX) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "forest_load: n_classes=%d outside [1,%d]", n_classes, RF_NCMAX);
    if (n_features < 1 || n_features > RSSEG_MAX_FEATURES) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "forest_load: n_features=%d outside [1,%d]", n_features, RSSEG_MAX_FEATURES);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t nn = tree_off[n_trees];
    if (nn < n_trees || nn > 0x7ffffff0) return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_load: bad node count %lld", (long long)nn);
    std::vector<rf_node> nodes((size_t)nn);
    std::vector<double> leaf;
    std::vector<int> toff(n_trees + 1);
    for (int t = 0; t <= n_trees; t++) toff[t] = (int)tree_off[t];
    for (int t = 0; t < n_trees; t++) {
        const int64_t b = tree_off[t], e = tree_off[t + 1];
        for (int64_t g = b; g < e; g++) {
            rf_node &nd = nodes[(size_t)g];
            if (left[g] == -1) {
                nd.thr = 0.f;
                nd.feature = -1;
                nd.left = (int)(leaf.size() / n_classes);
                nd.right = 0;
                for (int c = 0; c < n_classes; c++) leaf.push_back(value[(size_t)g * n_classes + c]);
            } else {
                if (left[g] < 0 || right[g] < 0 || b + left[g] >= e || b + right[g] >= e || feature[g] < 0 || feature[g] >= n_features)
                    return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_load: node %lld of tree %d is malformed", (long long)(g - b), t);
                float f = (float)threshold[g];
                if ((double)f > threshold[g]) f = nextafterf(f, -INFINITY);  // round toward -inf
                nd.thr = f;
                nd.feature = feature[g];
                nd.left = left[g];
                nd.right = (unsigned)right[g] | ((missing_go_left && missing_go_left[g]) ? 0x80000000u : 0u);
            }
        }
    }
    forest_dev &fd = ctx->forest;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (fd.d_nodes) HIPCHK(ctx, hipFree(fd.d_nodes));
    if (fd.d_leafval) HIPCHK(ctx, hipFree(fd.d_leafval));
    if (fd.d_treeoff) HIPCHK(ctx, hipFree(fd.d_treeoff));
    fd.d_nodes = fd.d_leafval = fd.d_treeoff = nullptr;
    HIPCHK(ctx, hipMalloc(&fd.d_nodes, nodes.size() * sizeof(rf_node)));
    HIPCHK(ctx, hipMalloc(&fd.d_leafval, leaf.size() * sizeof(double) + 64));
    HIPCHK(ctx, hipMalloc(&fd.d_treeoff, (toff.size() + n_classes * 2 + 2) * sizeof(long long)));
    HIPCHK(ctx, hipMemcpy(fd.d_nodes, nodes.data(), nodes.size() * sizeof(rf_node), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(fd.d_leafval, leaf.data(), leaf.size() * sizeof(double), hipMemcpyHostToDevice));
    // classes (int64) first, then the int32 tree offsets
    HIPCHK(ctx, hipMemcpy(fd.d_treeoff, classes, n_classes * sizeof(long long), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy((char *)fd.d_treeoff + n_classes * sizeof(long long), toff.data(), toff.size() * sizeof(int), hipMemcpyHostToDevice));
    fd.n_trees = n_trees;
    fd.n_classes = n_classes;
    fd.n_features = n_features;
    fd.n_nodes = nn;
    for (int c = 0; c < n_classes; c++) fd.classes[c] = classes[c];
    return RSSEG_OK;
}

extern "C" int rsseg_forest_predict(rsseg_ctx *ctx, const float *const *d_planes, int F, int64_t n, int64_t *d_out)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    forest_dev &fd = ctx->forest;
    if (!fd.d_nodes) return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_predict: no forest loaded");
    if (!d_planes || !d_out || n < 0) return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_predict: bad arguments");
    if (F != fd.n_features)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_predict: X has %d features, but the forest is expecting %d features as input", F, fd.n_features);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rf_planes pl;
    memset(&pl, 0, sizeof(pl));
    for (int f = 0; f < F; f++) {
        if (!d_planes[f]) return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_predict: plane %d is null", f);
        pl.p[f] = d_planes[f];
    }
    if (n == 0) return RSSEG_OK;
    const long long *d_classes = (const long long *)fd.d_treeoff;
    const int *d_toff = (const int *)((const char *)fd.d_treeoff + fd.n_classes * sizeof(long long));
    const size_t lds = sizeof(float) * (size_t)F * RF_THREADS;
    const unsigned grid = (unsigned)ceil_div64(n, RF_THREADS);
    {
        prof_scope ps(ctx, "forest");
        if (fd.n_classes <= 4)
            hipLaunchKernelGGL(k11_forest<4>, dim3(grid), dim3(RF_THREADS), lds, ctx->stream, pl, F, n, (const rf_node *)fd.d_nodes, d_toff,
                               fd.n_trees, (const double *)fd.d_leafval, fd.n_classes, d_classes, (long long *)d_out);
        else
            hipLaunchKernelGGL(k11_forest<RF_NCMAX>, dim3(grid), dim3(RF_THREADS), lds, ctx->stream, pl, F, n, (const rf_node *)fd.d_nodes,
                               d_toff, fd.n_trees, (const double *)fd.d_leafval, fd.n_classes, d_classes, (long long *)d_out);
    }
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}
