// K11 — random-forest inference: per-pixel walk of every tree, float64 vote accumulation in tree
// order, first-max argmax, classes_ lookup.
//
// Replaces predict_image (reference modules/supervised_classifiers.py:99-115) and
// supervised_classification_predict (modules/features/extract.py:690-719), i.e.
// RandomForestClassifier.predict: sklearn/ensemble/_forest.py:640 (float32 cast), 903-906, 948-962;
// Tree._apply_dense sklearn/tree/_tree.pyx:955-996.
//
// Forest layout (built once by rsseg_forest_load): every tree is renumbered in BREADTH-FIRST order, so
//   * the two children of a node are adjacent (left = c, right = c + 1): a node is 8 bytes
//       { float thr ; uint32 : bits 0-23 left child (or leaf-value row), 24-29 feature, 30 missing->left, 31 leaf }
//   * the first NTOP nodes of a tree are its upper levels: the workgroup copies that block into LDS
//     (double-buffered, the copy of the next pair of trees overlaps the walk of the current pair) and only
//     the levels below it are gathered from L2 / Infinity Cache.
// `X[i,f] <= threshold` compares a float32 feature with a float64 threshold; that is equivalent to
// comparing with the threshold rounded DOWN to float32, which is what thr holds.
// The pixel's features are staged once in LDS ([F][1024] floats, bank = lane, conflict-free for any
// per-lane feature choice).  A leaf whose value row is one-hot carries its class in the node (no gather).
// Gather-latency-bound, not HBM-bound: algorithmic HBM traffic is 4F B/px in + 8 B/px out.
#include <algorithm>
#include <cmath>
#include <queue>

#include "common.h"

struct __align__(8) rf_node {
    float thr;
    unsigned bits;
};
#define RF_LEAF 0x80000000u
#define RF_MISS 0x40000000u
#define RF_PURE 0x20000000u  // leaf only: value row is one-hot, class in bits 24-28

#define RF_THREADS_MAX 1024
#define RF_NCMAX 8

struct rf_planes {
    const float *p[RSSEG_MAX_FEATURES];
};

struct rf_tree {
    int node_off;  // first node of the tree in the node array
    int n_nodes;
    int leaf_off;  // first row of the tree in the leaf-value table
    int pad;
};

// One step of a walk: from node `nd` of a tree whose first `lim` nodes are in LDS (`buf`), the rest in `tn`.
// The next node is fetched through ONE generic pointer (flat load: the aperture check per lane replaces a divergent
// LDS / global branch pair).  NANS = false: the workgroup's pixels hold no NaN, so the missing-value rule is skipped.
template <int RF_THREADS, bool NANS>
__device__ __forceinline__ rf_node rf_step(const rf_node nd, const float *__restrict__ feat, const rf_node *buf, int lim, const rf_node *tn)
{
    const float x = feat[((nd.bits >> 24) & 63u) * RF_THREADS + threadIdx.x];
    const unsigned left = nd.bits & 0xffffffu;
    bool go_left = x <= nd.thr;
    if (NANS) go_left = go_left || ((nd.bits & RF_MISS) != 0 && x != x);
    const unsigned next = left + (go_left ? 0u : 1u);
    const rf_node *p = (int)next < lim ? buf + next : tn + next;
    return *p;
}

template <int NC>
__device__ __forceinline__ void rf_vote(const rf_node nd, const rf_tree &tr, const double *__restrict__ leafval, int n_classes, double (&acc)[NC])
{
    if (nd.bits & RF_PURE) {
        const int cls = (nd.bits >> 24) & 31u;
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] += (c == cls) ? 1.0 : 0.0;
    } else {
        const double *v = leafval + (size_t)(tr.leaf_off + (int)(nd.bits & 0xffffffu)) * n_classes;
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (c < n_classes) acc[c] += v[c];
    }
}

// Trees are walked TWO AT A TIME per pixel (independent dependency chains: the walk is a chain of dependent
// LDS / L2 gathers, so a second chain nearly doubles what a wave keeps in flight); votes are still added in
// tree order.  LDS: features + 2 x 2 top blocks (the pair being walked, the pair being copied in).
template <int NC, int RF_THREADS>
__global__ __launch_bounds__(RF_THREADS) void k11_forest(rf_planes pl, int F, int64_t n, const rf_node *__restrict__ nodes,
                                                         const rf_tree *__restrict__ trees, int n_trees, int ntop,
                                                         const double *__restrict__ leafval, int n_classes,
                                                         const long long *__restrict__ classes, long long *__restrict__ out)
{
    extern __shared__ __align__(16) char smem[];
    float *feat = reinterpret_cast<float *>(smem);                                   // [F][RF_THREADS]
    rf_node *top = reinterpret_cast<rf_node *>(feat + (size_t)F * RF_THREADS);       // [2 pairs][2 trees][ntop]
    const int64_t i = (int64_t)blockIdx.x * RF_THREADS + threadIdx.x;
    int my_nan = 0;
    for (int f = 0; f < F; f++) {
        const float v = i < n ? pl.p[f][i] : 0.f;
        my_nan |= v != v;
        feat[f * RF_THREADS + threadIdx.x] = v;
    }
    const bool any_nan = __syncthreads_or(my_nan) != 0;  // workgroup-uniform
    constexpr int NPRE = 8192 / RF_THREADS;  // 2 * ntop <= 8192 nodes per pair
    // pair 0's top blocks
    for (int h = 0; h < 2 && h < n_trees; h++) {
        const rf_tree t0 = trees[h];
        const int cnt = t0.n_nodes < ntop ? t0.n_nodes : ntop;
        for (int j = threadIdx.x; j < cnt; j += RF_THREADS) top[(size_t)h * ntop + j] = nodes[t0.node_off + j];
    }
    __syncthreads();
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = 0.0;
    const int n_pairs = (n_trees + 1) / 2;
    for (int pr = 0; pr < n_pairs; pr++) {
        const int t = 2 * pr;
        const bool two = t + 1 < n_trees;
        const rf_tree trA = trees[t], trB = trees[two ? t + 1 : t];
        const rf_node *bufA = top + (size_t)((pr & 1) * 2) * ntop, *bufB = bufA + ntop;
        // issue the copy of the next pair's top blocks (held in registers during the walk)
        rf_node pre[NPRE];
        int cntA = 0, cntB = 0, offA = 0, offB = 0;
        if (t + 2 < n_trees) {
            const rf_tree tn = trees[t + 2];
            cntA = tn.n_nodes < ntop ? tn.n_nodes : ntop;
            offA = tn.node_off;
            if (t + 3 < n_trees) {
                const rf_tree tm = trees[t + 3];
                cntB = tm.n_nodes < ntop ? tm.n_nodes : ntop;
                offB = tm.node_off;
            }
#pragma unroll
            for (int r = 0; r < NPRE; r++) {
                const int j = threadIdx.x + r * RF_THREADS;  // [0, 2*ntop): first the A block, then the B block
                if (j < ntop) { if (j < cntA) pre[r] = nodes[offA + j]; }
                else if (j - ntop < cntB) pre[r] = nodes[offB + (j - ntop)];
            }
        }
        if (i < n) {
            const rf_node *tnA = nodes + trA.node_off, *tnB = nodes + trB.node_off;
            const int limA = trA.n_nodes < ntop ? trA.n_nodes : ntop, limB = trB.n_nodes < ntop ? trB.n_nodes : ntop;
            rf_node a = bufA[0], b = two ? bufB[0] : a;
            if (!two) b.bits = RF_LEAF;
            if (any_nan) {
                while (!((a.bits & b.bits) & RF_LEAF)) {
                    if (!(a.bits & RF_LEAF)) a = rf_step<RF_THREADS, true>(a, feat, bufA, limA, tnA);
                    if (!(b.bits & RF_LEAF)) b = rf_step<RF_THREADS, true>(b, feat, bufB, limB, tnB);
                }
            } else {
                while (!((a.bits & b.bits) & RF_LEAF)) {
                    if (!(a.bits & RF_LEAF)) a = rf_step<RF_THREADS, false>(a, feat, bufA, limA, tnA);
                    if (!(b.bits & RF_LEAF)) b = rf_step<RF_THREADS, false>(b, feat, bufB, limB, tnB);
                }
            }
            rf_vote<NC>(a, trA, leafval, n_classes, acc);
            if (two) rf_vote<NC>(b, trB, leafval, n_classes, acc);
        }
        if (t + 2 < n_trees) {
            rf_node *nb = top + (size_t)(((pr + 1) & 1) * 2) * ntop;
#pragma unroll
            for (int r = 0; r < NPRE; r++) {
                const int j = threadIdx.x + r * RF_THREADS;
                if (j < ntop) { if (j < cntA) nb[j] = pre[r]; }
                else if (j - ntop < cntB) nb[j] = pre[r];
            }
        }
        __syncthreads();
    }
    if (i >= n) return;
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
    std::vector<rf_tree> trees(n_trees);
    std::vector<int> order, newid;
    for (int t = 0; t < n_trees; t++) {
        const int64_t b = tree_off[t], e = tree_off[t + 1];
        const int cnt = (int)(e - b);
        if (cnt < 1 || cnt > 0xffffff) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "forest_load: tree %d has %d nodes (max 16777215)", t, cnt);
        // breadth-first order; children of a node end up adjacent
        order.clear();
        newid.assign(cnt, -1);
        order.push_back(0);
        newid[0] = 0;
        for (size_t h = 0; h < order.size(); h++) {
            const int g = order[h];
            if (left[b + g] == -1) continue;
            const int l = left[b + g], r = right[b + g];
            if (l < 0 || r < 0 || l >= cnt || r >= cnt || newid[l] != -1 || newid[r] != -1 || feature[b + g] < 0 || feature[b + g] >= n_features)
                return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_load: node %d of tree %d is malformed", g, t);
            newid[l] = (int)order.size();
            order.push_back(l);
            newid[r] = (int)order.size();
            order.push_back(r);
        }
        if ((int)order.size() != cnt) return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_load: tree %d has unreachable nodes", t);
        trees[t].node_off = (int)b;
        trees[t].n_nodes = cnt;
        trees[t].leaf_off = (int)(leaf.size() / n_classes);
        trees[t].pad = 0;
        int nleaf = 0;
        for (int h = 0; h < cnt; h++) {
            const int64_t g = b + order[h];
            rf_node &nd = nodes[(size_t)(b + h)];
            if (left[g] == -1) {
                nd.thr = 0.f;
                const double *v = value + (size_t)g * n_classes;
                int ones = 0, zeros = 0, cls = 0;
                for (int c = 0; c < n_classes; c++) {
                    if (v[c] == 1.0) { ones++; cls = c; }
                    else if (v[c] == 0.0) zeros++;
                }
                if (ones == 1 && zeros == n_classes - 1) {
                    nd.bits = RF_LEAF | RF_PURE | ((unsigned)cls << 24);
                } else {
                    nd.bits = RF_LEAF | (unsigned)nleaf;
                    nleaf++;
                    for (int c = 0; c < n_classes; c++) leaf.push_back(v[c]);
                }
            } else {
                float f = (float)threshold[g];
                if ((double)f > threshold[g]) f = nextafterf(f, -INFINITY);  // round toward -inf
                nd.thr = f;
                nd.bits = (unsigned)newid[left[g]] | ((unsigned)feature[g] << 24) | ((missing_go_left && missing_go_left[g]) ? RF_MISS : 0u);
                if (newid[right[g]] != newid[left[g]] + 1) return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_load: internal layout error");
            }
        }
    }
    if (leaf.empty()) leaf.push_back(0.0);
    forest_dev &fd = ctx->forest;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (fd.d_nodes) HIPCHK(ctx, hipFree(fd.d_nodes));
    if (fd.d_leafval) HIPCHK(ctx, hipFree(fd.d_leafval));
    if (fd.d_treeoff) HIPCHK(ctx, hipFree(fd.d_treeoff));
    fd.d_nodes = fd.d_leafval = fd.d_treeoff = nullptr;
    HIPCHK(ctx, hipMalloc(&fd.d_nodes, nodes.size() * sizeof(rf_node) + 64));
    HIPCHK(ctx, hipMalloc(&fd.d_leafval, leaf.size() * sizeof(double) + 64));
    HIPCHK(ctx, hipMalloc(&fd.d_treeoff, n_classes * sizeof(long long) + trees.size() * sizeof(rf_tree) + 64));
    HIPCHK(ctx, hipMemcpy(fd.d_nodes, nodes.data(), nodes.size() * sizeof(rf_node), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(fd.d_leafval, leaf.data(), leaf.size() * sizeof(double), hipMemcpyHostToDevice));
    // classes (int64) first, then the per-tree records
    HIPCHK(ctx, hipMemcpy(fd.d_treeoff, classes, n_classes * sizeof(long long), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy((char *)fd.d_treeoff + n_classes * sizeof(long long), trees.data(), trees.size() * sizeof(rf_tree), hipMemcpyHostToDevice));
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
    const rf_tree *d_trees = (const rf_tree *)((const char *)fd.d_treeoff + fd.n_classes * sizeof(long long));
    // workgroup size: 1024 pixels (16 waves hide the gathers below the LDS-resident top block);
    // LDS = features TH * F * 4 B + two top blocks of ntop 8-byte nodes
    const int TH = 1024;
    int ntop = 4096;  // per tree; four blocks resident (two pairs)
    while (ntop > 256 && (size_t)F * TH * 4 + 4 * (size_t)ntop * sizeof(rf_node) > 150 * 1024) ntop >>= 1;
    const size_t lds = (size_t)F * TH * 4 + 4 * (size_t)ntop * sizeof(rf_node);
    const unsigned grid = (unsigned)ceil_div64(n, TH);
    auto launch = [&](auto kern) -> int {
        HIPCHK(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_scope ps(ctx, "forest");
        hipLaunchKernelGGL(kern, dim3(grid), dim3(TH), lds, ctx->stream, pl, F, n, (const rf_node *)fd.d_nodes, d_trees, fd.n_trees, ntop,
                           (const double *)fd.d_leafval, fd.n_classes, d_classes, (long long *)d_out);
        return RSSEG_OK;
    };
    int rc;
    if (fd.n_classes <= 4) rc = launch(k11_forest<4, 1024>);
    else rc = launch(k11_forest<RF_NCMAX, 1024>);
    if (rc != RSSEG_OK) return rc;
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}
