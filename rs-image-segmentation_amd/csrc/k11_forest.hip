// K11 — random-forest inference: per-pixel walk of every tree, float64 vote accumulation in tree
// order, first-max argmax, classes_ lookup.
//
// Replaces predict_image (reference modules/supervised_classifiers.py:99-115) and
// supervised_classification_predict (modules/features/extract.py:690-719), i.e.
// RandomForestClassifier.predict: sklearn/ensemble/_forest.py:640 (float32 cast), 903-906, 948-962;
// Tree._apply_dense sklearn/tree/_tree.pyx:955-996.
//
// Forest layout (built once by rsseg_forest_load): every tree is renumbered in BREADTH-FIRST order, so the two children
// of a node are adjacent (left = c, right = c + 1) and a node is 8 bytes:
//   internal  { float thr ; uint32 : bits 0-21 left child, bit 22 missing->left, bit 23 = 0, byte 3 = 4 * feature }
//   leaf      { NaN whose payload (bits 0-21) is the leaf's row in the vote table ; uint32 : bits 0-21 the leaf's OWN index,
//               bit 22 and bit 23 (leaf) set, byte 3 = 0 }
// Vote table: rows of NC = 4 / 8 / 16 / 32 / 64 float64 (n_classes padded with zeros); rows 0 .. n_classes-1 are the one-hot
// rows shared by every pure leaf, the mixed leaves follow.  Every leaf votes the same way — acc[c] += row[c], 16-byte
// loads, no branch; pure leaves of a wave hit the same few cache lines.
// `X[i,f] <= threshold` compares a float32 feature with a float64 threshold; that is equivalent to comparing with the
// threshold rounded DOWN to float32, which is what thr holds.  A walk step is next = left + (x > thr): on a leaf
// x > NaN is false for every x, so the step returns the leaf itself — leaves are fixed points and the inner loop needs
// no leaf test per chain, no select and no branch.
// The pixel's features are staged once in LDS (one row of F | 1 floats per pixel; 1024 pixels per workgroup up to 32
// features and 32 classes, 512 beyond — up to RSSEG_MAX_FEATURES = 64 features, 64 classes — so that the rows still
// leave room for tree nodes: the reference's non-hierarchical forest branch stacks every 2-D plane of the feature
// dictionary, scripts/3_classification.py:425-437).
//
// Two kernels:
//   k11_forest_lds   trees are taken in groups of <= 4 consecutive trees whose nodes (contiguous in the node array) fit
//                    the LDS beside the features; the whole group is copied with 16-byte loads (prefetched into
//                    registers during the walk of the previous group) and every step is two ds_reads.  Measured
//                    (profiles/r02_c5_pmc_sq.md): the walk is VALU-issue-bound — LDS array 13 % busy, bank conflicts
//                    2 % (neighbouring pixels share paths) — so the loop is written for instruction count: 9 VALU per
//                    tree step.
//   k11_forest_gen   any forest: the first ntop nodes of each of 4 trees in LDS, deeper nodes by global loads.
// Algorithmic HBM traffic is 4F B/px in + 8 B/px out; the kernel's bound is VALU issue, reported as node visits / s.
#include <algorithm>
#include <cmath>
#include <queue>

#include "common.h"

struct __align__(8) rf_node {
    float thr;
    unsigned bits;
};
// bits: [0,22) index of the left child within the tree (right = left + 1; a leaf: its own index), bit 22 missing-goes-
// left, bit 23 leaf, byte 3 = 4 * feature (0 on a leaf).  Byte 3 is a clean byte offset into a pixel's feature row, so
// the walk forms the feature address with ONE instruction (v_add_u32 with a byte-3 operand select) instead of
// shift + mask + add, and the child index needs one mask.
#define RF_LEAF 0x00800000u
#define RF_MISS 0x00400000u
#define RF_CHILD 0x003fffffu
#define RF_NAN_BITS 0x7fc00000u   // leaf thr: quiet NaN | payload
#define RF_PAY_MASK 0x003fffffu   // payload: row of the vote table

#define RF_NCMAX 64
static_assert(RSSEG_MAX_FEATURES <= 64, "byte 3 of a node holds 4 * feature");
#define RF_C 4        // trees walked at a time (independent chains of dependent LDS reads)

struct rf_planes {
    const float *p[RSSEG_MAX_FEATURES];
};

struct rf_tree {
    int node_off;  // first node of the tree in the node array
    int n_nodes;
    int leaf_off;  // first row of the tree in the leaf-value table
    int pad;
};

struct rf_group {   // k11_forest_lds: trees [first, first + count) whose nodes [node_base, node_base + n_nodes) share the LDS
    int first, count;
    int node_base;  // even (16-byte aligned copy); <= node_off of the first tree
    int n_nodes;    // nodes copied (from node_base)
};

typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef __attribute__((address_space(3))) const rf_node lds_cnode;
// an LDS address is 32 bits wide on the device; the host pass of the same source sees 64-bit pointers and would warn
#if defined(__HIP_DEVICE_COMPILE__)
#define RF_LDS_PTR(T, a) ((T *)(a))
#else
#define RF_LDS_PTR(T, a) ((T *)(uintptr_t)(a))
#endif

// The vote of a leaf: its row of the table, added in tree order (x + 0.0 == x, so a one-hot row adds a single 1.0).
template <int NC>
__device__ __forceinline__ void rf_row_load(const rf_node nd, const double *__restrict__ leafval, double (&row)[NC])
{
    const double2 *v = reinterpret_cast<const double2 *>(leafval + (size_t)(__float_as_uint(nd.thr) & RF_PAY_MASK) * NC);
#pragma unroll
    for (int c = 0; c < NC / 2; c++) {
        const double2 t = v[c];
        row[2 * c] = t.x;
        row[2 * c + 1] = t.y;
    }
}
template <int NC>
__device__ __forceinline__ void rf_vote(const rf_node nd, const double *__restrict__ leafval, double (&acc)[NC])
{
    double row[NC];
    rf_row_load<NC>(nd, leafval, row);
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] += row[c];
}

template <int NC>
__device__ __forceinline__ void rf_finish(const double (&acc)[NC], int n_trees, int n_classes, const long long *__restrict__ classes,
                                          long long *__restrict__ out, int64_t i)
{
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

template <int TH>
__device__ __forceinline__ int rf_stage_features(const rf_planes &pl, int F, int64_t n, int64_t i, float *feat)
{
    int my_nan = 0;
    for (int f = 0; f < F; f++) {
        const float v = i < n ? pl.p[f][i] : 0.f;
        my_nan |= v != v;
        feat[f * TH + threadIdx.x] = v;
    }
    return my_nan;
}

// ---- k11_forest_lds ------------------------------------------------------------------------------------------------
// A thread owns RF_PX pixels of its workgroup's 1024 and walks RF_C trees for each: RF_PX * RF_C independent chains.
// One round advances every chain by one node: all feature reads back to back, all node reads back to back, no control
// flow (leaves are fixed points).  Lanes leave the loop when all their chains sit on leaves.
#define RF_PX 1                      // pixels per thread (k11_forest_lds): TH threads per workgroup
#define RF_NCH (RF_PX * RF_C)

template <bool NANS>
__device__ __forceinline__ void rf_round_lds(rf_node (&nd)[RF_NCH], unsigned feat_tid, unsigned px_stride, const unsigned (&base)[RF_C])
{
    // feat_tid: LDS address of the thread's first pixel's feature row ([pixel][FP] floats, FP odd: conflict-free fills)
    float x[RF_NCH];
#pragma unroll
    for (int q = 0; q < RF_NCH; q++)   // chain q: pixel q / RF_C of the thread, tree q % RF_C of the group
        x[q] = *RF_LDS_PTR(lds_cfloat, feat_tid + (q / RF_C) * px_stride + (nd[q].bits >> 24));
#pragma unroll
    for (int q = 0; q < RF_NCH; q++) {
        bool go_right = x[q] > nd[q].thr;
        if (NANS) go_right = go_right || (x[q] != x[q] && !(nd[q].bits & RF_MISS));
        const unsigned next = (nd[q].bits & RF_CHILD) + (go_right ? 1u : 0u);
        lds_cnode *p = RF_LDS_PTR(lds_cnode, base[q % RF_C] + next * 8u);
        nd[q].thr = p->thr;
        nd[q].bits = p->bits;
    }
}

// Leaves are fixed points, so for deep forests (`two`: the host sets it from the deepest tree) the exit test runs every
// SECOND round: one test costs as much as a chain step, and a lane that reaches its last leaf after an odd number of
// rounds merely repeats it once.  Shallow forests (the reference's bundled model: depth <= 5) keep the test every round.
template <bool NANS>
__device__ __forceinline__ void rf_walk_lds(rf_node (&nd)[RF_NCH], unsigned feat_tid, unsigned px_stride, const unsigned (&base)[RF_C], bool two)
{
    for (;;) {
        unsigned all = nd[0].bits;
#pragma unroll
        for (int q = 1; q < RF_NCH; q++) all &= nd[q].bits;
        if (all & RF_LEAF) break;
        rf_round_lds<NANS>(nd, feat_tid, px_stride, base);
        if (two) rf_round_lds<NANS>(nd, feat_tid, px_stride, base);
    }
}

#define RF_NPRE (6 * RF_PX)   // 16-byte pieces (2 nodes) a thread prefetches per group: cap <= 12 * 1024 nodes

template <int NC, int RF_TH>
__global__ __launch_bounds__(RF_TH) void k11_forest_lds(rf_planes pl, int F, int64_t n, const rf_node *__restrict__ nodes,
                                                        const rf_tree *__restrict__ trees, const rf_group *__restrict__ groups, int n_groups,
                                                        int cap2 /* node area in 16-byte pieces */, int two_rounds, int n_trees,
                                                        const double *__restrict__ leafval,
                                                        int n_classes, const long long *__restrict__ classes, long long *__restrict__ out)
{
    constexpr int RF_LT = RF_TH / RF_PX;
    extern __shared__ __align__(16) char smem[];
    const int FP = F | 1;                                                      // odd row length: the fills below hit 64 distinct banks
    float *feat = reinterpret_cast<float *>(smem);                             // [RF_TH pixels][FP]
    uint4 *top = reinterpret_cast<uint4 *>(feat + (size_t)FP * RF_TH);         // the current group's nodes, two per uint4
    const int64_t i0 = (int64_t)blockIdx.x * RF_TH + threadIdx.x;             // pixel p of the thread: i0 + p * RF_LT
    int my_nan = 0;
    for (int f = 0; f < F; f++)
#pragma unroll
        for (int p = 0; p < RF_PX; p++) {
            const int64_t i = i0 + p * RF_LT;
            const float v = i < n ? pl.p[f][i] : 0.f;
            my_nan |= v != v;
            feat[(p * RF_LT + threadIdx.x) * FP + f] = v;
        }
    {
        const rf_group g0 = groups[0];
        const uint4 *src = reinterpret_cast<const uint4 *>(nodes + g0.node_base);
        for (int j = threadIdx.x; j < (g0.n_nodes + 1) / 2; j += RF_LT) top[j] = src[j];
        // a fixed-point leaf behind the node area for the unused chains of a group with fewer than RF_C trees
        if (threadIdx.x == 0) top[cap2] = make_uint4(RF_NAN_BITS, RF_LEAF | RF_MISS, RF_NAN_BITS, RF_LEAF | RF_MISS);
    }
    const bool any_nan = __syncthreads_or(my_nan) != 0;  // workgroup-uniform; also the barrier behind the fills above
    const unsigned feat_tid = (unsigned)(uintptr_t)(lds_cfloat *)feat + threadIdx.x * (unsigned)FP * 4u;
    const unsigned px_stride = (unsigned)RF_LT * (unsigned)FP * 4u;
    const unsigned top_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const uint4 *)top;
    double acc[RF_PX][NC];
#pragma unroll
    for (int p = 0; p < RF_PX; p++)
#pragma unroll
        for (int c = 0; c < NC; c++) acc[p][c] = 0.0;
    for (int g = 0; g < n_groups; g++) {
        const rf_group gr = groups[g];
        // the next group's nodes: one contiguous range, 16-byte loads, held in registers during the walk
        // six named registers, not an array: the compiler kept `uint4 pre[6]` in scratch memory across the walk (96 bytes
        // per lane written and read back per group = 800 B/px of HBM traffic: profiles/r02_c5_pmc_traffic.md, first r02 set)
        static_assert(RF_NPRE == 6, "the prefetch registers below are spelled out for six pieces per thread");
        const bool more = g + 1 < n_groups;
        const rf_group gn = groups[more ? g + 1 : g];
        const int npiece = more ? (gn.n_nodes + 1) / 2 : 0;
        const uint4 *psrc = reinterpret_cast<const uint4 *>(nodes + gn.node_base);
        const int plast = (gn.n_nodes + 1) / 2 - 1;
#define RF_PRE(r) const uint4 pre##r = psrc[(int)threadIdx.x + r * RF_LT < plast ? (int)threadIdx.x + r * RF_LT : plast];
        RF_PRE(0) RF_PRE(1) RF_PRE(2) RF_PRE(3) RF_PRE(4) RF_PRE(5)
#undef RF_PRE
        rf_node nd[RF_NCH];
        unsigned base[RF_C];
#pragma unroll
        for (int c = 0; c < RF_C; c++) {
            const rf_tree tr = trees[c < gr.count ? gr.first + c : gr.first];
            base[c] = c < gr.count ? top_addr + (unsigned)(tr.node_off - gr.node_base) * 8u : top_addr + (unsigned)cap2 * 16u;
            lds_cnode *p = RF_LDS_PTR(lds_cnode, base[c]);
#pragma unroll
            for (int px = 0; px < RF_PX; px++) {   // lanes without a pixel walk their zero features (no special case in the loop)
                nd[px * RF_C + c].thr = p->thr;
                nd[px * RF_C + c].bits = p->bits;
            }
        }
        if (any_nan) rf_walk_lds<true>(nd, feat_tid, px_stride, base, two_rounds != 0);
        else rf_walk_lds<false>(nd, feat_tid, px_stride, base, two_rounds != 0);
        // refill first (frees the prefetch registers: holding them AND the vote rows spilled 96 bytes per lane to scratch
        // memory, 800 B/px of extra HBM writes), then request the vote rows of all chains at once; the second barrier and
        // the next group's set-up cover most of the gather latency, and the rows are added in tree order behind it
        constexpr bool PIPE = NC <= 8;
        double rows[PIPE ? RF_NCH : 1][NC];
        if (PIPE) {
#pragma unroll
            for (int q = 0; q < RF_NCH; q++) rf_row_load<NC>(nd[q], leafval, rows[PIPE ? q : 0]);
        } else {
#pragma unroll
            for (int q = 0; q < RF_NCH; q++)
                if (q % RF_C < gr.count && i0 + (q / RF_C) * RF_LT < n) rf_vote<NC>(nd[q], leafval, acc[q / RF_C]);
        }
        if (more) {
            __syncthreads();  // every wave is done with the current group
#define RF_PUT(r) if ((int)threadIdx.x + r * RF_LT < npiece) top[threadIdx.x + r * RF_LT] = pre##r;
            RF_PUT(0) RF_PUT(1) RF_PUT(2) RF_PUT(3) RF_PUT(4) RF_PUT(5)
#undef RF_PUT
        }
        if (more) __syncthreads();
        if (PIPE) {
#pragma unroll
            for (int q = 0; q < RF_NCH; q++)
                if (q % RF_C < gr.count) {
#pragma unroll
                    for (int c = 0; c < NC; c++) acc[q / RF_C][c] += rows[PIPE ? q : 0][c];
                }
        }
    }
#pragma unroll
    for (int px = 0; px < RF_PX; px++)
        if (i0 + px * RF_LT < n) rf_finish<NC>(acc[px], n_trees, n_classes, classes, out, i0 + px * RF_LT);
}

// ---- k11_forest_gen ------------------------------------------------------------------------------------------------
// Every load of a round is issued unconditionally (a chain on its leaf reads node 0 of its block and keeps its leaf by a
// select): a load inside a per-chain `if` makes the compiler wait for each LDS read before it issues the next one,
// which serialises the chains.  Only a step to a node beyond the LDS block takes a predicated global load.
template <bool NANS, int RF_TH>
__device__ __forceinline__ void rf_walk_gen(rf_node (&nd)[RF_C], lds_cfloat *feat, lds_cnode *top, int ntop, const int (&lim)[RF_C],
                                            const rf_node *__restrict__ nodes, const int (&noff)[RF_C])
{
    for (;;) {
        unsigned all = 0xffffffffu;
#pragma unroll
        for (int c = 0; c < RF_C; c++) all &= nd[c].bits;
        if (all & RF_LEAF) break;   // every chain of this lane sits on a leaf (lanes leave the loop one by one)
        float x[RF_C];
        unsigned leafm[RF_C];  // all ones when the chain sits on its leaf (bit 31 of the node), as a mask: no control flow
#pragma unroll
        for (int c = 0; c < RF_C; c++) {
            leafm[c] = (unsigned)((int)(nd[c].bits << 8) >> 31);
            const unsigned f4 = nd[c].bits >> 24;          // 4 * feature; 0 on a leaf
            x[c] = feat[f4 * (RF_TH / 4) + threadIdx.x];
        }
        unsigned next[RF_C], out[RF_C];
        rf_node ld[RF_C];
#pragma unroll
        for (int c = 0; c < RF_C; c++) {
            bool go_right = x[c] > nd[c].thr;
            if (NANS) go_right = go_right || (x[c] != x[c] && !(nd[c].bits & RF_MISS));
            next[c] = (nd[c].bits & RF_CHILD) + (go_right ? 1u : 0u);
            out[c] = ((int)next[c] >= lim[c] ? 0xffffffffu : 0u) & ~leafm[c];
            const unsigned a = next[c] & ~(leafm[c] | out[c]);
            lds_cnode *p = top + c * ntop + a;
            ld[c].thr = p->thr;
            ld[c].bits = p->bits;
        }
#pragma unroll
        for (int c = 0; c < RF_C; c++) {
            nd[c].thr = leafm[c] ? nd[c].thr : ld[c].thr;
            nd[c].bits = leafm[c] ? nd[c].bits : ld[c].bits;
        }
        if (out[0] | out[1] | out[2] | out[3]) {
#pragma unroll
            for (int c = 0; c < RF_C; c++)
                if (out[c]) nd[c] = nodes[noff[c] + next[c]];
        }
    }
}

template <int NC, int RF_TH>
__global__ __launch_bounds__(RF_TH) void k11_forest_gen(rf_planes pl, int F, int64_t n, const rf_node *__restrict__ nodes,
                                                        const rf_tree *__restrict__ trees, int n_trees, int ntop,
                                                        const double *__restrict__ leafval, int n_classes,
                                                        const long long *__restrict__ classes, long long *__restrict__ out)
{
    extern __shared__ __align__(16) char smem[];
    float *feat = reinterpret_cast<float *>(smem);                                   // [F][RF_TH]
    rf_node *top = reinterpret_cast<rf_node *>(feat + (size_t)F * RF_TH);            // [RF_C][ntop]
    const int64_t i = (int64_t)blockIdx.x * RF_TH + threadIdx.x;
    const int my_nan = rf_stage_features<RF_TH>(pl, F, n, i, feat);
    constexpr int NPRE = 12;  // RF_C * ntop <= 12 * RF_TH nodes per group (the host keeps ntop <= 3 * RF_TH)
    for (int c = 0; c < RF_C && c < n_trees; c++) {
        const rf_tree t0 = trees[c];
        const int cnt = t0.n_nodes < ntop ? t0.n_nodes : ntop;
        for (int j = threadIdx.x; j < cnt; j += RF_TH) top[(size_t)c * ntop + j] = nodes[t0.node_off + j];
    }
    const bool any_nan = __syncthreads_or(my_nan) != 0;
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = 0.0;
    const int n_groups = (n_trees + RF_C - 1) / RF_C;
    for (int g = 0; g < n_groups; g++) {
        const int t = g * RF_C;
        rf_tree tr[RF_C];
#pragma unroll
        for (int c = 0; c < RF_C; c++) tr[c] = trees[t + c < n_trees ? t + c : n_trees - 1];
        rf_node pre[NPRE];
        int pcnt[RF_C], poff[RF_C];
        const bool more = t + RF_C < n_trees;
        if (more) {
#pragma unroll
            for (int c = 0; c < RF_C; c++) {
                const bool have = t + RF_C + c < n_trees;
                const rf_tree tnx = trees[have ? t + RF_C + c : n_trees - 1];
                pcnt[c] = have ? (tnx.n_nodes < ntop ? tnx.n_nodes : ntop) : 0;
                poff[c] = tnx.node_off;
            }
#pragma unroll
            for (int r = 0; r < NPRE; r++) {
                const int j = threadIdx.x + r * RF_TH;  // [0, RF_C * ntop): block c = j / ntop
                if (j < RF_C * ntop) {
                    const int c = j / ntop, o = j - c * ntop;
                    int cnt = pcnt[0], off = poff[0];
#pragma unroll
                    for (int cc = 1; cc < RF_C; cc++)
                        if (c == cc) { cnt = pcnt[cc]; off = poff[cc]; }
                    if (o < cnt) pre[r] = nodes[off + o];
                }
            }
        }
        {
            rf_node nd[RF_C];
            int lim[RF_C], noff[RF_C];
            lds_cfloat *lfeat = (lds_cfloat *)feat;
            lds_cnode *ltop = (lds_cnode *)top;
#pragma unroll
            for (int c = 0; c < RF_C; c++) {
                noff[c] = tr[c].node_off;
                lim[c] = tr[c].n_nodes < ntop ? tr[c].n_nodes : ntop;
                nd[c].thr = ltop[c * ntop].thr;
                nd[c].bits = ltop[c * ntop].bits;
                if (t + c >= n_trees || i >= n) nd[c].bits = RF_LEAF;  // no tree / no pixel: nothing to walk
            }
            if (any_nan) rf_walk_gen<true, RF_TH>(nd, lfeat, ltop, ntop, lim, nodes, noff);
            else rf_walk_gen<false, RF_TH>(nd, lfeat, ltop, ntop, lim, nodes, noff);
            if (i < n) {
#pragma unroll
                for (int c = 0; c < RF_C; c++)
                    if (t + c < n_trees) rf_vote<NC>(nd[c], leafval, acc);
            }
        }
        if (more) {
            __syncthreads();  // every wave is done with the current blocks
#pragma unroll
            for (int r = 0; r < NPRE; r++) {
                const int j = threadIdx.x + r * RF_TH;
                if (j < RF_C * ntop) {
                    const int c = j / ntop, o = j - c * ntop;
                    int cnt = pcnt[0];
#pragma unroll
                    for (int cc = 1; cc < RF_C; cc++)
                        if (c == cc) cnt = pcnt[cc];
                    if (o < cnt) top[j] = pre[r];
                }
            }
            __syncthreads();
        }
    }
    if (i < n) rf_finish<NC>(acc, n_trees, n_classes, classes, out, i);
}

// pixels per workgroup: 1024 while the feature rows and the vote accumulators allow it
static int rf_threads(int F, int n_classes) { return (F <= 32 && n_classes <= 32) ? 1024 : 512; }
// nodes of a tree group that fit the LDS beside the features of TH pixels (one 16-byte piece is kept for the dummy leaf)
static int rf_lds_cap(int F, int TH)
{
    const long bytes = 160L * 1024 - 256 - (long)(F | 1) * TH * 4 - 16;
    long cap = bytes / 8;
    cap = std::min<long>(cap, 2L * RF_NPRE * (TH / RF_PX)) & ~1L;
    return (int)std::max<long>(cap, 0);
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
    const int NCP = n_classes <= 4 ? 4 : (n_classes <= 8 ? 8 : (n_classes <= 16 ? 16 : (n_classes <= 32 ? 32 : 64)));  // row length of the vote table
    std::vector<double> leaf((size_t)n_classes * NCP, 0.0);
    for (int c = 0; c < n_classes; c++) leaf[(size_t)c * NCP + c] = 1.0;
    std::vector<rf_tree> trees(n_trees);
    std::vector<int> order, newid;
    int max_depth = 0;
    for (int t = 0; t < n_trees; t++) {
        const int64_t b = tree_off[t], e = tree_off[t + 1];
        const int cnt = (int)(e - b);
        if (cnt < 1 || cnt > (int)RF_CHILD) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "forest_load: tree %d has %d nodes (max %u)", t, cnt, RF_CHILD);
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
            // A NaN threshold (scikit-learn >= 1.4 writes one for a split that only separates missing from non-missing values:
            // `x <= NaN` is false, so every non-missing value goes RIGHT and a missing one where missing_go_to_left says) cannot
            // be a threshold here: NaN marks a leaf, and `x > thr` would send everything left.  Such a node is stored with its
            // children in the opposite order, threshold +inf and the missing flag inverted: `x > +inf` is false for every
            // non-missing x (first slot = the right child) and a missing x takes the second slot unless the stored flag is set.
            const bool swap = std::isnan(threshold[b + g]);
            const int first = swap ? r : l, second = swap ? l : r;
            newid[first] = (int)order.size();
            order.push_back(first);
            newid[second] = (int)order.size();
            order.push_back(second);
        }
        if ((int)order.size() != cnt) return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_load: tree %d has unreachable nodes", t);
        trees[t].node_off = (int)b;
        trees[t].n_nodes = cnt;
        {   // depth of the tree (breadth-first order: a child's depth is its parent's + 1)
            std::vector<int> depth((size_t)cnt, 0);
            for (int h = 0; h < cnt; h++) {
                const int g = order[h];
                if (left[b + g] != -1) depth[newid[left[b + g]]] = depth[newid[right[b + g]]] = depth[h] + 1;
                max_depth = std::max(max_depth, depth[h]);
            }
        }
        trees[t].leaf_off = 0;
        trees[t].pad = 0;
        for (int h = 0; h < cnt; h++) {
            const int64_t g = b + order[h];
            rf_node &nd = nodes[(size_t)(b + h)];
            if (left[g] == -1) {
                const double *v = value + (size_t)g * n_classes;
                int ones = 0, zeros = 0, cls = 0;
                for (int c = 0; c < n_classes; c++) {
                    if (v[c] == 1.0) { ones++; cls = c; }
                    else if (v[c] == 0.0) zeros++;
                }
                unsigned pay;
                if (ones == 1 && zeros == n_classes - 1) {
                    pay = (unsigned)cls;                    // the shared one-hot row of the class
                } else {
                    const size_t row = leaf.size() / NCP;
                    if (row > RF_PAY_MASK) return rs_fail(ctx, RSSEG_ERR_UNSUPPORTED, "forest_load: more than %u mixed leaves in the forest", RF_PAY_MASK);
                    pay = (unsigned)row;
                    for (int c = 0; c < NCP; c++) leaf.push_back(c < n_classes ? v[c] : 0.0);
                }
                const unsigned tb = RF_NAN_BITS | pay;   // the vote rides in the payload of a NaN threshold
                memcpy(&nd.thr, &tb, 4);
                nd.bits = RF_LEAF | RF_MISS | (unsigned)h;  // a leaf's "left child" is the leaf itself
            } else {
                const bool swap = std::isnan(threshold[g]);        // see the breadth-first numbering above
                float f = swap ? INFINITY : (float)threshold[g];
                if (!swap && (double)f > threshold[g]) f = nextafterf(f, -INFINITY);  // round toward -inf
                nd.thr = f;
                const bool miss_left = missing_go_left && missing_go_left[g];
                const int first = swap ? right[g] : left[g], second = swap ? left[g] : right[g];
                nd.bits = (unsigned)newid[first] | ((unsigned)feature[g] << 26) | ((miss_left != swap) ? RF_MISS : 0u);  // byte 3 = 4 * feature
                if (newid[second] != newid[first] + 1) return rs_fail(ctx, RSSEG_ERR_INVALID, "forest_load: internal layout error");
            }
        }
    }
    // ---- plan of k11_forest_lds: groups of <= RF_C consecutive trees whose nodes fit the LDS beside the features ----
    const int cap = rf_lds_cap(n_features, rf_threads(n_features, n_classes));
    std::vector<rf_group> groups;
    bool fits = true;
    for (int t = 0; t < n_trees && fits;) {
        rf_group g;
        g.first = t;
        g.node_base = trees[t].node_off & ~1;
        g.count = 0;
        g.n_nodes = 0;
        while (t < n_trees && g.count < RF_C && trees[t].node_off + trees[t].n_nodes - g.node_base <= cap) {
            g.n_nodes = trees[t].node_off + trees[t].n_nodes - g.node_base;
            g.count++;
            t++;
        }
        if (g.count == 0) fits = false;  // a single tree larger than the LDS area: the general kernel takes the forest
        else groups.push_back(g);
    }
    if (!fits) groups.clear();
    forest_dev &fd = ctx->forest;
    HIPCHK(ctx, rs_sync(ctx));
    if (fd.d_nodes) HIPCHK(ctx, hipFree(fd.d_nodes));
    if (fd.d_leafval) HIPCHK(ctx, hipFree(fd.d_leafval));
    if (fd.d_treeoff) HIPCHK(ctx, hipFree(fd.d_treeoff));
    if (fd.d_groups) HIPCHK(ctx, hipFree(fd.d_groups));
    fd.d_nodes = fd.d_leafval = fd.d_treeoff = fd.d_groups = nullptr;
    fd.n_groups = (int)groups.size();
    if (!groups.empty()) {
        HIPCHK(ctx, hipMalloc(&fd.d_groups, groups.size() * sizeof(rf_group)));
        HIPCHK(ctx, hipMemcpy(fd.d_groups, groups.data(), groups.size() * sizeof(rf_group), hipMemcpyHostToDevice));
    }
    HIPCHK(ctx, hipMalloc(&fd.d_nodes, nodes.size() * sizeof(rf_node) + 64));
    HIPCHK(ctx, hipMalloc(&fd.d_leafval, leaf.size() * sizeof(double) + 64));
    HIPCHK(ctx, hipMalloc(&fd.d_treeoff, n_classes * sizeof(long long) + trees.size() * sizeof(rf_tree) + 64));
    HIPCHK(ctx, hipMemcpy(fd.d_nodes, nodes.data(), nodes.size() * sizeof(rf_node), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(fd.d_leafval, leaf.data(), leaf.size() * sizeof(double), hipMemcpyHostToDevice));
    // classes (int64) first, then the per-tree records
    HIPCHK(ctx, hipMemcpy(fd.d_treeoff, classes, n_classes * sizeof(long long), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy((char *)fd.d_treeoff + n_classes * sizeof(long long), trees.data(), trees.size() * sizeof(rf_tree), hipMemcpyHostToDevice));
    fd.n_trees = n_trees;
    fd.max_depth = max_depth;
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
    const int TH = rf_threads(F, fd.n_classes);
    const unsigned grid = (unsigned)ceil_div64(n, TH);
    int rc;
    if (fd.n_groups > 0) {
        // every group of trees fits the LDS: features TH * (F | 1) * 4 B + cap nodes + the dummy leaf
        const int cap = rf_lds_cap(F, TH);
        const size_t lds = (size_t)(F | 1) * TH * 4 + (size_t)cap * sizeof(rf_node) + 16;
        auto launch = [&](auto kern) -> int {
            HIPCHK(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            prof_scope ps(ctx, "forest");
            hipLaunchKernelGGL(kern, dim3(grid), dim3(TH / RF_PX), lds, ctx->stream, pl, F, n, (const rf_node *)fd.d_nodes, d_trees,
                               (const rf_group *)fd.d_groups, fd.n_groups, cap / 2, fd.max_depth >= 10 ? 1 : 0, fd.n_trees, (const double *)fd.d_leafval, fd.n_classes, d_classes,
                               (long long *)d_out);
            return RSSEG_OK;
        };
        if (TH == 1024) {
            if (fd.n_classes <= 4) rc = launch(k11_forest_lds<4, 1024>);
            else if (fd.n_classes <= 8) rc = launch(k11_forest_lds<8, 1024>);
            else if (fd.n_classes <= 16) rc = launch(k11_forest_lds<16, 1024>);
            else rc = launch(k11_forest_lds<32, 1024>);
        } else {
            if (fd.n_classes <= 4) rc = launch(k11_forest_lds<4, 512>);
            else if (fd.n_classes <= 8) rc = launch(k11_forest_lds<8, 512>);
            else if (fd.n_classes <= 16) rc = launch(k11_forest_lds<16, 512>);
            else if (fd.n_classes <= 32) rc = launch(k11_forest_lds<32, 512>);
            else rc = launch(k11_forest_lds<64, 512>);
        }
    } else {
        // a tree larger than the LDS area: the first ntop (breadth-first) nodes of RF_C trees in LDS, in steps of 256
        int ntop = 3 * TH;
        while (ntop > 256 && (size_t)F * TH * 4 + (size_t)RF_C * ntop * sizeof(rf_node) > 158 * 1024) ntop -= 256;
        const size_t lds = (size_t)F * TH * 4 + (size_t)RF_C * ntop * sizeof(rf_node);
        auto launch = [&](auto kern) -> int {
            HIPCHK(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            prof_scope ps(ctx, "forest");
            hipLaunchKernelGGL(kern, dim3(grid), dim3(TH), lds, ctx->stream, pl, F, n, (const rf_node *)fd.d_nodes, d_trees, fd.n_trees, ntop,
                               (const double *)fd.d_leafval, fd.n_classes, d_classes, (long long *)d_out);
            return RSSEG_OK;
        };
        if (TH == 1024) {
            if (fd.n_classes <= 4) rc = launch(k11_forest_gen<4, 1024>);
            else if (fd.n_classes <= 8) rc = launch(k11_forest_gen<8, 1024>);
            else if (fd.n_classes <= 16) rc = launch(k11_forest_gen<16, 1024>);
            else rc = launch(k11_forest_gen<32, 1024>);
        } else {
            if (fd.n_classes <= 4) rc = launch(k11_forest_gen<4, 512>);
            else if (fd.n_classes <= 8) rc = launch(k11_forest_gen<8, 512>);
            else if (fd.n_classes <= 16) rc = launch(k11_forest_gen<16, 512>);
            else if (fd.n_classes <= 32) rc = launch(k11_forest_gen<32, 512>);
            else rc = launch(k11_forest_gen<64, 512>);
        }
    }
    if (rc != RSSEG_OK) return rc;
    HIPCHK(ctx, hipGetLastError());
    return stream_sync(ctx);
}
