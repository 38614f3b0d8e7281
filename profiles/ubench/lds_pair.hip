// LDS round trips of a forest walk, two tree LEVELS at a time (round 4).  Question: the walk of k11_forest_lds costs two LDS
// wave-instructions per node visit (feature, node) at ~4.26 cycles each; would a 16-byte "super-node" — a node together with the
// thresholds and feature indices of its two children, one ds_read_b128 per TWO levels — be cheaper?
//   old2      two rounds of { ds_read_b32 feature ; dependent ds_read_b64 node }                      4 LDS instructions / 2 levels
//   super3    ds_read_b32 feature of the node ; dependent ds_read_b32 feature of the chosen child ; dependent ds_read_b128   3 / 2 levels
//   super4    the three features (node, both children) in one go, then the ds_read_b128                4 / 2 levels, 2 dependent steps
// plus the raw rates of independent gathers (no dependent chain): b32 conflict-free, b32 / b64 / b128 at random addresses.
// Build: hipcc -O3 --offload-arch=gfx950 -o lds_pair lds_pair.hip ; run: ./lds_pair  (prints JSON)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((address_space(3))) const float lf;
typedef __attribute__((address_space(3))) const uint2 l2;
typedef __attribute__((address_space(3))) const uint4 l4;
typedef __attribute__((address_space(3))) const unsigned lu;

#define NFEAT 20
#define NPX 1024

// features: PIXEL-major rows of NFEAT | 1 floats (what k11_forest_lds stages) when PM, else feature-major [NFEAT][NPX]
template <bool PM>
__device__ __forceinline__ unsigned feat_addr(unsigned fbase, unsigned f)
{
    return PM ? fbase + f * 4u : fbase + f * (NPX * 4u);
}

template <int NCH, bool PM>
__global__ __launch_bounds__(1024) void k_old2(unsigned long long *t, unsigned *sink, int nodes, int steps)
{
    extern __shared__ __align__(16) char smem[];
    float *feat = (float *)smem;
    uint2 *tab = (uint2 *)(feat + (NFEAT | 1) * NPX);
    for (int j = threadIdx.x; j < (NFEAT | 1) * NPX; j += blockDim.x) feat[j] = (float)((j * 2654435761u) >> 8) * (1.0f / 16777216.0f);
    for (int j = threadIdx.x; j < nodes; j += blockDim.x) {
        const unsigned h = (j * 2246822519u) ^ 0x9e3779b9u;
        tab[j] = make_uint2(__float_as_uint(0.5f), (((h >> 7) % (nodes - 1)) & 0xffffffu) | ((h % NFEAT) << 24));
    }
    __syncthreads();
    const unsigned fbase = (unsigned)(uintptr_t)(lf *)feat + (PM ? threadIdx.x * (NFEAT | 1) * 4u : threadIdx.x * 4u), tbase = (unsigned)(uintptr_t)(l2 *)tab;
    uint2 nd[NCH];
#pragma unroll
    for (int q = 0; q < NCH; q++) nd[q] = tab[(threadIdx.x * 7 + q * 131) % nodes];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < 2 * steps; s++) {
        float x[NCH];
#pragma unroll
        for (int q = 0; q < NCH; q++) x[q] = *(lf *)feat_addr<PM>(fbase, nd[q].y >> 24);
#pragma unroll
        for (int q = 0; q < NCH; q++) {
            const unsigned next = (nd[q].y & 0xffffffu) + (x[q] > __uint_as_float(nd[q].x) ? 1u : 0u);
            l2 *p = (l2 *)(tbase + next * 8u);
            nd[q].x = p->x;
            nd[q].y = p->y;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    unsigned acc = 0;
#pragma unroll
    for (int q = 0; q < NCH; q++) acc += nd[q].y;
    if (acc == 0x12345) sink[0] = acc;
}

// super-node: x = thr(node), y = thr(left child), z = thr(right child), w = { bits 0-13: block of the four grandchildren (index / 4),
// 14-19 feature of the node, 20-25 feature of the left child, 26-31 feature of the right child }
template <int NCH, bool PM, bool THREE>
__global__ __launch_bounds__(1024) void k_super(unsigned long long *t, unsigned *sink, int snodes, int steps)
{
    extern __shared__ __align__(16) char smem[];
    float *feat = (float *)smem;
    uint4 *tab = (uint4 *)(feat + (NFEAT | 1) * NPX + 1);   // + 1 float: 16-byte alignment of the table ((NFEAT|1) * NPX is odd * 1024 -> already a multiple of 4 floats; keep simple)
    tab = (uint4 *)(((uintptr_t)tab + 15) & ~(uintptr_t)15);
    for (int j = threadIdx.x; j < (NFEAT | 1) * NPX; j += blockDim.x) feat[j] = (float)((j * 2654435761u) >> 8) * (1.0f / 16777216.0f);
    for (int j = threadIdx.x; j < snodes; j += blockDim.x) {
        const unsigned h = (j * 2246822519u) ^ 0x9e3779b9u;
        const unsigned blk = (h >> 7) % (snodes / 4 - 1);
        tab[j] = make_uint4(__float_as_uint(0.5f), __float_as_uint(0.25f), __float_as_uint(0.75f),
                            blk | ((h % NFEAT) << 14) | (((h >> 3) % NFEAT) << 20) | (((h >> 5) % NFEAT) << 26));
    }
    __syncthreads();
    const unsigned fbase = (unsigned)(uintptr_t)(lf *)feat + (PM ? threadIdx.x * (NFEAT | 1) * 4u : threadIdx.x * 4u), tbase = (unsigned)(uintptr_t)(l4 *)tab;
    uint4 nd[NCH];
#pragma unroll
    for (int q = 0; q < NCH; q++) nd[q] = tab[(threadIdx.x * 7 + q * 131) % snodes];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) {
        float xa[NCH], xb[NCH], xc[NCH];
#pragma unroll
        for (int q = 0; q < NCH; q++) {
            xa[q] = *(lf *)feat_addr<PM>(fbase, (nd[q].w >> 14) & 63u);
            if (!THREE) {
                xb[q] = *(lf *)feat_addr<PM>(fbase, (nd[q].w >> 20) & 63u);
                xc[q] = *(lf *)feat_addr<PM>(fbase, nd[q].w >> 26);
            }
        }
        bool r1[NCH];
        if (THREE) {
#pragma unroll
            for (int q = 0; q < NCH; q++) {
                r1[q] = xa[q] > __uint_as_float(nd[q].x);
                xb[q] = *(lf *)feat_addr<PM>(fbase, r1[q] ? nd[q].w >> 26 : (nd[q].w >> 20) & 63u);
            }
        }
#pragma unroll
        for (int q = 0; q < NCH; q++) {
            unsigned r2;
            if (THREE) {
                r2 = xb[q] > __uint_as_float(r1[q] ? nd[q].z : nd[q].y) ? 1u : 0u;
            } else {
                r1[q] = xa[q] > __uint_as_float(nd[q].x);
                r2 = (r1[q] ? xc[q] > __uint_as_float(nd[q].z) : xb[q] > __uint_as_float(nd[q].y)) ? 1u : 0u;
            }
            const unsigned next = (nd[q].w & 0x3fffu) * 4u + (r1[q] ? 2u : 0u) + r2;
            l4 *p = (l4 *)(tbase + next * 16u);
            nd[q].x = p->x; nd[q].y = p->y; nd[q].z = p->z; nd[q].w = p->w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    unsigned acc = 0;
#pragma unroll
    for (int q = 0; q < NCH; q++) acc += nd[q].w + nd[q].x;
    if (acc == 0x12345) sink[0] = acc;
}

// independent gathers: W = bytes per lane (4, 8, 16); RANDOM addresses or lane-consecutive (conflict-free) ones
template <int W, bool RANDOM>
__global__ __launch_bounds__(1024) void k_raw(unsigned long long *t, unsigned *sink, int steps)
{
    extern __shared__ __align__(16) char smem[];
    unsigned *tab = (unsigned *)smem;                     // 32768 dwords
    for (int j = threadIdx.x; j < 32768; j += blockDim.x) tab[j] = j * 2654435761u;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(lu *)tab;
    unsigned a[8], acc = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const unsigned h = (threadIdx.x * 2246822519u + q * 374761393u) ^ 0x9e3779b9u;
        const unsigned slot = RANDOM ? (h >> 9) % (32768 * 4 / W) : ((threadIdx.x + q * 1024) % (32768 * 4 / W));
        a[q] = base + slot * W;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (W == 4) acc += *(lu *)a[q];
            if (W == 8) { l2 *p = (l2 *)a[q]; acc += p->x ^ p->y; }
            if (W == 16) { l4 *p = (l4 *)a[q]; acc += p->x ^ p->y ^ p->z ^ p->w; }
        }
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (acc == 0x12345) sink[0] = acc;
}

static double median(std::vector<unsigned long long> v)
{
    std::sort(v.begin(), v.end());
    return (double)v[v.size() / 2];
}

template <typename K, typename... A>
static int run(const char *name, K kern, size_t lds, int threads, int steps, double per, unsigned long long *d_t, bool last, A... args)
{
    const int blocks = 256;
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<unsigned long long> h(blocks * 16);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_t, args..., steps);
        CHECK(hipDeviceSynchronize());
    }
    const int nw = blocks * threads / 64;
    CHECK(hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> v(h.begin(), h.begin() + nw);
    printf("  \"%s\": %.2f%s\n", name, median(v) / steps / per, last ? "" : ",");
    return 0;
}

int main()
{
    unsigned long long *d_t;
    unsigned *d_sink;
    CHECK(hipMalloc(&d_t, sizeof(unsigned long long) * 256 * 16));
    CHECK(hipMalloc(&d_sink, 64));
    const int steps = 1024;
    const size_t lds_old = (size_t)(NFEAT | 1) * NPX * 4 + 8192 * 8;
    const size_t lds_sup = (size_t)(NFEAT | 1) * NPX * 4 + 32 + 4096 * 16;
    printf("{\n \"note\": \"s_memtime ticks per TWO tree levels per wave (16 waves per CU = 1024 threads, 4 chains per thread unless named), "
           "medians over the waves of 256 workgroups; raw: ticks per gather wave-instruction, 8 independent gathers in flight per wave\",\n");
    printf(" \"two_levels\": {\n");
    if (run("old2_feature_major_4ch", k_old2<4, false>, lds_old, 1024, steps, 1.0, d_t, false, d_sink, 8192)) return 1;
    if (run("old2_pixel_major_4ch", k_old2<4, true>, lds_old, 1024, steps, 1.0, d_t, false, d_sink, 8192)) return 1;
    if (run("super3_feature_major_4ch", k_super<4, false, true>, lds_sup, 1024, steps, 1.0, d_t, false, d_sink, 4096)) return 1;
    if (run("super3_pixel_major_4ch", k_super<4, true, true>, lds_sup, 1024, steps, 1.0, d_t, false, d_sink, 4096)) return 1;
    if (run("super4_feature_major_4ch", k_super<4, false, false>, lds_sup, 1024, steps, 1.0, d_t, false, d_sink, 4096)) return 1;
    if (run("super4_pixel_major_4ch", k_super<4, true, false>, lds_sup, 1024, steps, 1.0, d_t, false, d_sink, 4096)) return 1;
    if (run("old2_pixel_major_8ch", k_old2<8, true>, lds_old, 1024, steps, 1.0, d_t, false, d_sink, 8192)) return 1;
    if (run("super3_pixel_major_8ch", k_super<8, true, true>, lds_sup, 1024, steps, 1.0, d_t, false, d_sink, 4096)) return 1;
    if (run("old2_pixel_major_4ch_8waves", k_old2<4, true>, lds_old, 512, steps, 1.0, d_t, false, d_sink, 8192)) return 1;
    if (run("super3_pixel_major_4ch_8waves", k_super<4, true, true>, lds_sup, 512, steps, 1.0, d_t, true, d_sink, 4096)) return 1;
    printf(" },\n \"raw_gather_ticks_per_instruction_16_waves\": {\n");
    if (run("b32_conflict_free", k_raw<4, false>, 131072, 1024, steps, 8.0 * 16.0, d_t, false, d_sink)) return 1;
    if (run("b32_random", k_raw<4, true>, 131072, 1024, steps, 8.0 * 16.0, d_t, false, d_sink)) return 1;
    if (run("b64_conflict_free", k_raw<8, false>, 131072, 1024, steps, 8.0 * 16.0, d_t, false, d_sink)) return 1;
    if (run("b64_random", k_raw<8, true>, 131072, 1024, steps, 8.0 * 16.0, d_t, false, d_sink)) return 1;
    if (run("b128_conflict_free", k_raw<16, false>, 131072, 1024, steps, 8.0 * 16.0, d_t, false, d_sink)) return 1;
    if (run("b128_random", k_raw<16, true>, 131072, 1024, steps, 8.0 * 16.0, d_t, true, d_sink)) return 1;
    printf(" }\n}\n");
    return 0;
}
