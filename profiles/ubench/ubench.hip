// Micro-benchmarks behind the roofline fractions of the two kernels that are NOT HBM-bound:
//   (a) issue cost of the VALU instructions that make up k4_glcm_thread<7,3> (v_pk_min_u16, v_pk_max_u16, v_perm_b32,
//       v_sad_u8, v_dot4, v_add3, v_fma_f64, v_add_f64, v_cvt ...), independent streams, 1 and 4 waves per SIMD;
//   (b) the LDS round trip the forest walk is made of: a dependent chain  node = lds[next(node, feature)]  with 1..8
//       independent chains per wave at 4 waves per SIMD.
// Build:  hipcc -O3 --offload-arch=gfx950 -o ubench ubench.hip     Run:  ./ubench > ubench.json
// Cycles are s_memtime ticks (shader clock) per wave-instruction on ONE SIMD, median over the workgroups.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

#define REP 64
#define ITER 256

// eight independent accumulators so that consecutive instructions never depend on each other
#define VALU_KERNEL(NAME, ASM, CONSTR_T)                                                                                  \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long *t, unsigned *sink, unsigned seed)                   \
    {                                                                                                                     \
        CONSTR_T a0 = (CONSTR_T)(threadIdx.x * 3 + seed), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,  \
                 a7 = a0 + 7, b = (CONSTR_T)(seed | 1), c = (CONSTR_T)(seed * 5 + 3);                                    \
        __syncthreads();                                                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        for (int it = 0; it < ITER; it++) {                                                                               \
            _Pragma("unroll") for (int r = 0; r < REP / 8; r++)                                                           \
            {                                                                                                             \
                asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                       \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                             : "v"(b), "v"(c));                                                                           \
            }                                                                                                             \
        }                                                                                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                      \
        if (a0 == (CONSTR_T)0x12345) sink[0] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                          \
    }

#define A_PKMIN(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define A_PKMAX(i) "v_pk_max_u16 %" #i ", %" #i ", %8\n"
#define A_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_PKSUB(i) "v_pk_sub_u16 %" #i ", %" #i ", %8\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_SAD(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
#define A_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 6\n"
#define A_FMAF32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_CVTF64(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define A_DOT2(i) "v_dot2_u32_u16 %" #i ", %" #i ", %8, %9\n"
#define A_ADD64E(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_CVTPK(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
VALU_KERNEL(k_xor_b32, A_XOR, unsigned)
VALU_KERNEL(k_lshlrev_b32, A_LSHL, unsigned)
VALU_KERNEL(k_or3_b32, A_OR3, unsigned)
VALU_KERNEL(k_dot2_u32_u16, A_DOT2, unsigned)
VALU_KERNEL(k_add_u32_e64, A_ADD64E, unsigned)
VALU_KERNEL(k_cndmask_b32, A_CNDMASK, unsigned)
VALU_KERNEL(k_mul_lo_u32, A_MULLO, unsigned)
VALU_KERNEL(k_cvt_f32_u32, A_CVTPK, unsigned)
VALU_KERNEL(k_pk_min_u16, A_PKMIN, unsigned)
VALU_KERNEL(k_pk_max_u16, A_PKMAX, unsigned)
VALU_KERNEL(k_pk_add_u16, A_PKADD, unsigned)
VALU_KERNEL(k_pk_sub_u16, A_PKSUB, unsigned)
VALU_KERNEL(k_perm_b32, A_PERM, unsigned)
VALU_KERNEL(k_sad_u8, A_SAD, unsigned)
VALU_KERNEL(k_dot4_u32_u8, A_DOT4, unsigned)
VALU_KERNEL(k_add3_u32, A_ADD3, unsigned)
VALU_KERNEL(k_add_u32, A_ADD, unsigned)
VALU_KERNEL(k_and_b32, A_AND, unsigned)
VALU_KERNEL(k_lshl_add_u32, A_LSHLADD, unsigned)
VALU_KERNEL(k_bfe_u32, A_BFE, unsigned)
VALU_KERNEL(k_fma_f32, A_FMAF32, unsigned)
VALU_KERNEL(k_mad_u32_u24, A_CVTF64, unsigned)

// float64 instructions: register pairs
#define VALU64_KERNEL(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long *t, unsigned *sink, unsigned seed)                   \
    {                                                                                                                     \
        double a0 = threadIdx.x * 0.5 + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7,  \
               b = 1.0000001, c = 1e-9;                                                                                   \
        __syncthreads();                                                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        for (int it = 0; it < ITER; it++) {                                                                               \
            _Pragma("unroll") for (int r = 0; r < REP / 8; r++)                                                           \
            {                                                                                                             \
                asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                       \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                             : "v"(b), "v"(c));                                                                           \
            }                                                                                                             \
        }                                                                                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                      \
        if (a0 == 0.12345) sink[0] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                                    \
    }
#define A_LDEXP64(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define A_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %8\n"
VALU64_KERNEL(k_ldexp_f64, A_LDEXP64)
VALU64_KERNEL(k_lshl_add_u64, A_LSHLADD64)
#define A_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define A_ADD64(i) "v_add_f64 %" #i ", %" #i ", %9\n"
#define A_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
VALU64_KERNEL(k_fma_f64, A_FMA64)
VALU64_KERNEL(k_add_f64, A_ADD64)
VALU64_KERNEL(k_mul_f64, A_MUL64)

// ---- (b) dependent LDS chains: table of 8-byte nodes {thr, next}, features [F][threads] ----
template <int NCH>
__global__ __launch_bounds__(1024) void k_lds_chain(unsigned long long *t, unsigned *sink, int nodes, int steps)
{
    extern __shared__ __align__(16) char smem[];
    float *feat = (float *)smem;                                 // [16][1024]
    uint2 *tab = (uint2 *)(feat + 16 * 1024);                    // nodes
    for (int j = threadIdx.x; j < 16 * 1024; j += blockDim.x) feat[j] = (float)((j * 2654435761u) >> 8) * (1.0f / 16777216.0f);
    for (int j = threadIdx.x; j < nodes; j += blockDim.x) {
        const unsigned h = (j * 2246822519u) ^ 0x9e3779b9u;
        tab[j] = make_uint2(__float_as_uint(0.5f), (((h >> 7) % (nodes - 1)) & 0xffffffu) | ((h & 15u) << 24));
    }
    __syncthreads();
    typedef __attribute__((address_space(3))) const float lf;
    typedef __attribute__((address_space(3))) const uint2 ln;
    const unsigned fbase = (unsigned)(uintptr_t)(lf *)feat + threadIdx.x * 4u, tbase = (unsigned)(uintptr_t)(ln *)tab;
    uint2 nd[NCH];
#pragma unroll
    for (int q = 0; q < NCH; q++) nd[q] = tab[(threadIdx.x * 7 + q * 131) % nodes];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) {
        float x[NCH];
#pragma unroll
        for (int q = 0; q < NCH; q++) x[q] = *(lf *)(fbase + ((nd[q].y >> 24) & 15u) * 4096u);
#pragma unroll
        for (int q = 0; q < NCH; q++) {
            const unsigned next = (nd[q].y & 0xffffffu) + (x[q] > __uint_as_float(nd[q].x) ? 1u : 0u);
            ln *p = (ln *)(tbase + next * 8u);
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

// ---- (c) the same chain with 4-BYTE nodes and integer features: what a rank-quantised forest would walk (VERDICT r02 item 6:
// thresholds replaced by their rank among the forest's thresholds of that feature, a pixel's features by their ranks, so that
// a node fits 32 bits: {rank : 12, feature : 4, next : 16} here) — one ds_read_b32 + one dependent ds_read_b32 per step ----
template <int NCH>
__global__ __launch_bounds__(1024) void k_lds_chain32(unsigned long long *t, unsigned *sink, int nodes, int steps)
{
    extern __shared__ __align__(16) char smem[];
    unsigned *feat = (unsigned *)smem;                           // [16][1024] ranks
    unsigned *tab = feat + 16 * 1024;                            // nodes
    for (int j = threadIdx.x; j < 16 * 1024; j += blockDim.x) feat[j] = ((j * 2654435761u) >> 20) & 0xfffu;
    for (int j = threadIdx.x; j < nodes; j += blockDim.x) {
        const unsigned h = (j * 2246822519u) ^ 0x9e3779b9u;
        tab[j] = (((h >> 7) % (nodes - 1)) & 0xffffu) | ((h & 15u) << 16) | (0x800u << 20);
    }
    __syncthreads();
    typedef __attribute__((address_space(3))) const unsigned lu;
    const unsigned fbase = (unsigned)(uintptr_t)(lu *)feat + threadIdx.x * 4u, tbase = (unsigned)(uintptr_t)(lu *)tab;
    unsigned nd[NCH];
#pragma unroll
    for (int q = 0; q < NCH; q++) nd[q] = tab[(threadIdx.x * 7 + q * 131) % nodes];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) {
        unsigned x[NCH];
#pragma unroll
        for (int q = 0; q < NCH; q++) x[q] = *(lu *)(fbase + ((nd[q] >> 16) & 15u) * 4096u);
#pragma unroll
        for (int q = 0; q < NCH; q++) {
            const unsigned next = (nd[q] & 0xffffu) + (x[q] > (nd[q] >> 20) ? 1u : 0u);
            nd[q] = *(lu *)(tbase + next * 4u);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    unsigned acc = 0;
#pragma unroll
    for (int q = 0; q < NCH; q++) acc += nd[q];
    if (acc == 0x12345) sink[0] = acc;
}

static double median(std::vector<unsigned long long> v)
{
    std::sort(v.begin(), v.end());
    return (double)v[v.size() / 2];
}

int main()
{
    unsigned long long *d_t;
    unsigned *d_sink;
    const int blocks = 256;
    CHECK(hipMalloc(&d_t, sizeof(unsigned long long) * blocks * 16));
    CHECK(hipMalloc(&d_sink, 64));
    std::vector<unsigned long long> h(blocks * 16);
    printf("{\n \"note\": \"s_memtime ticks per wave-instruction on one SIMD (independent instruction streams); waves_per_simd = threads / 256\",\n \"valu\": {\n");
    struct ent { const char *name; void (*fn)(unsigned long long *, unsigned *, unsigned); };
    const ent ents[] = {{"v_pk_min_u16", k_pk_min_u16}, {"v_pk_max_u16", k_pk_max_u16}, {"v_pk_add_u16", k_pk_add_u16}, {"v_pk_sub_u16", k_pk_sub_u16},
                        {"v_perm_b32", k_perm_b32}, {"v_sad_u8", k_sad_u8}, {"v_dot4_u32_u8", k_dot4_u32_u8}, {"v_add3_u32", k_add3_u32},
                        {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_bfe_u32", k_bfe_u32},
                        {"v_fma_f32", k_fma_f32}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64},
                        {"v_mul_f64", k_mul_f64}, {"v_xor_b32", k_xor_b32}, {"v_lshlrev_b32", k_lshlrev_b32}, {"v_or3_b32", k_or3_b32},
                        {"v_dot2_u32_u16", k_dot2_u32_u16}, {"v_add_u32_e64", k_add_u32_e64}, {"v_cndmask_b32", k_cndmask_b32},
                        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_ldexp_f64", k_ldexp_f64}, {"v_lshl_add_u64", k_lshl_add_u64}};
    const int nent = sizeof(ents) / sizeof(ents[0]);
    for (int e = 0; e < nent; e++) {
        printf("  \"%s\": {", ents[e].name);
        const int thr[3] = {256, 512, 1024};  // 1, 2, 4 waves per SIMD (one workgroup per CU)
        for (int k = 0; k < 3; k++) {
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(ents[e].fn, dim3(blocks), dim3(thr[k]), 0, 0, d_t, d_sink, 7u);
                CHECK(hipDeviceSynchronize());
            }
            const int nw = blocks * thr[k] / 64;
            CHECK(hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> v(h.begin(), h.begin() + nw);
            // the SIMD executed waves_per_simd * ITER * REP instructions in that time
            const double per = median(v) / ((double)ITER * REP) / (thr[k] / 256);
            printf("\"%d_waves_per_simd\": %.2f%s", thr[k] / 256, per, k < 2 ? ", " : "");
        }
        printf("}%s\n", e + 1 < nent ? "," : "");
    }
    printf(" },\n \"lds_dependent_chain\": {\n  \"note\": \"ticks per ROUND (one ds_read_b32 + one dependent ds_read_b64 per chain, all chains of a wave issued back to back), 16 waves per CU, 8192-node table\",\n");
    const size_t lds = 16 * 1024 * 4 + 8192 * 8;
    CHECK(hipFuncSetAttribute((const void *)k_lds_chain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void *)k_lds_chain<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void *)k_lds_chain<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void *)k_lds_chain<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int steps = 2048;
    const int thrs[3] = {256, 512, 1024};
    for (int k = 0; k < 3; k++) {
        printf("  \"%d_waves_per_cu\": {", thrs[k] / 64);
        for (int c = 0; c < 4; c++) {
            const int nch = 1 << c;
            for (int rep = 0; rep < 2; rep++) {
                if (nch == 1) hipLaunchKernelGGL(k_lds_chain<1>, dim3(blocks), dim3(thrs[k]), lds, 0, d_t, d_sink, 8192, steps);
                if (nch == 2) hipLaunchKernelGGL(k_lds_chain<2>, dim3(blocks), dim3(thrs[k]), lds, 0, d_t, d_sink, 8192, steps);
                if (nch == 4) hipLaunchKernelGGL(k_lds_chain<4>, dim3(blocks), dim3(thrs[k]), lds, 0, d_t, d_sink, 8192, steps);
                if (nch == 8) hipLaunchKernelGGL(k_lds_chain<8>, dim3(blocks), dim3(thrs[k]), lds, 0, d_t, d_sink, 8192, steps);
                CHECK(hipDeviceSynchronize());
            }
            const int nw = blocks * thrs[k] / 64;
            CHECK(hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> v(h.begin(), h.begin() + nw);
            printf("\"%d_chains\": %.1f%s", nch, median(v) / steps, c < 3 ? ", " : "");
        }
        printf("}%s\n", k < 2 ? "," : "");
    }
    printf(" },\n \"lds_dependent_chain_b32_nodes\": {\n  \"note\": \"the same round with 4-byte nodes and integer (rank) features: one ds_read_b32 + one dependent ds_read_b32 per chain\",\n");
    const size_t lds32 = 16 * 1024 * 4 + 8192 * 4;
    CHECK(hipFuncSetAttribute((const void *)k_lds_chain32<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32));
    CHECK(hipFuncSetAttribute((const void *)k_lds_chain32<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32));
    CHECK(hipFuncSetAttribute((const void *)k_lds_chain32<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32));
    CHECK(hipFuncSetAttribute((const void *)k_lds_chain32<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32));
    for (int k = 0; k < 3; k++) {
        printf("  \"%d_waves_per_cu\": {", thrs[k] / 64);
        for (int c = 0; c < 4; c++) {
            const int nch = 1 << c;
            for (int rep = 0; rep < 2; rep++) {
                if (nch == 1) hipLaunchKernelGGL(k_lds_chain32<1>, dim3(blocks), dim3(thrs[k]), lds32, 0, d_t, d_sink, 8192, steps);
                if (nch == 2) hipLaunchKernelGGL(k_lds_chain32<2>, dim3(blocks), dim3(thrs[k]), lds32, 0, d_t, d_sink, 8192, steps);
                if (nch == 4) hipLaunchKernelGGL(k_lds_chain32<4>, dim3(blocks), dim3(thrs[k]), lds32, 0, d_t, d_sink, 8192, steps);
                if (nch == 8) hipLaunchKernelGGL(k_lds_chain32<8>, dim3(blocks), dim3(thrs[k]), lds32, 0, d_t, d_sink, 8192, steps);
                CHECK(hipDeviceSynchronize());
            }
            const int nw = blocks * thrs[k] / 64;
            CHECK(hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> v(h.begin(), h.begin() + nw);
            printf("\"%d_chains\": %.1f%s", nch, median(v) / steps, c < 3 ? ", " : "");
        }
        printf("}%s\n", k < 2 ? "," : "");
    }
    printf(" }\n}\n");
    return 0;
}
