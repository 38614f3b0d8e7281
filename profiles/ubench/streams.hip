// Read bandwidth of the k-means access pattern: a workgroup walks a chunk of 16 tiles x 1024 pixels; per tile every lane
// requests one 16-byte vector from each of NS planes back to back (NS x 1 KiB in flight per wave), as km_lloyd / km_kpp do.
// Total bytes are the same for every NS (the planes shrink as NS grows), so the numbers compare the access pattern only.
// RW = 1 adds a read-modify-write plane (k-means++'s closest-distance plane).
// Build: hipcc -O3 --offload-arch=gfx950 -o streams streams.hip     Run: ./streams > streams.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct planes { const float *p[32]; };

template <int NS, int RW>
__global__ __launch_bounds__(256) void k_read(planes pl, float *rw, int64_t n, float *sink)
{
    const int64_t chunk0 = (int64_t)blockIdx.x * 16384;
    float acc = 0.f;
    for (int t = 0; t < 16; t++) {
        const int64_t base = chunk0 + (int64_t)t * 1024 + threadIdx.x * 4;
        if (base + 4 > n) break;
        float4 v[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) v[s] = *reinterpret_cast<const float4 *>(pl.p[s] + base);
        float4 c = make_float4(0, 0, 0, 0);
        if (RW) c = *reinterpret_cast<const float4 *>(rw + base);
#pragma unroll
        for (int s = 0; s < NS; s++) acc += v[s].x + v[s].y + v[s].z + v[s].w;
        if (RW) {
            c.x += acc; c.y += acc; c.z += acc; c.w += acc;
            if (RW == 2) {   // non-temporal stores
                __builtin_nontemporal_store(c.x, rw + base); __builtin_nontemporal_store(c.y, rw + base + 1);
                __builtin_nontemporal_store(c.z, rw + base + 2); __builtin_nontemporal_store(c.w, rw + base + 3);
            } else if (RW == 3) {   // write to a second plane (no read-modify-write of one address)
                *reinterpret_cast<float4 *>(rw + n + base) = c;
            } else {
                *reinterpret_cast<float4 *>(rw + base) = c;
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// The k-means read pattern with another workgroup -> pixel mapping: SHAPE 1 = a workgroup reads 16 tiles of 1024 px that lie
// RL px apart (a "strip" of a virtual RL-px-wide raster: consecutive workgroups take consecutive tiles of the same 16 virtual
// rows), instead of 16 consecutive tiles (SHAPE 0, what the kernels do).  Pixel order is free for KMeans (exact sums).
template <int NS, int RW, int SHAPE>
__global__ __launch_bounds__(256) void k_read_shape(planes pl, float *rw, int64_t n, float *sink, int64_t RL)
{
    float acc = 0.f;
    const int64_t gx = RL / 1024;
    const int64_t tx = blockIdx.x % gx, by = blockIdx.x / gx;
    for (int t = 0; t < 16; t++) {
        const int64_t base = SHAPE == 0 ? (int64_t)blockIdx.x * 16384 + (int64_t)t * 1024 + threadIdx.x * 4
                                        : (by * 16 + t) * RL + tx * 1024 + threadIdx.x * 4;
        if (base + 4 > n) break;
        float4 v[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) v[s] = *reinterpret_cast<const float4 *>(pl.p[s] + base);
        float4 c = make_float4(0, 0, 0, 0);
        if (RW) c = *reinterpret_cast<const float4 *>(rw + base);
#pragma unroll
        for (int s = 0; s < NS; s++) acc += v[s].x + v[s].y + v[s].z + v[s].w;
        if (RW) {
            c.x += acc; c.y += acc; c.z += acc; c.w += acc;
            *reinterpret_cast<float4 *>(rw + base) = c;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int NS, int RW, int SHAPE> double run_shape(const float *buf, float *rw, float *sink, int64_t RL)
{
    const int64_t n = (int64_t)16384 * 16384;
    planes pl;
    for (int s = 0; s < 32; s++) pl.p[s] = buf + (int64_t)(s < NS ? s : 0) * n;
    const unsigned grid = (unsigned)(n / 16384);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_read_shape<NS, RW, SHAPE>), dim3(grid), dim3(256), 0, 0, pl, rw, n, sink, RL);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_read_shape<NS, RW, SHAPE>), dim3(grid), dim3(256), 0, 0, pl, rw, n, sink, RL);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return (double)n * 4.0 * (NS + (RW ? 2 : 0)) / (ms / reps * 1e-3) / 1e12;
}

// Write-heavy pattern of the fused index + projection pass (k3_indices_project, config 3): NR float32 planes read (7 raw
// bands = 28 B/px), NW float32 planes written (7 indices + 3 components = 40 B/px) and one uint8 plane written (1 B/px).
// MAP 0: grid-stride over 16-byte vectors with a persistent grid (what the kernel does); MAP 1: a workgroup owns a contiguous
// chunk of 16 tiles x 1024 px (the k-means kernels' mapping).  NT 1: non-temporal stores; NT 2: non-temporal loads too; NT 3: non-temporal loads, plain stores (what the kernel does).
struct wplanes { float *p[16]; uint8_t *q; };
template <int NR, int NW, int MAP, int NT>
__global__ __launch_bounds__(256) void k_rw(planes pl, wplanes wp, int64_t n)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int64_t n4 = n >> 2;
    auto body = [&](int64_t i) {
        f4v v[NR];
#pragma unroll
        for (int s = 0; s < NR; s++)
            v[s] = NT >= 2 ? __builtin_nontemporal_load(reinterpret_cast<const f4v *>(pl.p[s]) + i) : reinterpret_cast<const f4v *>(pl.p[s])[i];
        f4v acc = v[0];
#pragma unroll
        for (int s = 1; s < NR; s++) acc += v[s];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const f4v o = acc * (float)(w + 1);
            if (NT == 1 || NT == 2) __builtin_nontemporal_store(o, reinterpret_cast<f4v *>(wp.p[w]) + i);
            else reinterpret_cast<f4v *>(wp.p[w])[i] = o;
        }
        const uint32_t qq = (uint32_t)(int)acc[0] & 0xff;
        if (NT == 1 || NT == 2) __builtin_nontemporal_store(qq * 0x01010101u, reinterpret_cast<uint32_t *>(wp.q) + i);
        else reinterpret_cast<uint32_t *>(wp.q)[i] = qq * 0x01010101u;
    };
    if (MAP == 0) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) body(i);
    } else {
        for (int64_t c = blockIdx.x; c * 4096 < n4; c += gridDim.x)
            for (int t = 0; t < 16; t++) {
                const int64_t i = c * 4096 + t * 256 + threadIdx.x;
                if (i < n4) body(i);
            }
    }
}
template <int NR, int NW, int MAP, int NT> double run_rw(const float *buf, float *wbuf, unsigned grid, int64_t n, double *ms_out)
{
    planes pl;
    wplanes wp;
    for (int s = 0; s < 32; s++) pl.p[s] = buf + (int64_t)(s < NR ? s : 0) * n;
    for (int s = 0; s < 16; s++) wp.p[s] = wbuf + (int64_t)(s < NW ? s : 0) * n;
    wp.q = (uint8_t *)(wbuf + (int64_t)NW * n);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_rw<NR, NW, MAP, NT>), dim3(grid), dim3(256), 0, 0, pl, wp, n);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0.f;
    const int reps = 8;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k_rw<NR, NW, MAP, NT>), dim3(grid), dim3(256), 0, 0, pl, wp, n);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
        sum += ms;
    }
    *ms_out = sum / reps;
    const double bytes = (double)n * (4.0 * NR + 4.0 * NW + 1.0);
    return bytes / (sum / reps * 1e-3) / 1e12;
}

// Plane copy with 4-byte or 16-byte accesses per lane (what bounds k5_resize: one float per lane per store, one wave
// instruction = 256 B): grid-stride, 8192 workgroups, 16384^2 float32 in, 16384^2 float32 out (8 B/px).
template <int VEC>
__global__ __launch_bounds__(256) void k_copy(const float *__restrict__ src, float *__restrict__ dst, int64_t n)
{
    if (VEC == 4) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += (int64_t)gridDim.x * 256)
            reinterpret_cast<f4v *>(dst)[i] = reinterpret_cast<const f4v *>(src)[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
    }
}
// the same with the 256-column strip / row walk of k5_resize: a workgroup owns 256 columns x R rows, one float per lane per row
__global__ __launch_bounds__(256) void k_copy_strip(const float *__restrict__ src, float *__restrict__ dst, int H, int W, int gx, int R)
{
    const int tx = (int)(blockIdx.x % (unsigned)gx), by = (int)(blockIdx.x / (unsigned)gx);
    const int px = tx * 256 + (int)threadIdx.x;
    if (px >= W) return;
    const int r1 = (by + 1) * R < H ? (by + 1) * R : H;
#pragma unroll 8
    for (int r = by * R; r < r1; r++) dst[(size_t)r * W + px] = src[(size_t)r * W + px];
}
template <int VEC> double run_copy(const float *src, float *dst, int64_t n, double *ms_out)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_copy<VEC>), dim3(8192), dim3(256), 0, 0, src, dst, n);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k_copy<VEC>), dim3(8192), dim3(256), 0, 0, src, dst, n);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    *ms_out = ms / 10;
    return (double)n * 8.0 / (ms / 10 * 1e-3) / 1e12;
}

template <int NS, int RW> double run(const float *buf, float *rw, float *sink, int64_t total_floats)
{
    const int64_t n = (total_floats / NS) & ~(int64_t)16383;
    planes pl;
    for (int s = 0; s < 32; s++) pl.p[s] = buf + (int64_t)(s < NS ? s : 0) * n;
    const unsigned grid = (unsigned)(n / 16384);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_read<NS, RW>), dim3(grid), dim3(256), 0, 0, pl, rw, n, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_read<NS, RW>), dim3(grid), dim3(256), 0, 0, pl, rw, n, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)n * 4.0 * (NS + (RW ? 2 : 0));
    return bytes / (ms / reps * 1e-3) / 1e12;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "shape")) {
        const int64_t n = (int64_t)16384 * 16384;
        float *buf, *rw, *sink;
        CHECK(hipMalloc(&buf, (size_t)n * 4 * 15));
        CHECK(hipMalloc(&rw, (size_t)n * 4));
        CHECK(hipMalloc(&sink, 64));
        CHECK(hipMemset(buf, 0, (size_t)n * 4 * 15));
        CHECK(hipMemset(rw, 0, (size_t)n * 4));
        printf("{\n \"note\": \"TB/s of 15 float32 planes of 16384^2 px read with 16-byte loads (16 GB), read-only / with a read-modify-write plane, against the workgroup -> pixel mapping\",\n");
        printf(" \"chunk_of_16_consecutive_tiles\": {\"read_only\": %.3f, \"with_rw_plane\": %.3f},\n", run_shape<15, 0, 0>(buf, rw, sink, 16384), run_shape<15, 1, 0>(buf, rw, sink, 16384));
        printf(" \"strip_row_length_4096px\": {\"read_only\": %.3f, \"with_rw_plane\": %.3f},\n", run_shape<15, 0, 1>(buf, rw, sink, 4096), run_shape<15, 1, 1>(buf, rw, sink, 4096));
        printf(" \"strip_row_length_16384px\": {\"read_only\": %.3f, \"with_rw_plane\": %.3f},\n", run_shape<15, 0, 1>(buf, rw, sink, 16384), run_shape<15, 1, 1>(buf, rw, sink, 16384));
        printf(" \"strip_row_length_65536px\": {\"read_only\": %.3f, \"with_rw_plane\": %.3f},\n", run_shape<15, 0, 1>(buf, rw, sink, 65536), run_shape<15, 1, 1>(buf, rw, sink, 65536));
        printf(" \"strip_row_length_262144px\": {\"read_only\": %.3f, \"with_rw_plane\": %.3f}\n}\n", run_shape<15, 0, 1>(buf, rw, sink, 262144), run_shape<15, 1, 1>(buf, rw, sink, 262144));
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "copy")) {
        const int64_t n = (int64_t)16384 * 16384;
        float *src, *dst;
        CHECK(hipMalloc(&src, (size_t)n * 4));
        CHECK(hipMalloc(&dst, (size_t)n * 4));
        CHECK(hipMemset(src, 0, (size_t)n * 4));
        CHECK(hipMemset(dst, 0, (size_t)n * 4));
        double ms1, ms4;
        const double r1 = run_copy<1>(src, dst, n, &ms1), r4 = run_copy<4>(src, dst, n, &ms4);
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        const int gx = 64, R = 128;
        hipLaunchKernelGGL(k_copy_strip, dim3(gx * 128), dim3(256), 0, 0, src, dst, 16384, 16384, gx, R);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_copy_strip, dim3(gx * 128), dim3(256), 0, 0, src, dst, 16384, 16384, gx, R);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("{\n \"note\": \"copy of one 16384^2 float32 plane (4 B/px read + 4 B/px written = 2.15 GB), average of 10 launches\",\n"
               " \"dword_per_lane_grid_stride\": {\"TBs\": %.3f, \"ms\": %.3f},\n \"16_bytes_per_lane_grid_stride\": {\"TBs\": %.3f, \"ms\": %.3f},\n"
               " \"dword_per_lane_256_column_strips_of_128_rows\": {\"TBs\": %.3f, \"ms\": %.3f}\n}\n",
               r1, ms1, r4, ms4, (double)n * 8.0 / (ms / 10 * 1e-3) / 1e12, ms / 10);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "write_heavy")) {
        // the config-3 raster: 16384 x 16384 px; 7 planes read, 10 float32 + 1 uint8 planes written
        const int64_t n = (int64_t)16384 * 16384;
        float *buf, *wbuf;
        CHECK(hipMalloc(&buf, (size_t)n * 4 * 7));
        CHECK(hipMalloc(&wbuf, (size_t)n * 4 * 10 + (size_t)n + 4096));
        CHECK(hipMemset(buf, 0, (size_t)n * 4 * 7));
        CHECK(hipMemset(wbuf, 0, (size_t)n * 4 * 10 + (size_t)n));
        double ms;
        printf("{\n \"note\": \"TB/s (and ms) of the fused index + projection pass's bare access pattern at 16384^2: 7 float32 planes read (28 B/px), 10 float32 + 1 uint8 planes written (41 B/px); 69 B/px = 18.5 GB per launch, average of 8 launches\",\n");
#define ROW(name, NR, NW, MAP, NT, grid) { const double r = run_rw<NR, NW, MAP, NT>(buf, wbuf, grid, n, &ms); printf(" \"%s\": {\"TBs\": %.3f, \"ms\": %.3f},\n", name, r, ms); }
        ROW("grid_stride_2048wg", 7, 10, 0, 0, 2048)
        ROW("grid_stride_1024wg", 7, 10, 0, 0, 1024)
        ROW("grid_stride_4096wg", 7, 10, 0, 0, 4096)
        ROW("grid_stride_8192wg", 7, 10, 0, 0, 8192)
        ROW("grid_stride_one_wg_per_4KiB", 7, 10, 0, 0, 65536 * 4)
        ROW("grid_stride_2048wg_nt_store", 7, 10, 0, 1, 2048)
        ROW("grid_stride_2048wg_nt_load_store", 7, 10, 0, 2, 2048)
        ROW("grid_stride_4096wg_nt_load_store", 7, 10, 0, 2, 4096)
        ROW("grid_stride_2048wg_nt_load_plain_store", 7, 10, 0, 3, 2048)
        ROW("grid_stride_8192wg_nt_load_plain_store", 7, 10, 0, 3, 8192)
        ROW("one_wg_per_4KiB_nt_load_plain_store", 7, 10, 0, 3, 65536 * 4)
        ROW("chunked_2048wg", 7, 10, 1, 0, 2048)
        ROW("chunked_16384wg", 7, 10, 1, 0, 16384)
        ROW("chunked_16384wg_nt_load_store", 7, 10, 1, 2, 16384)
        ROW("read_7_write_1", 7, 1, 0, 2, 2048)
        ROW("read_7_write_4", 7, 4, 0, 2, 2048)
        ROW("read_1_write_10", 1, 10, 0, 2, 2048)
        { const double r = run_rw<7, 7, 0, 2>(buf, wbuf, 2048, n, &ms); printf(" \"read_7_write_7\": {\"TBs\": %.3f, \"ms\": %.3f}\n}\n", r, ms); }
        return 0;
    }
    const int64_t total = (int64_t)15 * 268435456;  // 16 GB of float32, what one Lloyd sweep of config 3 reads
    float *buf, *rw, *sink;
    CHECK(hipMalloc(&buf, total * 4));
    CHECK(hipMalloc(&rw, (size_t)total * 4));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, total * 4));
    CHECK(hipMemset(rw, 0, (size_t)total * 4));
    printf("{\n \"note\": \"TB/s of 16-byte-per-lane reads, 16 GB per launch, NS planes requested back to back per lane; rw: plus one read-modify-write plane\",\n");
    printf(" \"read_only\": {\"1\": %.3f, \"2\": %.3f, \"4\": %.3f, \"8\": %.3f, \"15\": %.3f, \"16\": %.3f},\n", run<1, 0>(buf, rw, sink, total),
           run<2, 0>(buf, rw, sink, total), run<4, 0>(buf, rw, sink, total), run<8, 0>(buf, rw, sink, total), run<15, 0>(buf, rw, sink, total),
           run<16, 0>(buf, rw, sink, total));
    printf(" \"with_rw_plane\": {\"1\": %.3f, \"4\": %.3f, \"15\": %.3f},\n", run<1, 1>(buf, rw, sink, total), run<4, 1>(buf, rw, sink, total),
           run<15, 1>(buf, rw, sink, total));
    printf(" \"with_rw_plane_nontemporal_store\": {\"15\": %.3f},\n", run<15, 2>(buf, rw, sink, total));
    printf(" \"with_read_plane_and_separate_write_plane\": {\"15\": %.3f}\n}\n", run<15, 3>(buf, rw, sink, total));
    return 0;
}
