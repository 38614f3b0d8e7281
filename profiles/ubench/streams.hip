// Read bandwidth of the k-means access pattern: a workgroup walks a chunk of 16 tiles x 1024 pixels; per tile every lane
// requests one 16-byte vector from each of NS planes back to back (NS x 1 KiB in flight per wave), as km_lloyd / km_kpp do.
// Total bytes are the same for every NS (the planes shrink as NS grows), so the numbers compare the access pattern only.
// RW = 1 adds a read-modify-write plane (k-means++'s closest-distance plane).
// Build: hipcc -O3 --offload-arch=gfx950 -o streams streams.hip     Run: ./streams > streams.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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

int main()
{
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
