// How fast can 256 CUs pull a 20 MB array once (the config-2 logits), as a function of how the requests
// are spread over waves?  hipcc --offload-arch=gfx950 -O3 tools/bw_probe4.hip -o /tmp/bw4 && /tmp/bw4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int U>
__global__ void rd(const float4* __restrict__ src, size_t n4, float* out) {
    // block b reads float4 [b*chunk, (b+1)*chunk) ; every wave issues U loads before using any
    const size_t chunk = n4 / gridDim.x;
    const float4* p = src + (size_t)blockIdx.x * chunk;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < chunk; i += (size_t)blockDim.x * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t j = i + (size_t)u * blockDim.x; v[u] = p[j < chunk ? j : i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int U>
__global__ void rw(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t chunk = n4 / gridDim.x;
    const float4* p = src + (size_t)blockIdx.x * chunk;
    float4* d = dst + (size_t)blockIdx.x * chunk;
    for (size_t i = threadIdx.x; i < chunk; i += (size_t)blockDim.x * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t j = i + (size_t)u * blockDim.x; v[u] = p[j < chunk ? j : i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t j = i + (size_t)u * blockDim.x; if (j < chunk) { v[u].x += 1.f; d[j] = v[u]; } }
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}
int main() {
    const size_t bytes = (size_t)4096 * 36 * 35 * 4, n4 = bytes / 16;
    float4 *src, *dst; float* out;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&out, 64);
    hipMemset(src, 0, bytes);
    printf("array %.1f MB\n", bytes / 1e6);
#define RUN(GRID, THREADS, UU)                                                                                     \
    {                                                                                                              \
        float us = timeit([&] { hipLaunchKernelGGL(rd<UU>, dim3(GRID), dim3(THREADS), 0, 0, src, n4, out); }, 50);  \
        float us2 = timeit([&] { hipLaunchKernelGGL(rw<UU>, dim3(GRID), dim3(THREADS), 0, 0, src, dst, n4); }, 50); \
        printf("grid %5d x %4d thr  U=%d : read %.1f us (%.2f TB/s)   read+write %.1f us (%.2f TB/s)\n", GRID,     \
               THREADS, UU, us, bytes / us * 1e-6, us2, 2 * bytes / us2 * 1e-6);                                   \
    }
    RUN(256, 576, 1) RUN(256, 576, 2) RUN(256, 576, 4) RUN(256, 576, 8)
    RUN(256, 1024, 1) RUN(256, 1024, 2) RUN(256, 1024, 4) RUN(256, 1024, 8)
    RUN(512, 512, 2) RUN(512, 512, 4) RUN(1024, 256, 2) RUN(1024, 256, 4) RUN(1024, 256, 8) RUN(2048, 256, 4)
    RUN(4096, 256, 2) RUN(4096, 256, 4)
    return 0;
}
