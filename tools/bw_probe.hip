// HBM read-bandwidth ceilings on this box for the two access patterns of the mixture kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_stream(const float4* __restrict__ p, size_t n4, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i]; s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}
// wave (a,c): rows b = c, c+C, ... ; U rows in flight
template <int U>
__global__ void k_seg(const float* __restrict__ p, int B, int A, int N, int C, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wave >= A * C) return;
    const int a = wave % A, c = wave / A;
    const size_t AN = (size_t)A * N;
    float s = 0.f;
    for (int b0 = c; b0 < B; b0 += C * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int b = b0 + u * C; if (b > B - 1) b = B - 1;
            v[u] = *reinterpret_cast<const float4*>(p + (size_t)b * AN + (size_t)a * N + 4 * lane);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 12345.678f) out[0] = s;
}
// contiguous-chunk variant: wave (a,c) walks rows [c*len, (c+1)*len)
template <int U>
__global__ void k_seg_chunk(const float* __restrict__ p, int B, int A, int N, int C, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wave >= A * C) return;
    const int a = wave % A, c = wave / A;
    const size_t AN = (size_t)A * N;
    const int len = (B + C - 1) / C;
    float s = 0.f;
    for (int b0 = c * len; b0 < min(B, (c + 1) * len); b0 += U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int b = b0 + u; if (b > B - 1) b = B - 1;
            v[u] = *reinterpret_cast<const float4*>(p + (size_t)b * AN + (size_t)a * N + 4 * lane);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 12345.678f) out[0] = s;
}
template <typename F> float timeit(F f, int n = 10) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
int main() {
    const int B = 8192, A = 128, N = 256;
    const size_t n = (size_t)B * A * N;
    float *p, *out; hipMalloc(&p, n * 4); hipMalloc(&out, 4); hipMemset(p, 0, n * 4);
    if (getenv("RANDOM_DATA")) {
        float* h = (float*)malloc(n * 4);
        unsigned s = 12345u;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) * (1.0f / 16777216.0f)) * 4.f - 2.f; }
        hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice);
        free(h);
    }
    const double gb = n * 4 / 1e9;
    for (int grid : {2048, 4096, 8192, 16384}) {
        float us = timeit([&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, (const float4*)p, n / 4, out); });
        printf("stream grid=%5d: %.1f us  %.2f TB/s\n", grid, us, gb / us * 1e3);
    }
    for (int C : {16, 32, 64, 128}) {
        const int blocks = (A * C * 64 + 255) / 256;
        float u1 = timeit([&] { hipLaunchKernelGGL((k_seg<1>), dim3(blocks), dim3(256), 0, 0, p, B, A, N, C, out); });
        float u2 = timeit([&] { hipLaunchKernelGGL((k_seg<2>), dim3(blocks), dim3(256), 0, 0, p, B, A, N, C, out); });
        float u4 = timeit([&] { hipLaunchKernelGGL((k_seg<4>), dim3(blocks), dim3(256), 0, 0, p, B, A, N, C, out); });
        float u8 = timeit([&] { hipLaunchKernelGGL((k_seg<8>), dim3(blocks), dim3(256), 0, 0, p, B, A, N, C, out); });
        float c4 = timeit([&] { hipLaunchKernelGGL((k_seg_chunk<4>), dim3(blocks), dim3(256), 0, 0, p, B, A, N, C, out); });
        printf("seg C=%3d: U1 %.2f  U2 %.2f  U4 %.2f  U8 %.2f TB/s | chunk U4 %.2f TB/s\n", C, gb / u1 * 1e3, gb / u2 * 1e3, gb / u4 * 1e3, gb / u8 * 1e3, gb / c4 * 1e3);
    }
    return 0;
}
