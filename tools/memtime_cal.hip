#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = 1.f + threadIdx.x * 1e-6f, b = 1.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}
int main() {
    unsigned long long* o; float* sink; hipMalloc(&o, 64); hipMalloc(&sink, 256 * 256 * 4);
    for (int iters : {256, 4096}) {
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, o, sink, iters);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
        printf("iters %d: %d MFMA 32x32x2 per wave: s_memtime %llu ticks (%.2f per MFMA), wall_clock64 %llu ticks -> s_memtime rate = %.1f x wall clock\n",
               iters, iters * 16, h[0], (double)h[0] / (iters * 16), h[1], (double)h[0] / h[1]);
    }
}
