// How fast can waves issue fp32 MFMAs on gfx950?  Pure register loop, no memory: 4 (or 8/16) independent accumulator
// chains per wave, W waves per SIMD.  Answers whether ONE wave per SIMD can keep the matrix pipe busy.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int CHAINS>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float a0, float b0) {
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    if (SHAPE == 32) {
        f32x16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16 / CHAINS; ++u)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
        float s = 0.f;
        for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x4 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32 / CHAINS; ++u)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        }
        float s = 0.f;
        for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int SHAPE, int CHAINS>
static void run(int blocks_per_cu, float* out) {
    const int iters = 4096;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mfma<SHAPE, CHAINS>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_mfma<SHAPE, CHAINS>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // flops per wave per iteration: 16 x (32*32*2*2) = 32 x (16*16*4*2) = 65536
    const double flops = (double)grid * 4 * iters * 65536.0;
    printf("mfma %dx%d  chains %2d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s\n", SHAPE, SHAPE, CHAINS, blocks_per_cu, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * sizeof(float) * 4);
    for (int w = 1; w <= 4; w *= 2) {
        run<32, 4>(w, out); run<32, 2>(w, out); run<32, 1>(w, out);
        run<16, 16>(w, out); run<16, 8>(w, out); run<16, 4>(w, out);
    }
    return 0;
}
