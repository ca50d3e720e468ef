// Round-4 probe: the issue-rate CEILING of an exact three-way bf16 split product against the fp32 MFMA on gfx950.
// One k-step of 16 on a 32 x 32 output tile is  8 x v_mfma_f32_32x32x2_f32   (fp32 operands)             = 8 x 64 cycles
//                                           or  9 x v_mfma_f32_32x32x16_bf16 (3 A planes x 3 B planes)    = 9 x 32 cycles.
// (a) pure register loops; (b) a wave owning a 64 x 64 tile whose fragments come from LDS every k-step (3 + 3 planes x 2
// fragments = 12 ds_read_b128 feeding 36 MFMAs; fp32: 4 fragment sets per k = 2).  No global memory: what a hand-written
// 3-plane kernel could reach at best.  "TF-equiv" counts 2 x 32 x 32 x 16 flop per k-step per tile, whatever it took.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_split_probe.hip -o /tmp/msp && /tmp/msp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_fp32_reg(float* out, int ksteps, float a0) {
    float a = a0 + threadIdx.x * 1e-6f, b = 1.0f;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int it = 0; it < ksteps; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_split_reg(float* out, int ksteps, float a0) {
    bf16x8 a[3], b[3];
    for (int p = 0; p < 3; ++p)
        for (int j = 0; j < 8; ++j) { a[p][j] = (__bf16)(a0 + p + threadIdx.x * 1e-3f); b[p][j] = (__bf16)(1.0f / (1 + p)); }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int it = 0; it < ksteps; ++it) {
#pragma unroll
        for (int i = 2; i >= 0; --i)
#pragma unroll
            for (int j = 2; j >= 0; --j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// (b) LDS-fed, a wave owns 64 x 64 (2 x 2 tiles of 32 x 32).  LDS holds ONE k-step's fragments, re-read every step (the
// read traffic and its waits are what is being measured; the values do not matter).
__global__ __launch_bounds__(256) void k_split_lds(float* out, int ksteps) {
    __shared__ bf16x8 frag[4][12][64];                    // [wave][plane-fragment][lane]: 16 B per lane, conflict-free b128 reads
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int f = 0; f < 12; ++f)
        for (int j = 0; j < 8; ++j) frag[w][f][l][j] = (__bf16)(0.001f * (f + 1) + l * 1e-4f);
    __syncthreads();
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < ksteps; ++it) {
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[t][p] = *(volatile bf16x8*)&frag[w][t * 3 + p][l];
                b[t][p] = *(volatile bf16x8*)&frag[w][6 + t * 3 + p][l];
            }
#pragma unroll
        for (int i = 2; i >= 0; --i)
#pragma unroll
            for (int j = 2; j >= 0; --j)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t >> 1][i], b[t & 1][j], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fp32_lds(float* out, int ksteps) {
    __shared__ float frag[4][8][4][64];                   // [wave][k pair of the k-step][fragment][lane]
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int u = 0; u < 8; ++u) for (int f = 0; f < 4; ++f) frag[w][u][f][l] = 0.001f * (f + 1) + l * 1e-4f;
    __syncthreads();
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < ksteps; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a0 = *(volatile float*)&frag[w][u][0][l], a1 = *(volatile float*)&frag[w][u][1][l];
            float b0 = *(volatile float*)&frag[w][u][2][l], b1 = *(volatile float*)&frag[w][u][3][l];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, int grid, int ksteps, double tiles_per_wave) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)grid * 4 * ksteps * tiles_per_wave * 2.0 * 32 * 32 * 16;
    printf("%-44s grid %4d : %.3f ms  %.1f TF-equiv\n", name, grid, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * sizeof(float) * 4);
    const int ks = 8192;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = 256 * bpc;
        run("fp32 32x32x2 x8 per k16, registers", [&] { hipLaunchKernelGGL(k_fp32_reg, dim3(grid), dim3(256), 0, 0, out, ks, 1.0f); }, grid, ks, 1);
        run("bf16 3x3 planes x9 per k16, registers", [&] { hipLaunchKernelGGL(k_split_reg, dim3(grid), dim3(256), 0, 0, out, ks, 1.0f); }, grid, ks, 1);
        run("fp32 64x64 per wave, fragments from LDS", [&] { hipLaunchKernelGGL(k_fp32_lds, dim3(grid), dim3(256), 0, 0, out, ks / 4); }, grid, ks / 4, 4);
        run("bf16 3x3 64x64 per wave, fragments from LDS", [&] { hipLaunchKernelGGL(k_split_lds, dim3(grid), dim3(256), 0, 0, out, ks / 4); }, grid, ks / 4, 4);
    }
    return 0;
}
