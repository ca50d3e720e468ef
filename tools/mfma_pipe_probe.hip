// Round 6 probe: what an LDS-fed fp32 MFMA loop reaches on gfx950 when the fragment reads are NOT serialised in front of
// their MFMAs (tools/mfma_split_probe.hip's LDS-fed loops read through `volatile` and measured 84-92 TF-equiv: the number
// the "own GEMMs cannot pay" arithmetic of rounds 3-4 rests on).  Block tile 128 x 128, four waves as 2 x 2, a wave owns
// 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 (v_mfma_f32_32x32x2_f32, 64 accumulator registers); k-tile of 16 = 8 MFMA k-steps,
// a k-step consuming the k pair {kk, kk + 8} so that a lane's eight A values of a k-tile are 32 contiguous bytes:
//   A in LDS as [m][20 floats] (k contiguous, rows padded to 80 B: conflict-free ds_read_b128), 2 reads per fragment row block;
//   B in LDS as [k][128] (n contiguous: ds_read_b32, a lane half per k row).
// No global memory in the loop; one barrier per k-tile as a real double-buffered kernel would have.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_pipe_probe.hip -o /tmp/mpp && /tmp/mpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BARRIER>
__global__ __launch_bounds__(256) void k_pipe(float* out, int ktiles) {
    __shared__ __attribute__((aligned(16))) float A[2][128][20];
    __shared__ __attribute__((aligned(16))) float B[2][16][128];
    const int t = threadIdx.x, w = t >> 6, l = t & 63, wm = w >> 1, wn = w & 1, h = l >> 5, c = l & 31;
    for (int i = t; i < 2 * 128 * 20; i += 256) (&A[0][0][0])[i] = 1e-3f * (i % 7);
    for (int i = t; i < 2 * 16 * 128; i += 256) (&B[0][0][0])[i] = 1e-3f * (i % 5);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < ktiles; ++kt) {
        const int buf = kt & 1;
        f32x4 a[2][2];
        float b[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* p = &A[buf][wm * 64 + i * 32 + c][h * 8];
            a[i][0] = *(const f32x4*)p;
            a[i][1] = *(const f32x4*)(p + 4);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j][kk] = B[buf][kk + 8 * h][wn * 64 + j * 32 + c];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk >> 2][kk & 3], b[j][kk], acc[i][j], 0, 0, 0);
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + t] = s;
}

template <typename F>
static void run(const char* name, F launch, int grid, int ktiles) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)grid * 128.0 * 128.0 * 16.0 * ktiles * 2.0;
    printf("%-40s grid %4d : %.3f ms  %.1f TFLOP/s\n", name, grid, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * sizeof(float));
    const int kt = 4096;
    for (int bpc = 1; bpc <= 3; ++bpc) {
        const int grid = 256 * bpc;
        run("fp32 LDS-fed, barrier per k-tile", [&] { hipLaunchKernelGGL(k_pipe<1>, dim3(grid), dim3(256), 0, 0, out, kt); }, grid, kt);
        run("fp32 LDS-fed, no barrier", [&] { hipLaunchKernelGGL(k_pipe<0>, dim3(grid), dim3(256), 0, 0, out, kt); }, grid, kt);
    }
    return 0;
}
