// LDS-DMA issue/landing cost in isolation: 256 blocks, W loader waves per block, each wave issues K
// 1-KiB global_load_lds_dwordx4 back to back (recipe with M0 save/restore, or M0 set once), then vmcnt(0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int K, int MODE>
__global__ __launch_bounds__(1024) void dma_probe(const float4* __restrict__ src, size_t blk_stride4, unsigned* lat) {
    extern __shared__ float4 sm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int W = blockDim.x >> 6;
    const float4* g = src + (size_t)blockIdx.x * blk_stride4 + (size_t)wave * K * 64 + lane;
    const unsigned base = (unsigned)(uintptr_t)sm + (unsigned)(wave * K * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 2) {            // plain register loads, same bytes
        float4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = g[k * 64];
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) a += v[k].x;
        if (a == 1.2345f) sm[0] = v[0];
        if (lane == 0) { lat[(blockIdx.x * W + wave) * 2] = (unsigned)(t1 - t0); lat[(blockIdx.x * W + wave) * 2 + 1] = (unsigned)(t2 - t0); }
        return;
    }
    if (MODE == 1) asm volatile("s_mov_b32 m0, %0" ::"s"(base) : "memory");
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float4* p = g + k * 64;
        if (MODE == 0) {
            unsigned keep;
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + k * 1024);
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
        } else {
            asm volatile("global_load_lds_dwordx4 %0, off offset:0" ::"v"(p) : "memory");   // all into the same 1 KiB (probe only)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { lat[(blockIdx.x * W + wave) * 2] = (unsigned)(t1 - t0); lat[(blockIdx.x * W + wave) * 2 + 1] = (unsigned)(t2 - t0); }
}
template <int K, int MODE> void run(const float4* src, unsigned* lat, int W, const char* name) {
    hipFuncSetAttribute((const void*)&dma_probe<K, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    static float* junk = nullptr;
    if (!junk) hipMalloc(&junk, 512u << 20);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(junk, rep, 512u << 20);          // cold L2 / MALL: the loads go to HBM
        hipDeviceSynchronize();
        hipLaunchKernelGGL((dma_probe<K, MODE>), dim3(256), dim3(W * 64), W * K * 1024, 0, src, (size_t)W * K * 64, lat);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(256 * W * 2);
    hipMemcpy(h.data(), lat, h.size() * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned> a, b;
    for (int i = 0; i < 256 * W; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-28s W=%2d K=%2d (%3d KB/CU): issue median %5u max %5u | landed median %5u max %5u ticks\n", name, W, K, W * K,
           a[a.size() / 2], a.back(), b[b.size() / 2], b.back());
}
int main() {
    float4* src; unsigned* lat;
    hipMalloc(&src, 256u << 20); hipMalloc(&lat, 256 * 16 * 2 * 4);
    hipMemset(src, 0, 256u << 20);
    printf("(caches flushed before every launch)\n");
    run<12, 0>(src, lat, 1, "dma, M0 save/restore");
    run<12, 2>(src, lat, 1, "register loads");
    run<12, 0>(src, lat, 2, "dma, M0 save/restore");
    run<12, 1>(src, lat, 2, "dma, M0 once");
    run<12, 2>(src, lat, 2, "register loads");
    run<24, 0>(src, lat, 2, "dma, M0 save/restore");
    run<24, 1>(src, lat, 2, "dma, M0 once");
    run<12, 0>(src, lat, 4, "dma, M0 save/restore");
    run<12, 2>(src, lat, 4, "register loads");
    run<6, 0>(src, lat, 8, "dma, M0 save/restore");
    run<6, 2>(src, lat, 8, "register loads");
    run<3, 0>(src, lat, 9, "dma, M0 save/restore");
    run<3, 2>(src, lat, 9, "register loads");
    run<8, 2>(src, lat, 9, "register loads");
    run<8, 0>(src, lat, 9, "dma, M0 save/restore");
    run<3, 0>(src, lat, 16, "dma, M0 save/restore");
    run<3, 2>(src, lat, 16, "register loads");
    run<6, 0>(src, lat, 4, "dma, M0 save/restore");
    run<6, 2>(src, lat, 4, "register loads");
    return 0;
}
