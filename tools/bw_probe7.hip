// Does LDS polling by other waves slow a loader wave's LDS-DMA issue?  2 loader waves x 12 DMAs, plus S
// waves spinning on an LDS counter (s_sleep N between polls) until the loaders are done.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int K>
__global__ __launch_bounds__(1024) void k(const float4* __restrict__ src, unsigned* lat, int nload, int sleepn, int mode) {
    extern __shared__ float4 sm[];
    __shared__ int flag;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    if (wave < nload) {
        const float4* g = src + (size_t)blockIdx.x * nload * K * 64 + (size_t)wave * K * 64 + lane;
        const unsigned base = (unsigned)(uintptr_t)sm + (unsigned)(wave * K * 1024);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            unsigned keep;
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + kk * 1024);
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g + kk * 64), "s"(dst) : "memory");
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { lat[(blockIdx.x * nload + wave) * 2] = (unsigned)(t1 - t0); lat[(blockIdx.x * nload + wave) * 2 + 1] = (unsigned)(t2 - t0); atomicAdd(&flag, 1); }
    } else {
        if (mode == 0) {               // LDS polling
            while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nload) {
                if (sleepn == 1) __builtin_amdgcn_s_sleep(1); else if (sleepn == 8) __builtin_amdgcn_s_sleep(8); else if (sleepn == 32) __builtin_amdgcn_s_sleep(32);
            }
        } else {                        // VALU busy loop (no LDS), fixed length
            float a = (float)lane;
            for (int i = 0; i < 3000; ++i) a = __builtin_fmaf(a, 1.0001f, 0.5f);
            if (a == 1.234f) sm[0].x = a;
        }
    }
}
void run(const float4* src, unsigned* lat, int spin, int sleepn, int mode, const char* name) {
    const int K = 12, nload = 2;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<K>), dim3(256), dim3((nload + spin) * 64), nload * K * 1024, 0, src, lat, nload, sleepn, mode);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(256 * nload * 2);
    hipMemcpy(h.data(), lat, h.size() * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned> a, b;
    for (int i = 0; i < 256 * nload; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-34s: issue 12 DMAs median %5u max %5u | landed median %5u max %5u ticks\n", name, a[a.size() / 2], a.back(), b[b.size() / 2], b.back());
}
int main() {
    float4* src; unsigned* lat;
    hipMalloc(&src, 64u << 20); hipMalloc(&lat, 256 * 4 * 2 * 4);
    hipMemset(src, 0, 64u << 20);
    run(src, lat, 0, 0, 0, "no other waves");
    run(src, lat, 9, 1, 0, "9 waves polling LDS, s_sleep 1");
    run(src, lat, 9, 8, 0, "9 waves polling LDS, s_sleep 8");
    run(src, lat, 9, 32, 0, "9 waves polling LDS, s_sleep 32");
    run(src, lat, 9, 0, 0, "9 waves polling LDS, no sleep");
    run(src, lat, 9, 0, 1, "9 waves VALU busy");
    return 0;
}
