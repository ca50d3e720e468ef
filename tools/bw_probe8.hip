// LDS-DMA ring for the WIDE mixture shape (8192 x 128 x 256, 1 KiB segments): does a loader wave per block that
// streams the rows of its block's compute waves into per-wave LDS rings beat the register-load pattern of
// k_mix_wide (~5.3 TB/s)?  Consumers do the forward's arithmetic shape (V=1) or a bare sum (V=0).
//   hipcc --offload-arch=gfx950 -O3 tools/bw_probe8.hip -o /tmp/bw8 && /tmp/bw8
#include "../pfpn_amd/csrc/pfpn_common.h"
#include <stdlib.h>
#include <vector>
using namespace pfpn;

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void wait_vmcnt_le(int n) {      // wait until at most n (rounded down to a step of 4) are outstanding
    if (n >= 60) asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
    else if (n >= 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
    else if (n >= 52) asm volatile("s_waitcnt vmcnt(52)" ::: "memory");
    else if (n >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (n >= 44) asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
    else if (n >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    else if (n >= 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
    else if (n >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (n >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ int ldc(const int* c) { return __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// block = 1 loader wave + NCW compute waves; compute wave w of block j is global wave j*NCW + w -> (a, c).
// The loader works in ROUNDS: round k = row number k of every compute wave of the block (NCW DMAs back to back, per-wave
// pointers bumped by the row stride -- no division, no per-item bookkeeping); ONE LDS word counts landed rounds.
template <int V, int NCW, int S, int DR>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ p, const float* __restrict__ x, int B, int A, int N, int C,
                                          float* __restrict__ out) {
    extern __shared__ float4 sm[];
    constexpr int U = 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int* landed = reinterpret_cast<int*>(sm + (size_t)NCW * S * 64);      // rounds landed
    int* used = landed + 1;                                               // [NCW] rows consumed
    if (threadIdx.x < 1 + NCW) landed[threadIdx.x] = 0;
    __syncthreads();
    const size_t AN = (size_t)A * N;
    const int nw = A * C;
    const int maxrows = (B + C - 1) / C;
    if (wave == 0) {
        const unsigned base = (unsigned)(uintptr_t)sm;
        const char* ptr[NCW];
        int lastk[NCW];
#pragma unroll
        for (int w = 0; w < NCW; ++w) {
            const int gw = min(blockIdx.x * NCW + w, nw - 1), a = gw % A, c = gw / A;
            ptr[w] = reinterpret_cast<const char*>(p + (size_t)c * AN + (size_t)a * N + 4 * lane);
            lastk[w] = (B - c + C - 1) / C - 1;
        }
        const size_t rstride = (size_t)C * AN * 4;
        int signalled = 0, slot = 0;
        for (int k = 0; k < maxrows; ++k) {
            // ring slot k % S free in every wave?
            for (;;) {
                const int u_ = (lane < NCW) ? ldc(&used[lane]) : 0x7fffffff;
                if (__ballot(u_ < k - S + 1) == 0) break;
                if (k > signalled) {                       // retire the oldest round meanwhile
                    wait_vmcnt_le((k - signalled - 1) * NCW);
                    ++signalled;
                    if (lane == 0) __hip_atomic_store(landed, signalled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int w = 0; w < NCW; ++w) {
                dma16(ptr[w], __builtin_amdgcn_readfirstlane(base + (unsigned)((w * S + slot) * 1024)));
                if (k < lastk[w]) ptr[w] += rstride;       // short waves re-read their last row
            }
            if (++slot == S) slot = 0;
            if (k + 1 - signalled >= DR) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DR - 1) * NCW) : "memory");
                ++signalled;
                if (lane == 0) __hip_atomic_store(landed, signalled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(landed, maxrows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    const int w = wave - 1;
    const int gw = blockIdx.x * NCW + w;
    const bool real = gw < nw;
    const int a = real ? gw % A : 0, c = real ? gw / A : 0;
    const int nrows = real ? (B - c + C - 1) / C : 0;
    const float4* ring = sm + (size_t)w * S * 64;
    float4 bl = make_float4(0.1f * lane, 0.2f, 0.3f, 0.4f), bi = make_float4(3.f, 3.f, 3.f, 3.f), bc = make_float4(60.f, 60.f, 60.f, 60.f);
    float acc = 0.f;
    float xn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xn[u] = (V >= 1) ? x[(size_t)min(c + u * C, B - 1) * A + a] : 0.f;
    for (int k0 = 0; k0 < maxrows; k0 += U) {
        const int need = min(k0 + U, maxrows);
        while (ldc(landed) < need) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        float4 l[U]; float xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            l[u] = ring[((k0 + u) & (S - 1)) * 64 + lane];
            xv[u] = xn[u];
            const int b = min(c + (k0 + U + u) * C, B - 1);      // the NEXT group's action values (one group ahead)
            xn[u] = (V >= 1) ? x[(size_t)b * A + a] : 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&used[w], need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k0 >= nrows) continue;
        if (V >= 2) {                                   // ring copy: V=2 plain stores, V=3 non-temporal
            typedef float f4v __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = c + (k0 + u) * C;
                if (b < B) {
                    f4v* dst = reinterpret_cast<f4v*>(out + (size_t)b * AN + (size_t)a * N) + lane;
                    const f4v v = {l[u].x, l[u].y, l[u].z, l[u].w};
                    if (V == 3) __builtin_nontemporal_store(v, dst); else *dst = v;
                }
            }
            continue;
        }
        float mx[U], sum[U], num[U];
#pragma unroll
        for (int u = 0; u < U; ++u) mx[u] = fmaxf(fmaxf(l[u].x, l[u].y), fmaxf(l[u].z, l[u].w));
        if (V >= 1) wave_reduce4<true>(mx[0], mx[1], mx[2], mx[3], lane);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (V >= 1) {
                const float mxl = mx[u] * PFPN_LOG2E;
                float s = 0.f, n = 0.f;
#define EL(c_) { const float le = fmaf(l[u].c_, PFPN_LOG2E, -mxl); const float zq = (xv[u] - bl.c_) * bi.c_; \
                 const float e = __builtin_amdgcn_exp2f(le); const float ef = __builtin_amdgcn_exp2f(le + fmaf(-zq, zq, bc.c_)); s += e; n += ef; }
                EL(x) EL(y) EL(z) EL(w)
#undef EL
                sum[u] = s; num[u] = n;
            } else { sum[u] = l[u].x + l[u].y + l[u].z + l[u].w; num[u] = sum[u]; }
        }
        if (V >= 1) { wave_reduce4<false>(sum[0], sum[1], sum[2], sum[3], lane); wave_reduce4<false>(num[0], num[1], num[2], num[3], lane); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pz = num[u] * __builtin_amdgcn_rcpf(sum[u] + 1e-30f);
            const int b = c + (k0 + u) * C;
            if (V >= 1) { if (b < B) out[(size_t)b * A + a] = __builtin_amdgcn_logf(pz); }
            else acc += pz;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
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
    float *p, *x, *out; hipMalloc(&p, n * 4); hipMalloc(&x, (size_t)B * A * 4); hipMalloc(&out, n * 4);
    {
        std::vector<float> h(n);
        unsigned s = 12345u;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) * (1.0f / 16777216.0f)) * 4.f - 2.f; }
        hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
    }
    hipMemset(x, 0, (size_t)B * A * 4);
    const double bytes = (double)n * 4;
#define RUN(V_, NCW_, S_, DR_, C_)                                                                                     \
    do {                                                                                                               \
        hipFuncSetAttribute((const void*)&k<V_, NCW_, S_, DR_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
        const int C = C_;                                                                                              \
        const int blocks = (A * C + NCW_ - 1) / NCW_;                                                                  \
        const size_t lds = (size_t)NCW_ * S_ * 1024 + (1 + NCW_) * 4 + 64;                                             \
        const float us = timeit([&]() { hipLaunchKernelGGL((k<V_, NCW_, S_, DR_>), dim3(blocks), dim3((1 + NCW_) * 64), lds, 0, p, x, B, A, N, C, out); }); \
        printf("V=%d NCW=%2d S=%2d DR=%d C=%2d blocks=%4d lds=%6zu : %7.1f us  %.2f TB/s  (%.3f of 8)\n", V_, NCW_, S_, DR_, C, blocks, \
               lds, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);                                                      \
        fflush(stdout);                                                                                                \
    } while (0)
    RUN(0, 8, 8, 6, 32); RUN(1, 8, 8, 4, 48);
    RUN(2, 8, 8, 6, 32); RUN(3, 8, 8, 6, 32);
    RUN(2, 8, 16, 4, 16); RUN(3, 8, 16, 4, 16);
    RUN(2, 12, 8, 4, 24); RUN(3, 12, 8, 4, 24);
    RUN(2, 8, 8, 4, 48); RUN(3, 8, 8, 4, 48);
    RUN(2, 15, 8, 4, 30); RUN(3, 15, 8, 4, 30);
    RUN(2, 4, 16, 8, 32); RUN(3, 4, 16, 8, 32);
    {
        const float us = timeit([&]() { hipMemcpyAsync(out, p, n * 4, hipMemcpyDeviceToDevice, 0); });
        printf("hipMemcpy D2D: %7.1f us  %.2f TB/s read+write\n", us, 2 * bytes / us / 1e6);
    }
    return 0;
}
