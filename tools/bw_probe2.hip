// Which ingredient of the wide mixture kernel costs bandwidth?  Same access pattern, compute added stepwise.
#include "../pfpn_amd/csrc/pfpn_common.h"
#include <stdlib.h>
using namespace pfpn;
template <int V, bool PF>
__global__ __launch_bounds__(256) void k(const float* __restrict__ p, const float* __restrict__ x, int B, int A, int N, int C, float* __restrict__ out) {
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wave >= A * C) return;
    const int a = wave % A, c = wave / A;
    const size_t AN = (size_t)A * N;
    float4 bl = make_float4(0.1f * lane, 0.2f, 0.3f, 0.4f), bi = make_float4(3.f, 3.f, 3.f, 3.f), bc = make_float4(60.f, 60.f, 60.f, 60.f);
    float acc = 0.f;
    float4 ln[U]; float xn[U];
    auto load = [&](int b0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = min(b0 + u * C, B - 1);
            ln[u] = *reinterpret_cast<const float4*>(p + (size_t)b * AN + (size_t)a * N + 4 * lane);
            xn[u] = (V >= 1) ? x[(size_t)b * A + a] : 0.f;
        }
    };
    const int step = C * U;
    int b0 = c;
    if (PF) load(b0);
    for (; b0 + (U - 1) * C < B; b0 += step) {
        float4 l[U]; float xv[U];
        if (PF) {
#pragma unroll
            for (int u = 0; u < U; ++u) { l[u] = ln[u]; xv[u] = xn[u]; }
            load(b0 + step);
        } else {
            load(b0);
#pragma unroll
            for (int u = 0; u < U; ++u) { l[u] = ln[u]; xv[u] = xn[u]; }
        }
        float mx[U], sum[U], num[U];
#pragma unroll
        for (int u = 0; u < U; ++u) mx[u] = fmaxf(fmaxf(l[u].x, l[u].y), fmaxf(l[u].z, l[u].w));
        if (V >= 2) wave_reduce4<true>(mx[0], mx[1], mx[2], mx[3], lane);
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
        if (V >= 2) { wave_reduce4<false>(sum[0], sum[1], sum[2], sum[3], lane); wave_reduce4<false>(num[0], num[1], num[2], num[3], lane); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pz = num[u] * __builtin_amdgcn_rcpf(sum[u] + 1e-30f);
            if (V >= 3) out[(size_t)(b0 + u * C) * A + a] = __builtin_amdgcn_logf(pz);
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
    float *p, *x, *out; hipMalloc(&p, n * 4); hipMalloc(&x, (size_t)B * A * 4); hipMalloc(&out, (size_t)B * A * 4);
    hipMemset(p, 0, n * 4); hipMemset(x, 0, (size_t)B * A * 4);
    if (getenv("RANDOM_DATA")) {      // random fp32 in [-2, 2): zero-filled buffers flatter the memory system
        float* h = (float*)malloc(n * 4);
        unsigned s = 12345u;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) * (1.0f / 16777216.0f)) * 4.f - 2.f; }
        hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice);
        free(h);
    }
    const double gb = n * 4 / 1e9;
    if (getenv("ONLY_V3")) {
        const int C = 32, blocks = (A * C * 64 + 255) / 256;
        float us = timeit([&] { hipLaunchKernelGGL((k<3, true>), dim3(blocks), dim3(256), 0, 0, p, x, B, A, N, C, out); });
        printf("V3 PF C=32: %.1f us %.2f TB/s\n", us, gb / us * 1e3);
        return 0;
    }
    for (int C : {8, 16, 32, 64}) {
        const int blocks = (A * C * 64 + 255) / 256;
#define RUN(V, PF) timeit([&] { hipLaunchKernelGGL((k<V, PF>), dim3(blocks), dim3(256), 0, 0, p, x, B, A, N, C, out); })
        printf("C=%2d  noPF: V0 %.2f V1 %.2f V2 %.2f V3 %.2f | PF: V0 %.2f V1 %.2f V2 %.2f V3 %.2f  TB/s\n", C,
               gb / RUN(0, false) * 1e3, gb / RUN(1, false) * 1e3, gb / RUN(2, false) * 1e3, gb / RUN(3, false) * 1e3,
               gb / RUN(0, true) * 1e3, gb / RUN(1, true) * 1e3, gb / RUN(2, true) * 1e3, gb / RUN(3, true) * 1e3);
    }
    return 0;
}
