// Latency of the packed kernel's own load pattern, in isolation: 256 blocks x 9 waves, wave w of block b
// loads its 2240-B slice (3 float4 per lane, clamped) of tile b, b+256, ... and reports s_memtime deltas.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(576) void slice_probe(const float* __restrict__ logits, int ntiles, int tile_floats,
                                                   unsigned* __restrict__ lat, float* out, int use_lds) {
    extern __shared__ float4 sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    int it = 0;
    const unsigned long long t00 = __builtin_amdgcn_s_memtime();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const float4* g4 = reinterpret_cast<const float4*>(logits + (size_t)tile * tile_floats + wave * 560);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        float4 v0 = g4[min(lane, 139)], v1 = g4[min(lane + 64, 139)], v2 = g4[min(lane + 128, 139)];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (use_lds) { sm[wave * 140 + min(lane, 139)] = v0; sm[wave * 140 + min(lane + 64, 139)] = v1; }
        acc += v0.x + v1.y + v2.z;
        if (lane == 0) { lat[((size_t)blockIdx.x * 9 + wave) * 8 + it * 2] = (unsigned)(t0 - t00); lat[((size_t)blockIdx.x * 9 + wave) * 8 + it * 2 + 1] = (unsigned)(t1 - t0); }
    }
    if (acc == 1234.5f) out[0] = acc;
}
int main() {
    const int B = 4096, AN = 1260, R = 4, ntiles = B / R, tf = R * AN;
    float *src, *out; unsigned* lat;
    hipMalloc(&src, (size_t)B * AN * 4); hipMalloc(&out, 64); hipMalloc(&lat, 256 * 9 * 8 * 4);
    hipMemset(src, 0, (size_t)B * AN * 4);
    float* junk; hipMalloc(&junk, 256u << 20);
    for (int lds = 0; lds < 2; ++lds) {
        for (int rep = 0; rep < 3; ++rep) {
            if (rep == 2) hipMemset(junk, 1, 256u << 20);      // evict L2/MALL before the last rep
            hipLaunchKernelGGL(slice_probe, dim3(256), dim3(576), lds ? 64 * 1024 : 0, 0, src, ntiles, tf, lat, out, lds);
            hipDeviceSynchronize();
            std::vector<unsigned> h(256 * 9 * 8);
            hipMemcpy(h.data(), lat, h.size() * 4, hipMemcpyDeviceToHost);
            printf("lds=%d rep %d%s\n", lds, rep, rep == 2 ? " (caches flushed)" : "");
            for (int it = 0; it < 4; ++it) {
                std::vector<unsigned> v;
                for (int i = 0; i < 256 * 9; ++i) v.push_back(h[i * 8 + it * 2 + 1]);
                std::sort(v.begin(), v.end());
                printf("   tile %d: issue at %5u (blk0 w0)  latency ticks min %u  median %u  p90 %u  max %u | block 0 waves:", it,
                       h[it * 2], v[0], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
                for (int w = 0; w < 9; ++w) printf(" %u", h[w * 8 + it * 2 + 1]);
                printf("\n");
            }
        }
    }
    return 0;
}
