// What v_permlane32_swap / v_permlane16_swap (gfx950) do to two registers, lane by lane, through the clang builtins.
// hipcc --offload-arch=gfx950 tools/permlane_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    const unsigned a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 ret[0]", "swap32 ret[1]", "swap16 ret[0]", "swap16 ret[1]"};
    for (int v = 0; v < 4; ++v) {
        printf("%s:", names[v]);
        for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * v + l]);
        printf("\n");
    }
    return 0;
}
