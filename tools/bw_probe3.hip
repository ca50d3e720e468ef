// The real pfpn_logprob_fwd/bwd on hipMalloc'd buffers, timed with HIP events (no Python, no torch).
#include "../pfpn_amd/csrc/mixture_kernels.hip"
template <typename F> float timeit(F f, int n = 10) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
int main() {
    const int B = 8192, A = 128, N = 256;
    const size_t n = (size_t)B * A * N;
    float *p, *d, *x, *loc, *ls, *lp, *g, *dloc, *dls; void* ws;
    hipMalloc(&p, n * 4); hipMalloc(&d, n * 4); hipMalloc(&x, (size_t)B * A * 4); hipMalloc(&loc, A * N * 4); hipMalloc(&ls, A * N * 4);
    hipMalloc(&lp, B * 4); hipMalloc(&g, B * 4); hipMalloc(&dloc, A * N * 4); hipMalloc(&dls, A * N * 4);
    const size_t wsb = pfpn_mixture_workspace_bytes(B, A, N); hipMalloc(&ws, wsb);
    float* h = (float*)malloc(n * 4);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) * (1.0f / 16777216.0f)) * 4.f - 2.f; }
    hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, h, (size_t)B * A * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < A * N; ++i) h[i] = -1.f + 2.f * (i % N) / (N - 1);
    hipMemcpy(loc, h, A * N * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < A * N; ++i) h[i] = logf(2.f / (N - 1));
    hipMemcpy(ls, h, A * N * 4, hipMemcpyHostToDevice);
    hipMemset(g, 0, B * 4);
    const double gb = n * 4 / 1e9;
    float tf = timeit([&] { pfpn_logprob_fwd(p, x, loc, ls, B, A, N, 0, lp, nullptr, ws, wsb, nullptr); });
    float tb = timeit([&] { pfpn_logprob_bwd(p, x, loc, ls, g, B, A, N, 0, d, dloc, dls, nullptr, ws, wsb, nullptr); });
    printf("C-ABI direct: fwd %.1f us %.2f TB/s | bwd %.1f us %.2f TB/s\n", tf, gb / tf * 1e3, tb, 2 * gb / tb * 1e3);
    return 0;
}
