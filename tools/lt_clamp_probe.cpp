// Does hipBLASLt fuse bias + clamp(0, 6) (= BiasAdd + Relu6) into an fp32 GEMM epilogue on gfx950, and what does it cost?
// Row-major Y[M,N] = X[M,K] W[K,N] is the column-major product Y^T[N,M] = W^T[N,K] X^T[K,M]: A := W (N x K, lda = N),
// B := X (K x M, ldb = K), D := Y (N x M, ldd = N), bias over D's rows = the N output features.
// Build: hipcc -O2 --offload-arch=gfx950 tools/lt_clamp_probe.cpp -lhipblaslt -o /tmp/lt_probe
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define CK(x) do { auto s_ = (x); if (s_ != 0) { printf("FAIL %s -> %d\n", #x, (int)s_); return 1; } } while (0)

static int run(hipblasLtHandle_t h, int M, int K, int N, bool fuse, float* us_out, float* maxerr) {
    float *X, *W, *Y, *bias; void* ws; size_t wsz = 64 << 20;
    hipMalloc(&X, sizeof(float) * M * K); hipMalloc(&W, sizeof(float) * K * N); hipMalloc(&Y, sizeof(float) * M * N);
    hipMalloc(&bias, sizeof(float) * N); hipMalloc(&ws, wsz);
    std::vector<float> hx(M * K), hw(K * N), hb(N);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = ((i * 2654435761u) % 2001) / 1000.f - 1.f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (((i * 40503u) % 2001) / 1000.f - 1.f) * 0.5f;
    for (int i = 0; i < N; ++i) hb[i] = (i % 7) - 3.f;
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice);
    hipblasLtMatmulDesc_t desc; hipblasLtMatrixLayout_t la, lb, ld;
    CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t opn = HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opn, sizeof(opn)));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opn, sizeof(opn)));
    if (fuse) {
        hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_CLAMP_BIAS_EXT;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
        float lo = 0.f, hi = 6.f;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE_ACT_ARG0_EXT, &lo, sizeof(lo)));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE_ACT_ARG1_EXT, &hi, sizeof(hi)));
    }
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_32F, N, K, N));
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_32F, K, M, K));
    CK(hipblasLtMatrixLayoutCreate(&ld, HIP_R_32F, N, M, N));
    hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
    hipblasLtMatmulHeuristicResult_t res[32]; int n = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, ld, ld, pref, 32, res, &n));
    if (n == 0) { printf("no algorithm (fuse=%d)\n", fuse); return 2; }
    float alpha = 1.f, beta = 0.f, best = 1e9f; int bi = -1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int a = 0; a < n; ++a) {
        for (int w = 0; w < 3; ++w)
            if (hipblasLtMatmul(h, desc, &alpha, W, la, X, lb, &beta, Y, ld, Y, ld, &res[a].algo, ws, wsz, 0) != 0) goto next;
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) hipblasLtMatmul(h, desc, &alpha, W, la, X, lb, &beta, Y, ld, Y, ld, &res[a].algo, ws, wsz, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        { float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 20 < best) { best = ms / 20; bi = a; } }
    next:;
    }
    hipblasLtMatmul(h, desc, &alpha, W, la, X, lb, &beta, Y, ld, Y, ld, &res[bi].algo, ws, wsz, 0);
    std::vector<float> hy((size_t)M * N); hipMemcpy(hy.data(), Y, hy.size() * 4, hipMemcpyDeviceToHost);
    double me = 0;
    for (int m = 0; m < M; m += 397) for (int c = 0; c < N; c += 61) {
        double s = 0; for (int k = 0; k < K; ++k) s += (double)hx[(size_t)m * K + k] * hw[(size_t)k * N + c];
        if (fuse) { s += hb[c]; s = s < 0 ? 0 : (s > 6 ? 6 : s); }
        me = fmax(me, fabs(s - hy[(size_t)m * N + c]));
    }
    *us_out = best * 1e3f; *maxerr = (float)me;
    printf("M=%d K=%d N=%d fuse=%d: %d algos, best %.1f us, max err %.2e\n", M, K, N, fuse, n, best * 1e3f, me);
    return 0;
}
int main() {
    hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
    float us, err;
    run(h, 4096, 197, 2048, false, &us, &err); run(h, 4096, 197, 2048, true, &us, &err);
    run(h, 4096, 1024, 512, false, &us, &err); run(h, 4096, 1024, 512, true, &us, &err);
    return 0;
}
