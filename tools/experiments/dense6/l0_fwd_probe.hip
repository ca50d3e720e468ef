// Round 6 probe: layer 0 of the merged trunks as ONE hand-written launch,
//   y = relu6(x [M, K] . w [K, N] + bias)  (+ the Relu6Grad mask bits of k_bias_act_mask),  M = 4096, K = 197, N = 2048,
// against what it would replace: the library GEMM (28.7 us) + the epilogue pass (12.3 us).  fp32 MFMA 32x32x2, block tile
// 128 x 128, k-tile 16, four waves as 2 x 2 (64 x 64 per wave), double-buffered LDS, global loads of tile kt + 1 issued in front
// of tile kt's MFMAs (tools/mfma_pipe_probe.hip: this loop shape reaches 150 TFLOP/s LDS-fed at two blocks per CU).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/experiments/dense6/l0_fwd_probe.hip -o /tmp/l0f && /tmp/l0f
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#ifndef VARIANT
#define VARIANT 0
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16, APAD = 20;

__global__ __launch_bounds__(256, 2) void k_dense_fwd(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int M, int K, int N,
                                                      float* __restrict__ y, unsigned* __restrict__ mask32) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * APAD + 2 * BK * BN];        // 36,864 B
    float (*As)[BM][APAD] = reinterpret_cast<float (*)[BM][APAD]>(smem);
    float (*Bs)[BK][BN] = reinterpret_cast<float (*)[BK][BN]>(smem + 2 * BM * APAD);
    const int t = threadIdx.x, wv = t >> 6, l = t & 63, wm = wv >> 1, wn = wv & 1, h = l >> 5, c = l & 31;
    // XCD-aware tile order: blocks land on XCD (blockIdx % 8); each XCD works a contiguous range of tiles (same x rows)
    const int ntn = N / BN, ntiles = (M / BM) * ntn;
    const int per = ntiles / 8;
    const int tile = (ntiles % 8 == 0) ? (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int nk = (K + BK - 1) / BK;

    const int ar = t >> 1, ak = (t & 1) * 8;               // A: row ar, 8 consecutive k
    const int bk = t >> 4, bn = (t & 15) * 8;              // B: k row bk, 8 consecutive n
    const float* xa = x + (size_t)(m0 + ar) * ldx + ak;
    const float* wb = w + (size_t)bk * N + n0 + bn;
    struct Regs { float a[8]; f32x4 b[2]; };
    auto gload = [&](int kt, Regs& R) {
        const int k0 = kt * BK;
        {
            const bool kin = k0 + (t & 15) < K;
            const float* p = x + (size_t)(m0 + (t >> 4)) * ldx + k0 + (t & 15);
#pragma unroll
            for (int j = 0; j < 8; ++j) R.a[j] = kin ? p[(size_t)16 * j * ldx] : 0.f;
        }
        if (k0 + bk < K) {
            const float* p = wb + (size_t)k0 * N;
            R.b[0] = *reinterpret_cast<const f32x4*>(p);
            R.b[1] = *reinterpret_cast<const f32x4*>(p + 4);
        } else {
            R.b[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            R.b[1] = R.b[0];
        }
    };
    auto lstore = [&](int buf, const Regs& R) {
#pragma unroll
        for (int j = 0; j < 8; ++j) As[buf][(t >> 4) + 16 * j][t & 15] = R.a[j];
        *reinterpret_cast<f32x4*>(&Bs[buf][bk][bn]) = R.b[0];
        *reinterpret_cast<f32x4*>(&Bs[buf][bk][bn + 4]) = R.b[1];
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](int buf) {
        f32x4 a[2][2];
        float b[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* p = &As[buf][wm * 64 + i * 32 + c][h * 8];
            a[i][0] = *reinterpret_cast<const f32x4*>(p);
            a[i][1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j][kk] = Bs[buf][kk + 8 * h][wn * 64 + j * 32 + c];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk >> 2][kk & 3], b[j][kk], acc[i][j], 0, 0, 0);
    };
    Regs R0, R1;
    // tile kt + 2's global loads are in flight while tile kt computes: registers R0 (even tiles) / R1 (odd tiles)
    gload(0, R0);
    lstore(0, R0);
    if (nk > 1) gload(1, R1);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 2 < nk) gload(kt + 2, R0);
        compute(0);
        if (kt + 1 < nk) lstore(1, R1);
        __syncthreads();
        if (kt + 1 < nk) {
            if (kt + 3 < nk) gload(kt + 3, R1);
            compute(1);
            if (kt + 2 < nk) lstore(0, R0);
            __syncthreads();
        }
    }
    // epilogue: bias + ReLU6 through an LDS stage of 64 rows x 128 columns (two halves), then whole-row float4 stores and the
    // mask bits in k_bias_act_mask's layout: float4 i = row * N/4 + col/4, bit (i & 63) of word (i >> 6) * 4 + component
#if VARIANT == 2
    {
        float sacc = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        y[(size_t)(m0 + (t >> 1)) * N + n0 + (t & 1)] = sacc;
        return;
    }
#endif
#if VARIANT >= 5
    {
        // straight from the accumulators: a register's 32 lanes are 128 contiguous, line-aligned bytes of one row; the bits of a
        // row's 32 columns are one half of a ballot (plain row-major bit order: word32 (row * N + col) / 32)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + c;
            const float bj = bias[col];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    const float v = fminf(fmaxf(acc[i][j][r] + bj, 0.f), 6.f);
                    y[(size_t)row * N + col] = v;
                    const unsigned long long bb = __ballot(v > 0.f && v < 6.f);
                    if (c == 0) mask32[((size_t)row * N + col) >> 5] = (unsigned)(bb >> (32 * h));
                }
        }
        return;
    }
#endif
    float (*stage)[BN] = reinterpret_cast<float (*)[BN]>(smem);
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[j] = bias[n0 + wn * 64 + j * 32 + c];
    for (int hh = 0; hh < 2; ++hh) {
        if (wm == hh) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                        stage[row][wn * 64 + j * 32 + c] = fminf(fmaxf(acc[i][j][r] + bv[j], 0.f), 6.f);
                    }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int rl = (t >> 5) + 8 * p;
            const int row = m0 + hh * 64 + rl;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&stage[rl][c * 4]);
#if VARIANT == 1
            if (v[0] == 123.f)
#endif
            *reinterpret_cast<f32x4*>(y + (size_t)row * N + n0 + c * 4) = v;
            if (mask32 && VARIANT != 1) {
                const unsigned long long b0 = __ballot(v[0] > 0.f && v[0] < 6.f), b1 = __ballot(v[1] > 0.f && v[1] < 6.f);
                const unsigned long long b2 = __ballot(v[2] > 0.f && v[2] < 6.f), b3 = __ballot(v[3] > 0.f && v[3] < 6.f);
                if (c == 0) {
                    const size_t w64 = (((size_t)row * N + n0) >> 8) * 4;
                    unsigned* mw = mask32 + w64 * 2 + ((n0 >> 7) & 1);
                    const int sh = h * 32;
                    mw[0] = (unsigned)(b0 >> sh); mw[2] = (unsigned)(b1 >> sh);
                    mw[4] = (unsigned)(b2 >> sh); mw[6] = (unsigned)(b3 >> sh);
                }
            }
        }
        __syncthreads();
    }
}

int main() {
#ifndef KDIM
#define KDIM 197
#endif
    const int M = 4096, K = KDIM, N = 2048;
    std::vector<float> hx((size_t)M * K), hw((size_t)K * N), hb(N);
    srand(1);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 6.f;
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    float *x, *w, *b, *y; unsigned* mk;
    const size_t mask_words64 = (size_t)M * N / 256 * 4;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&b, hb.size() * 4);
    hipMalloc(&y, (size_t)M * N * 4); hipMalloc(&mk, mask_words64 * 8);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemset(y, 0xff, (size_t)M * N * 4); hipMemset(mk, 0xaa, mask_words64 * 8);
    const int grid = (M / BM) * (N / BN);
    hipLaunchKernelGGL(k_dense_fwd, dim3(grid), dim3(256), 0, 0, x, K, w, b, M, K, N, y, mk);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<float> hy((size_t)M * N); std::vector<unsigned long long> hm(mask_words64);
    hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hm.data(), mk, hm.size() * 8, hipMemcpyDeviceToHost);
    double maxerr = 0; long badbits = 0, checked = 0, nearkink = 0;
    for (int r = 0; r < M; r += (KDIM > 500 ? 997 : 37)) {
        for (int col = 0; col < N; ++col) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hx[(size_t)r * K + k] * hw[(size_t)k * N + col];
            const double pre = s + hb[col];
            const double ref = fmin(fmax(pre, 0.0), 6.0);
            maxerr = fmax(maxerr, fabs(ref - hy[(size_t)r * N + col]));
            const size_t i4 = ((size_t)r * N + col) / 4;
            const int bit = (int)((hm[(i4 >> 6) * 4 + (col & 3)] >> (i4 & 63)) & 1);
            const float got = hy[(size_t)r * N + col];
            if (VARIANT < 5 && bit != (got > 0.f && got < 6.f)) ++badbits;                      // bits must agree with the stored activation
            if (fabs(pre) < 1e-4 || fabs(pre - 6) < 1e-4) ++nearkink;
            ++checked;
        }
    }
    printf("checked %ld outputs: max |err| %.3e, mask bits inconsistent with y: %ld (near a kink: %ld)\n", checked, maxerr, badbits, nearkink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_dense_fwd, dim3(grid), dim3(256), 0, 0, x, K, w, b, M, K, N, y, mk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("fused L0 forward (GEMM + bias + ReLU6 + mask bits): %.2f us per launch = %.1f TFLOP/s\n", ms / 50 * 1e3,
               2.0 * M * K * N / (ms / 50 * 1e-3) / 1e12);
    }
    return 0;
}
