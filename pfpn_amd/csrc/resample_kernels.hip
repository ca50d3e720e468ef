// K5: dead-particle resampling for gfx950.
//
// Replaces ParticleFilteringA2CNetwork.build_resample_ops (reference networks/actor_critic/a2c.py:385-474), both
// source-choice branches: `resample == -1` (categorical draw over the average activation, :400-404; what every
// shipped setting uses) and `resample == k > 0` (uniform choice among the k most active particles, :405-408);
// ~370 TF graph nodes: Where, Multinomial / TopKV2, GatherNd,
// UniqueWithCounts + map_fn, ScatterNdUpdate/ScatterUpdate).  Runs once every
// `resample_interval` learner steps: latency-only, but all index arithmetic is bit-exact:
//   * avg = sum_active / rowsum (fp32, left-to-right sum), log_avg = fl32(log_f64(avg));
//   * categorical draw = TF-1.14 CPU Multinomial rule on log_avg with float64 running sums
//     (sequential in one lane, identical rounding order to the oracle);
//   * top-k branch: the k largest averages, ties to the lower index (TF TopKV2's rule; the reference asks for
//     sorted=False, whose order is unspecified -- the draw over the set is uniform, so any order is the same
//     distribution), candidate number min(k-1, floor(u*k));
//   * counts/deltas in integers; the two bias scatters in the reference's order.
// One block per action dimension updates loc/logstd/bias and records (dead, src); a second
// kernel copies weight columns row by row through LDS (gather-before-scatter semantics:
// every clone reads the OLD source column even when the source is itself dead).
#include "pfpn_common.h"

namespace pfpn {

constexpr int RS_MAXN = 1024;

__global__ __launch_bounds__(256) void k_resample_dims(
    float* __restrict__ loc, float* __restrict__ logstd, float* __restrict__ bias,
    const float* __restrict__ max_active, const float* __restrict__ sum_active,
    const double* __restrict__ u_resample, const float* __restrict__ u_noise,
    int A, int N, float threshold, int tanh_mode, int topk, float fixed_sigma,
    int* __restrict__ dead_out, int* __restrict__ src_out)
{
    __shared__ double cdf[RS_MAXN];
    __shared__ float o_loc[RS_MAXN], o_ls[RS_MAXN], o_bias[RS_MAXN], nb[RS_MAXN];
    __shared__ int cnt[RS_MAXN], dead[RS_MAXN], src[RS_MAXN];
    __shared__ double total_s;
    const int a = blockIdx.x;
    const int t = threadIdx.x;
    const float* sa = sum_active + a * N;
    for (int n = t; n < N; n += blockDim.x) {
        o_loc[n] = loc[a * N + n];
        o_ls[n] = logstd[a * N + n];
        o_bias[n] = bias[a * N + n];
        cnt[n] = 0;
        dead[n] = max_active[a * N + n] < threshold ? 1 : 0;            // a2c.py:395
    }
    __syncthreads();
    if (topk > 0) {
        // a2c.py:405-408: cand = top_k(avg, k) per action dim, choice ~ U{0..k-1} per dead particle
        __shared__ float avg_s[RS_MAXN];
        __shared__ int cand[RS_MAXN];
        __shared__ float rowsum;
        if (t == 0) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += sa[n];                     // a2c.py:394 (row sum)
            rowsum = s;
        }
        __syncthreads();
        for (int n = t; n < N; n += blockDim.x) {
            avg_s[n] = sa[n] / rowsum;
            cand[n] = n;                      // NaN averages (an all-zero row) rank nothing: every slot stays a valid index
        }
        __syncthreads();
        for (int n = t; n < N; n += blockDim.x) {
            const float v = avg_s[n];
            int rank = 0;
            for (int i = 0; i < N; ++i) rank += (avg_s[i] > v || (avg_s[i] == v && i < n)) ? 1 : 0;
            if (rank < topk) cand[rank] = n;
        }
        __syncthreads();
        for (int j = t; j < N; j += blockDim.x) {
            int c = (int)(u_resample[a * N + j] * (double)topk);
            c = min(max(c, 0), topk - 1);
            const int k = cand[c];
            if (dead[j]) atomicAdd(&cnt[k], 1);
            src[j] = k;
        }
        __syncthreads();
    } else {
    if (t == 0) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += sa[n];                         // a2c.py:394 (row sum)
        // log of the average activation, fp32-rounded from a float64 log; max over finite values
        double mx = -INFINITY;
        for (int n = 0; n < N; ++n) {
            const float avg = sa[n] / s;
            const float la = (float)log((double)avg);                   // a2c.py:402 tf.math.log
            nb[n] = la;
            if (isfinite(la)) mx = fmax(mx, (double)la);
        }
        double run = 0.0;
        for (int n = 0; n < N; ++n) {
            run += exp((double)nb[n] - mx);
            cdf[n] = run;
        }
        total_s = run;
    }
    __syncthreads();
    const double total = total_s;
    for (int j = t; j < N; j += blockDim.x) {
        // draw number j of row a (TF draws all N; only dead slots consume theirs, a2c.py:403)
        const double target = u_resample[a * N + j] * total;
        int lo = 0, hi = N;                 // first index with cdf > target (upper_bound)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] > target) hi = mid; else lo = mid + 1;
        }
        const int k = min(lo, N - 1);
        if (dead[j]) atomicAdd(&cnt[k], 1);
        src[j] = k;
    }
    __syncthreads();
    }
    // bias of a source u: old - log(count + 1 - delta), delta = 1 when u is itself dead (a2c.py:453-458)
    for (int n = t; n < N; n += blockDim.x)
        nb[n] = cnt[n] > 0 ? o_bias[n] - logf((float)cnt[n] + 1.f - (float)dead[n]) : o_bias[n];
    __syncthreads();
    for (int j = t; j < N; j += blockDim.x) {
        const int col = a * N + j;
        dead_out[col] = dead[j];
        src_out[col] = src[j];
        if (dead[j]) {
            const int k = src[j];
            const bool fixed = fixed_sigma > 0.f;                       // fixed_sigma: jitter with init_sigma, a2c.py:438-439
            const float sc = fixed ? fixed_sigma : expf(o_ls[k]);
            float base = o_loc[k];
            if (fixed && tanh_mode) base = tanhf(base);                 // a2c.py:430-432
            float noise = sc * u_noise[col];                            // a2c.py:441
            noise = noise + (noise < 0.f ? -1e-4f : 1e-4f);             // a2c.py:442-444
            float nl = base + noise;                                    // a2c.py:445
            if (tanh_mode) {                                            // a2c.py:448-450 (reference quirk)
                nl = fmaxf(fminf(nl, 0.999999f), -0.999999f);
                nl = atanhf(nl);
            }
            loc[col] = nl;
            if (!fixed) logstd[col] = fmaxf(fminf(o_ls[k], 2.f), -20.f);   // a2c.py:451, :463-465 (not with fixed_sigma)
            bias[col] = nb[k];                                          // second scatter wins (a2c.py:466)
        } else {
            bias[col] = nb[j];                                          // first scatter (a2c.py:465)
        }
    }
}

// weight[h, c] = old weight[h, src(c)] for every dead column c.  One block per weight row.
__global__ __launch_bounds__(256) void k_resample_weight(float* __restrict__ weight, int H, int AN, int N,
                                                         const int* __restrict__ dead,
                                                         const int* __restrict__ src) {
    extern __shared__ float row[];
    float* w = weight + (size_t)blockIdx.x * AN;
    for (int c = threadIdx.x; c < AN; c += blockDim.x) row[c] = w[c];
    __syncthreads();
    for (int c = threadIdx.x; c < AN; c += blockDim.x)
        if (dead[c]) w[c] = row[(c / N) * N + src[c]];
}

}  // namespace pfpn

using namespace pfpn;

extern "C" int pfpn_resample(float* loc, float* logstd, float* bias, float* weight, int H,
                             const float* max_active, const float* sum_active, const double* u_resample,
                             const float* u_noise, int A, int N, float threshold, int tanh_mode, int topk,
                             float fixed_sigma, int32_t* dead_out, int32_t* src_out, void* stream) {
    PFPN_BEGIN();
    if (!loc || !logstd || !bias || !weight || !max_active || !sum_active || !u_resample || !u_noise ||
        !dead_out || !src_out || A < 1 || N < 1 || N > RS_MAXN || H < 1 || topk < 0)
        return PFPN_ERR_ARG;
    if (topk > N) topk = N;                                               // a2c.py:405  k = min(n, resample)
    const size_t lds = (size_t)A * N * sizeof(float);
    if (lds > 64 * 1024) return PFPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (threshold <= 0.f) threshold = (float)(0.05 / (double)N);         // a2c.py:391 (.05/n in float64, then fp32)
    hipLaunchKernelGGL(k_resample_dims, dim3(A), dim3(256), 0, st, loc, logstd, bias, max_active, sum_active,
                       u_resample, u_noise, A, N, threshold, tanh_mode, topk, fixed_sigma, (int*)dead_out, (int*)src_out);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_resample_weight, dim3(H), dim3(256), lds, st, weight, H, A * N, N, (const int*)dead_out,
                       (const int*)src_out);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}
