/* Host twins of three C-ABI entry points (include/pfpn_hip.h), in plain C for the CPU.
 *
 * *** TEST / BENCH INFRASTRUCTURE -- NOT PRODUCT CODE. ***  Nothing under pfpn_amd/ loads this library; the product has
 * no CPU path (tests/test_abi_cpu.py::test_no_cpu_fallback_in_product).  It exists for two things: a compiled-code CPU
 * baseline of the dominant kernel next to the NumPy / torch-CPU ones (oracle/cpu_baseline.py, leg `mixture_c`), and a
 * second, independent restatement that tests/test_oracle_cpu_twin.py holds against oracle/pfpn_oracle.py.
 *
 * Same argument meaning as the device entry points, host pointers, no stream, `_cpu` suffix:
 *   pfpn_logprob_fwd_cpu   MixtureGaussianDistribution.log_prob       reference networks/utils.py:108-134
 *   pfpn_logprob_bwd_cpu   its tf.gradients expansion + the NaN/Inf guard   utils.py:109-117 (oracle.mixture_log_prob_bwd)
 *   pfpn_gae_cpu           generalized_advantage_estimate             reference networks/actor_critic/a2c.py:30-40, utils.py:5-15
 * Arithmetic in fp32 in the reference's op order (probability space, no log-sum-exp: p may underflow to 0 -> log p = -inf);
 * sums run left to right (NumPy sums pairwise: results agree to fp32 rounding, not bit for bit).  Rows are independent:
 * `#pragma omp parallel for` over the batch; particle-grid gradients go through per-thread partial sums added in thread order.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HALF_LOG_2PI 0.9189385332046727f
#define LOG2F_ 0.6931471805599453f

static float softplusf(float v) { /* tf.nn.softplus = logaddexp(0, v) */
    return v > 0.f ? v + log1pf(expf(-v)) : log1pf(expf(v));
}
static float tanh_correction(float x) { /* 2 (log 2 - x - softplus(-2x)) = log(1 - tanh(x)^2), utils.py:132-133 */
    return 2.f * (LOG2F_ - x - softplusf(-2.f * x));
}

void pfpn_cpu_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int pfpn_cpu_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* logits [B,A,N], x [B,A], loc / logstd [A,N] -> logp [B], pmix [B,A] (optional) */
int pfpn_logprob_fwd_cpu(const float* logits, const float* x, const float* loc, const float* logstd, int B, int A, int N,
                         int tanh_mode, float* logp, float* pmix) {
    if (!logits || !x || !loc || !logstd || !logp || B < 1 || A < 1 || N < 1) return 1;
    float* scale = (float*)malloc(sizeof(float) * 2 * (size_t)A * N);
    if (!scale) return 2;
    float* cst = scale + (size_t)A * N;
    for (int i = 0; i < A * N; ++i) {
        scale[i] = expf(logstd[i]);
        cst[i] = HALF_LOG_2PI + logf(scale[i]);
    }
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        float row = 0.f;
        for (int a = 0; a < A; ++a) {
            const float* l = logits + ((size_t)b * A + a) * N;
            const float xv = x[(size_t)b * A + a];
            float mx = l[0];
            for (int n = 1; n < N; ++n) mx = l[n] > mx ? l[n] : mx;
            float se = 0.f;
            for (int n = 0; n < N; ++n) se += expf(l[n] - mx);
            float p = 0.f;
            for (int n = 0; n < N; ++n) {
                const float w = expf(l[n] - mx) / se;
                const float z = (xv - loc[a * N + n]) / scale[a * N + n];
                p += w * expf(-0.5f * (z * z) - cst[a * N + n]);
            }
            float lp = logf(p);
            if (tanh_mode) lp -= tanh_correction(xv);
            if (pmix) pmix[(size_t)b * A + a] = p;
            row += lp;
        }
        logp[b] = row;
    }
    free(scale);
    return 0;
}

/* -> dlogits [B,A,N], dloc / dlogstd [A,N], dx [B,A] (optional) */
int pfpn_logprob_bwd_cpu(const float* logits, const float* x, const float* loc, const float* logstd, const float* dlogp,
                         int B, int A, int N, int tanh_mode, float* dlogits, float* dloc, float* dlogstd, float* dx) {
    if (!logits || !x || !loc || !logstd || !dlogp || !dlogits || !dloc || !dlogstd || B < 1 || A < 1 || N < 1) return 1;
    const size_t AN = (size_t)A * N;
    const int T = pfpn_cpu_threads();
    float* scale = (float*)malloc(sizeof(float) * (2 * AN + 2 * AN * (size_t)T));
    if (!scale) return 2;
    float* cst = scale + AN;
    float* part = cst + AN;                                  /* [T][2][A*N] */
    memset(part, 0, sizeof(float) * 2 * AN * (size_t)T);
    for (size_t i = 0; i < AN; ++i) {
        scale[i] = expf(logstd[i]);
        cst[i] = HALF_LOG_2PI + logf(scale[i]);
    }
#pragma omp parallel
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        float* pl = part + (size_t)tid * 2 * AN;
        float* ps = pl + AN;
        float* w = (float*)malloc(sizeof(float) * 3 * (size_t)N);
        float* phi = w + N;
        float* zz = phi + N;
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            for (int a = 0; a < A; ++a) {
                const float* l = logits + ((size_t)b * A + a) * N;
                float* dl = dlogits + ((size_t)b * A + a) * N;
                const float xv = x[(size_t)b * A + a];
                float mx = l[0];
                for (int n = 1; n < N; ++n) mx = l[n] > mx ? l[n] : mx;
                float se = 0.f;
                for (int n = 0; n < N; ++n) se += expf(l[n] - mx);
                float p = 0.f;
                for (int n = 0; n < N; ++n) {
                    w[n] = expf(l[n] - mx) / se;
                    zz[n] = (xv - loc[a * N + n]) / scale[a * N + n];
                    phi[n] = expf(-0.5f * (zz[n] * zz[n]) - cst[a * N + n]);
                    p += w[n] * phi[n];
                }
                float q = dlogp[b] / p;
                if (isnan(q) || isinf(q)) q = 0.f;           /* the reference's foo() guard, utils.py:109-117 */
                float dxa = 0.f;
                for (int n = 0; n < N; ++n) {
                    const float qw = q * w[n];
                    dl[n] = qw * (phi[n] - p);
                    const float t = qw * phi[n];
                    const float tz = t * zz[n] / scale[a * N + n];
                    pl[a * N + n] += tz;
                    ps[a * N + n] += t * (zz[n] * zz[n] - 1.f);
                    dxa += tz;
                }
                if (dx) dx[(size_t)b * A + a] = -dxa + (tanh_mode ? dlogp[b] * 2.f * tanhf(xv) : 0.f);
            }
        }
        free(w);
    }
    for (size_t i = 0; i < AN; ++i) {
        float s0 = 0.f, s1 = 0.f;
        for (int t = 0; t < T; ++t) {                        /* thread order: reproducible for a fixed thread count */
            s0 += part[(size_t)t * 2 * AN + i];
            s1 += part[(size_t)t * 2 * AN + AN + i];
        }
        dloc[i] = s0;
        dlogstd[i] = s1;
    }
    free(scale);
    return 0;
}

/* E ragged episodes back to back: reward [n], value [n + E] (episode e owns value[off[e]+e .. off[e+1]+e], its last entry the
 * bootstrap), offsets [E+1] -> adv [n].  td in fp32, the reverse scan in float64 (the reference's Python float accumulator). */
int pfpn_gae_cpu(const float* reward, const float* value, const long long* offsets, int E, double gamma, double gae_gamma,
                 float* adv) {
    if (!reward || !value || !offsets || !adv || E < 1) return 1;
#pragma omp parallel for schedule(dynamic)
    for (int e = 0; e < E; ++e) {
        const long long lo = offsets[e], hi = offsets[e + 1];
        const float* v = value + lo + e;
        double acc = 0.0;
        for (long long t = hi - 1; t >= lo; --t) {
            float td = reward[t] + (float)gamma * v[t - lo + 1];
            td = td - v[t - lo];
            acc = (double)td + gae_gamma * acc;
            adv[t] = (float)acc;
        }
    }
    return 0;
}
