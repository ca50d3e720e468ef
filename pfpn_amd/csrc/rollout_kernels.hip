// Rollout-side particle kernels for gfx950: categorical particle draw + Gaussian jitter + log-prob of
// the drawn action (K3), activation statistics for the resampler, deterministic (argmax) action,
// categorical entropy, and the Gumbel-softmax reparameterised sample of the SAC path (K4).
//
// Replaces the TF op subgraphs of MixtureGaussianDistribution.sample / .mean / .entropy
// (reference networks/utils.py:146-236) and the running max/sum statistics of
// ParticleFilteringA2CNetwork.init (reference networks/actor_critic/a2c.py:346-365).
//
// Index parity: the categorical draw reproduces the TF-1.14 CPU `Multinomial` rule -- a float64
// running sum of exp(double(l) - max), left to right, first index whose sum exceeds u * total --
// summed in particle order so the rounding order equals the oracle's (bit-exact indices).
// Every random number is an explicit input (TF's Philox stream is not reproducible without TF).
#include "pfpn_common.h"

namespace pfpn {

// One thread per (b, a) segment.  Outputs k, action, per-segment log p of the drawn action, and
// the softmax normaliser (max, 1/sum) reused by the statistics kernel.
__global__ __launch_bounds__(256) void k_sample_plain(
    const float* __restrict__ logits, const float* __restrict__ loc, const float* __restrict__ logstd,
    const double* __restrict__ u_cat, const float* __restrict__ eps_sel,
    int B, int A, int N,
    int* __restrict__ k_out, float* __restrict__ action, float* __restrict__ lpseg,
    float* __restrict__ smax, float* __restrict__ sinv)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B * A) return;
    const int a = s % A;
    const float* row = logits + (size_t)s * N;
    const float* locr = loc + a * N;
    const float* lsr = logstd + a * N;
    float mx = row[0];
    for (int n = 1; n < N; ++n) mx = fmaxf(mx, row[n]);
    // TF Multinomial: max and exp in double
    const double mxd = (double)mx;
    double total = 0.0;
    for (int n = 0; n < N; ++n) total += exp((double)row[n] - mxd);
    const double target = u_cat[s] * total;
    double run = 0.0;
    int k = N - 1;
    for (int n = 0; n < N; ++n) {
        run += exp((double)row[n] - mxd);
        if (run > target) { k = n; break; }
    }
    const float sck = expf(lsr[k]);
    const float act = locr[k] + sck * eps_sel[s];
    // log-prob of the drawn action under the same mixture (fp32, as pfpn_logprob_fwd)
    float sum = 0.f, num = 0.f;
    for (int n = 0; n < N; ++n) {
        const float sc = expf(lsr[n]);
        const float en = expf(row[n] - mx);
        const float z = (act - locr[n]) * (1.0f / sc);
        const float ph = expf(fmaf(-0.5f * z, z, -(PFPN_HALF_LOG_2PI + logf(sc))));
        sum += en;
        num += en * ph;
    }
    const float inv = 1.0f / sum;
    k_out[s] = k;
    action[s] = act;
    lpseg[s] = logf(num * inv);
    smax[s] = mx;
    sinv[s] = inv;
}

// The same draw with one WAVE per (b, a) segment (N <= 64: lane n <-> particle n).  The expensive parts -- exp in double,
// three expf and a logf per particle -- happen once per lane in parallel; what has to stay sequential for index parity, the
// float64 running sum (and, so that log-prob and normaliser keep their bits, the two fp32 sums), walks the lanes in order
// on wave-uniform values: N dependent adds instead of 2 N double exp + 4 N expf in one thread (27.6 -> ~5 us for the
// 288 segments of an 8-environment rollout step).  The first index whose running sum exceeds u * total is one ballot.
__global__ __launch_bounds__(256) void k_sample_wave(
    const float* __restrict__ logits, const float* __restrict__ loc, const float* __restrict__ logstd,
    const double* __restrict__ u_cat, const float* __restrict__ eps_sel,
    int B, int A, int N,
    int* __restrict__ k_out, float* __restrict__ action, float* __restrict__ lpseg,
    float* __restrict__ smax, float* __restrict__ sinv)
{
    const int lane = threadIdx.x & 63;
    const int s = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (s >= B * A) return;
    const int a = s % A;
    const bool live = lane < N;
    const int n = live ? lane : 0;
    const float l = logits[(size_t)s * N + n];
    const float lc = loc[a * N + n], ls = logstd[a * N + n];
    const float mx = wave_max_dpp(live ? l : -INFINITY);
    // TF Multinomial: max and exp in double
    const double ed = live ? exp((double)l - (double)mx) : 0.0;
    const int ed_lo = __double2loint(ed), ed_hi = __double2hiint(ed);
    double run = 0.0, my_run = 0.0;
    for (int j = 0; j < N; ++j) {
        const double e_j = __hiloint2double(__builtin_amdgcn_readlane(ed_hi, j), __builtin_amdgcn_readlane(ed_lo, j));
        run += e_j;
        if (lane == j) my_run = run;
    }
    const double target = u_cat[s] * run;                   // run == total
    const unsigned long long over = __ballot(live && my_run > target);
    const int k = over ? (int)__builtin_ctzll(over) : N - 1;
    const float sck = expf(__shfl(ls, k, 64));
    const float act = __shfl(lc, k, 64) + sck * eps_sel[s];
    // log-prob of the drawn action under the same mixture (fp32, as pfpn_logprob_fwd), sums in particle order
    const float sc = expf(ls);
    const float en = live ? expf(l - mx) : 0.f;
    const float z = (act - lc) * (1.0f / sc);
    const float ph = expf(fmaf(-0.5f * z, z, -(PFPN_HALF_LOG_2PI + logf(sc))));
    const float ep = en * ph;
    float sum = 0.f, num = 0.f;
    for (int j = 0; j < N; ++j) {
        sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(en), j));
        num += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ep), j));
    }
    if (lane == 0) {
        const float inv = 1.0f / sum;
        k_out[s] = k;
        action[s] = act;
        lpseg[s] = logf(num * inv);
        smax[s] = mx;
        sinv[s] = inv;
    }
}

// softmax normalisers only (for pfpn_active_stats without a draw)
__global__ __launch_bounds__(256) void k_softmax_norm(const float* __restrict__ logits, int S, int N,
                                                      float* __restrict__ smax, float* __restrict__ sinv) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const float* row = logits + (size_t)s * N;
    float mx = row[0];
    for (int n = 1; n < N; ++n) mx = fmaxf(mx, row[n]);
    float sum = 0.f;
    for (int n = 0; n < N; ++n) sum += expf(row[n] - mx);
    smax[s] = mx;
    sinv[s] = 1.0f / sum;
}

// Column statistics of the softmax weights over a chunk of rows: thread <-> particle column
// (a,n) (coalesced across the row), rows walked in order -> deterministic.
// partial[c][0][col] = max_b w, partial[c][1][col] = sum_b w over rows b = c, c+C, ...? no:
// contiguous chunks [c*len, (c+1)*len) so the final sum adds chunks in row order.
__global__ __launch_bounds__(256) void k_stats_partial(
    const float* __restrict__ logits, const float* __restrict__ smax, const float* __restrict__ sinv,
    int B, int A, int N, int chunk_len, float* __restrict__ partial,
    float* __restrict__ max_active = nullptr, float* __restrict__ sum_active = nullptr)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int AN = A * N;
    if (col >= AN) return;
    const int a = col / N;
    const int c = blockIdx.y;
    const int b0 = c * chunk_len, b1 = min(B, b0 + chunk_len);
    float m = 0.f, s = 0.f;
    for (int b = b0; b < b1; ++b) {
        const float w = expf(logits[(size_t)b * AN + col] - smax[b * A + a]) * sinv[b * A + a];
        m = fmaxf(m, w);
        s += w;
    }
    if (max_active) {
        // a single chunk (rollout batches of <= 64 rows): the running statistics are updated here, exactly as
        // k_stats_apply does with one slab -- one launch less per environment step
        max_active[col] = fmaxf(max_active[col], m);      // a2c.py:356
        sum_active[col] = sum_active[col] + s;            // a2c.py:359
        return;
    }
    partial[((size_t)c * 2 + 0) * AN + col] = m;
    partial[((size_t)c * 2 + 1) * AN + col] = s;
}

__global__ __launch_bounds__(1024) void k_stats_apply(const float* __restrict__ partial, int C, int AN,
                                                      float* __restrict__ max_active,
                                                      float* __restrict__ sum_active) {
    __shared__ float red[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    // slab c holds [max | sum] rows of AN floats: stride 2*AN between chunks
    const float m = slab_reduce_1024<true>(partial, C, (size_t)2 * AN, col, col < AN, red);
    const float s = slab_reduce_1024<false>(partial + AN, C, (size_t)2 * AN, col, col < AN, red);
    if ((threadIdx.x >> 6) == 0 && col < AN) {
        max_active[col] = fmaxf(max_active[col], m);      // a2c.py:356
        sum_active[col] = sum_active[col] + s;            // a2c.py:359
    }
}

__global__ __launch_bounds__(256) void k_row_sum2(const float* __restrict__ lpseg, int B, int A,
                                                  float* __restrict__ logp) {
    const int lane = threadIdx.x & 63;
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= B) return;
    float s = 0.f;
    for (int a = lane; a < A; a += 64) s += lpseg[(size_t)b * A + a];
    s = wave_sum(s);
    if (lane == 0) logp[b] = s;
}

// deterministic action: loc[a, argmax_n logits] (first index on ties), tanh in tanh mode
__global__ __launch_bounds__(256) void k_mean_action(const float* __restrict__ logits,
                                                     const float* __restrict__ loc, int B, int A, int N,
                                                     int tanh_mode, int* __restrict__ k_out,
                                                     float* __restrict__ action) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B * A) return;
    const int a = s % A;
    const float* row = logits + (size_t)s * N;
    int k = 0;
    float mx = row[0];
    for (int n = 1; n < N; ++n) if (row[n] > mx) { mx = row[n]; k = n; }
    float v = loc[a * N + k];
    if (tanh_mode) v = tanhf(v);
    if (k_out) k_out[s] = k;
    action[s] = v;
}

// categorical entropy H[b,a] = log s1 - sum w v   (utils.py:146-151)
__global__ __launch_bounds__(256) void k_entropy_fwd(const float* __restrict__ logits, int S, int N,
                                                     float* __restrict__ H) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const float* row = logits + (size_t)s * N;
    float mx = row[0];
    for (int n = 1; n < N; ++n) mx = fmaxf(mx, row[n]);
    float s1 = 0.f;
    for (int n = 0; n < N; ++n) s1 += expf(row[n] - mx);
    const float ls1 = logf(s1), inv = 1.0f / s1;
    float h = 0.f;
    for (int n = 0; n < N; ++n) {
        const float v = row[n] - mx;
        h += (expf(v) * inv) * (ls1 - v);
    }
    H[s] = h;
}

// dlogits = -w (log w + H) dH   (accumulate: += into an existing gradient)
__global__ __launch_bounds__(256) void k_entropy_bwd(const float* __restrict__ logits,
                                                     const float* __restrict__ dH, int S, int N,
                                                     int accumulate, float* __restrict__ dlogits) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const float* row = logits + (size_t)s * N;
    float* drow = dlogits + (size_t)s * N;
    float mx = row[0];
    for (int n = 1; n < N; ++n) mx = fmaxf(mx, row[n]);
    float s1 = 0.f;
    for (int n = 0; n < N; ++n) s1 += expf(row[n] - mx);
    const float ls1 = logf(s1), inv = 1.0f / s1;
    float h = 0.f;
    for (int n = 0; n < N; ++n) {
        const float v = row[n] - mx;
        h += (expf(v) * inv) * (ls1 - v);
    }
    const float g = dH[s];
    for (int n = 0; n < N; ++n) {
        const float v = row[n] - mx;
        const float w = expf(v) * inv;
        const float d = -(w * ((v - ls1) + h)) * g;
        drow[n] = accumulate ? drow[n] + d : d;
    }
}

// SAC reparameterised sample: y = logits + gumbel, k = argmax y, p = loc + scale * eps,
// a = tanh(p_k), s_ = p_k   (utils.py:156-186)
__global__ __launch_bounds__(256) void k_rsample_fwd(
    const float* __restrict__ logits, const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ gumbel, const float* __restrict__ eps, int B, int A, int N,
    int* __restrict__ k_out, float* __restrict__ a_out, float* __restrict__ s_out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B * A) return;
    const int a = s % A;
    const float* row = logits + (size_t)s * N;
    const float* gr = gumbel + (size_t)s * N;
    int k = 0;
    float mx = row[0] + gr[0];
    for (int n = 1; n < N; ++n) {
        const float y = row[n] + gr[n];
        if (y > mx) { mx = y; k = n; }
    }
    const float pk = loc[a * N + k] + expf(logstd[a * N + k]) * eps[(size_t)s * N + k];
    k_out[s] = k;
    s_out[s] = pk;
    a_out[s] = tanhf(pk);
}

// backward of the custom gradients `mask`/`mask2` (utils.py:165-183) through softmax(logits+g)
// and p = loc + scale*eps.  Per segment: dlogits row, and the (k, d p_k) pair for the parameter
// scatter done by k_rsample_param_partial / k_param_grad_final.
__global__ __launch_bounds__(256) void k_rsample_bwd(
    const float* __restrict__ logits, const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ gumbel, const float* __restrict__ eps,
    const float* __restrict__ da, const float* __restrict__ ds, int B, int A, int N, int accumulate,
    float* __restrict__ dlogits, int* __restrict__ k_tmp, float* __restrict__ gp_tmp,
    float* __restrict__ gs_tmp)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B * A) return;
    const int a = s % A;
    const float* row = logits + (size_t)s * N;
    const float* gr = gumbel + (size_t)s * N;
    const float* er = eps + (size_t)s * N;
    const float* locr = loc + a * N;
    const float* lsr = logstd + a * N;
    float* drow = dlogits + (size_t)s * N;
    int k = 0;
    float mx = row[0] + gr[0];
    for (int n = 1; n < N; ++n) {
        const float y = row[n] + gr[n];
        if (y > mx) { mx = y; k = n; }
    }
    float sum = 0.f;
    for (int n = 0; n < N; ++n) sum += expf((row[n] + gr[n]) - mx);
    const float inv = 1.0f / sum;
    const float sck = expf(lsr[k]);
    const float tk = tanhf(locr[k] + sck * er[k]);
    const float one_m = 1.f - tk * tk;
    const float gda = da[s], gds = ds[s];
    const float cs = gds / fmaxf(1e-6f, one_m);
    float dot = 0.f;                                        // sum_n gw_n w_n
    for (int n = 0; n < N; ++n) {
        const float w = expf((row[n] + gr[n]) - mx) * inv;
        const float gap = tanhf(locr[n] + expf(lsr[n]) * er[n]) - tk;
        dot += (gap * gda + gap * cs) * w;
    }
    for (int n = 0; n < N; ++n) {
        const float w = expf((row[n] + gr[n]) - mx) * inv;
        const float gap = tanhf(locr[n] + expf(lsr[n]) * er[n]) - tk;
        const float d = w * ((gap * gda + gap * cs) - dot);
        drow[n] = accumulate ? drow[n] + d : d;
    }
    const float gp = gda * one_m + gds;                     // d L / d p_k
    k_tmp[s] = k;
    gp_tmp[s] = gp;
    gs_tmp[s] = gp * sck * er[k];                           // d L / d logstd_k
}

// ---------------------------------------------------------------------------------------
// Quad forms of the two kernels above for N <= 64 (every shipped setting): 4 lanes per (row, action-dim)
// segment -- lane g owns particles g, g+4, ... in registers -- so a wave covers 16 consecutive segments
// (one contiguous 64*N-byte span of logits / gumbel / eps: coalesced, unlike one thread per 140-byte row),
// every exp/tanh is evaluated once, and max / argmax / sums are two quad-permute DPP steps.
// The argmax keeps the serial rule: the FIRST maximum wins (ties -> lowest particle index).
// ---------------------------------------------------------------------------------------
constexpr int RS_G = 4;
constexpr int RS_NP = 16;                               // particles per lane, N <= 64

__device__ __forceinline__ void quad_argmax(float& v, int& i) {
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
        const float ov = __shfl_xor(v, m, 64);
        const int oi = __shfl_xor(i, m, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// Both quad kernels stage their wave's 16 consecutive segments -- one contiguous span of 16*N floats per tensor -- through
// LDS with float4 wave loads: a lane's own particles g, g+4, .. sit 16 bytes apart per load instruction, i.e. 16 separate
// lines per wave load when read straight from memory (rsample backward at 1024 x 36 x 35: 14.7 us for 25 MB).  The
// arithmetic per lane is unchanged, so the results keep their bits.
constexpr int RS_SEGS = 16;                             // segments per wave (64 lanes / RS_G)
constexpr int RS_STAGE_MIN_SEGS = 1024 * 36;            // stage from here on (SAC step, same box: 256-row step 2 % slower staged, 1024 rows 0.7 % faster, 4096 rows 4 %)

// floats [base, base + count) of `src` -> lds[0, count): float4 wave loads (base is a multiple of 16 floats), scalar tail
__device__ __forceinline__ void rs_stage_in(float* __restrict__ lds, const float* __restrict__ src, size_t base, int count,
                                            int lane) {
    const int nq = (((uintptr_t)(src + base) & 15) == 0) ? count >> 2 : 0;     // (a view at an odd offset: scalar loads)
    for (int q = lane; q < nq; q += 64)
        *reinterpret_cast<float4*>(lds + 4 * q) = *reinterpret_cast<const float4*>(src + base + 4 * q);
    for (int i = 4 * nq + lane; i < count; i += 64) lds[i] = src[base + i];
}

// STAGE = false: the lanes read their particles straight from memory (small batches: the staging's barrier and LDS round
// trip cost more than the scattered loads; measured 2 % of a 256-row SAC step).
template <bool STAGE>
__global__ __launch_bounds__(256) void k_rsample_fwd_quad(
    const float* __restrict__ logits, const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ gumbel, const float* __restrict__ eps, int B, int A, int N,
    int* __restrict__ k_out, float* __restrict__ a_out, float* __restrict__ s_out)
{
    extern __shared__ float rs_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = B * A;
    const int seg0 = (blockIdx.x * 4 + wv) * RS_SEGS;
    const int span = ((RS_SEGS * N + 3) & ~3);            // floats per staged tensor and wave
    float* ll = rs_lds + (size_t)wv * 2 * span;
    float* lg = ll + span;
    const int nseg = max(0, min(RS_SEGS, S - seg0));
    if (STAGE) {
        rs_stage_in(ll, logits, (size_t)seg0 * N, nseg * N, lane);
        rs_stage_in(lg, gumbel, (size_t)seg0 * N, nseg * N, lane);
        __syncthreads();
    }
    const int j = lane >> 2, g = lane & 3;
    const bool live = j < nseg;
    const int s = live ? seg0 + j : 0;
    const int a = s % A;
    const float* row = STAGE ? ll + j * N : logits + (size_t)s * N;
    const float* gr = STAGE ? lg + j * N : gumbel + (size_t)s * N;
    float mx = -INFINITY;
    int k = 0x7fffffff;
    if (live)
        for (int n = g; n < N; n += RS_G) {
            const float y = row[n] + gr[n];
            if (y > mx) { mx = y; k = n; }
        }
    quad_argmax(mx, k);
    if (live && g == 0) {
        const float pk = loc[a * N + k] + expf(logstd[a * N + k]) * eps[(size_t)s * N + k];
        k_out[s] = k;
        s_out[s] = pk;
        a_out[s] = tanhf(pk);
    }
}

template <bool STAGE>
__global__ __launch_bounds__(256) void k_rsample_bwd_quad(
    const float* __restrict__ logits, const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ gumbel, const float* __restrict__ eps,
    const float* __restrict__ da, const float* __restrict__ ds, int B, int A, int N, int accumulate,
    float* __restrict__ dlogits, int* __restrict__ k_tmp, float* __restrict__ gp_tmp,
    float* __restrict__ gs_tmp)
{
    extern __shared__ float rs_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = B * A;
    const int seg0 = (blockIdx.x * 4 + wv) * RS_SEGS;
    const int span = ((RS_SEGS * N + 3) & ~3);
    float* ll = rs_lds + (size_t)wv * 4 * span;
    float* lg = ll + span;
    float* le = lg + span;
    float* ld = le + span;                                // dlogits: in (accumulate) and out
    const int nseg = max(0, min(RS_SEGS, S - seg0));
    const size_t base = (size_t)seg0 * N;
    if (STAGE) {
        rs_stage_in(ll, logits, base, nseg * N, lane);
        rs_stage_in(lg, gumbel, base, nseg * N, lane);
        rs_stage_in(le, eps, base, nseg * N, lane);
        if (accumulate) rs_stage_in(ld, dlogits, base, nseg * N, lane);
        __syncthreads();
    }
    const int j = lane >> 2, g = lane & 3;
    const bool live = j < nseg;
    const int s = live ? seg0 + j : 0;
    const int a = s % A;
    const float* row = STAGE ? ll + j * N : logits + (size_t)s * N;
    const float* gr = STAGE ? lg + j * N : gumbel + (size_t)s * N;
    const float* er = STAGE ? le + j * N : eps + (size_t)s * N;
    const float* locr = loc + a * N;
    const float* lsr = logstd + a * N;
    float y[RS_NP], th[RS_NP];
    float mx = -INFINITY;
    int k = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < RS_NP; ++i) {
        const int n = g + RS_G * i;
        y[i] = -INFINITY; th[i] = 0.f;
        if (n < N && live) {
            y[i] = row[n] + gr[n];
            th[i] = tanhf(locr[n] + expf(lsr[n]) * er[n]);       // tanh of every particle's candidate sample
            if (y[i] > mx) { mx = y[i]; k = n; }
        }
    }
    quad_argmax(mx, k);
    if (!live) k = 0;                                     // (dead quads: finite arithmetic below, nothing stored)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RS_NP; ++i) {
        y[i] = (g + RS_G * i < N && live) ? expf(y[i] - mx) : 0.f;        // y <- softmax numerator
        sum += y[i];
    }
    sum = quad_sum(sum);
    const float inv = 1.0f / sum;
    const float sck = expf(lsr[k]);
    const float tk = tanhf(locr[k] + sck * er[k]);
    const float one_m = 1.f - tk * tk;
    const float gda = da[s], gds = ds[s];
    const float cs = gds / fmaxf(1e-6f, one_m);
    float dot = 0.f;                                        // sum_n gw_n w_n
#pragma unroll
    for (int i = 0; i < RS_NP; ++i) {
        const float gap = th[i] - tk;
        th[i] = gap * gda + gap * cs;                       // th <- gw_n
        y[i] *= inv;                                        // y <- w_n
        dot += th[i] * y[i];
    }
    dot = quad_sum(dot);
    if (live) {
        float* drow = STAGE ? ld + j * N : dlogits + (size_t)s * N;
#pragma unroll
        for (int i = 0; i < RS_NP; ++i) {
            const int n = g + RS_G * i;
            if (n < N) {
                const float d = y[i] * (th[i] - dot);
                drow[n] = accumulate ? drow[n] + d : d;
            }
        }
        if (g == 0) {
            const float gp = gda * one_m + gds;             // d L / d p_k
            k_tmp[s] = k;
            gp_tmp[s] = gp;
            gs_tmp[s] = gp * sck * er[k];                   // d L / d logstd_k
        }
    }
    if (!STAGE) return;
    __syncthreads();
    // the wave's span of dlogits back to memory: float4 wave stores, scalar tail
    const int count = nseg * N, nq = (((uintptr_t)(dlogits + base) & 15) == 0) ? count >> 2 : 0;
    for (int q = lane; q < nq; q += 64)
        *reinterpret_cast<float4*>(dlogits + base + 4 * q) = *reinterpret_cast<const float4*>(ld + 4 * q);
    for (int i = 4 * nq + lane; i < count; i += 64) dlogits[base + i] = ld[i];
}

// Parameter scatter, deterministic and parallel: wave (chunk c, dim a) walks its rows IN ORDER, 64 at a time
// (one coalesced load of the (k, dp, dlogstd) triples, then a 64-step register loop: lane n adds row j's pair
// when k_j == n), and leaves a slab of per-particle partial sums; k_param_grad_final adds the slabs in chunk order.
__global__ __launch_bounds__(256) void k_rsample_param_partial(
    const int* __restrict__ k_tmp, const float* __restrict__ gp_tmp, const float* __restrict__ gs_tmp,
    int B, int A, int N, int C, int chunk_len, float* __restrict__ partial)
{
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= C * A) return;
    const int a = w % A, c = w / A;
    const int b0 = c * chunk_len, b1 = min(B, b0 + chunk_len);
    for (int n0 = 0; n0 < N; n0 += 64) {
        const int n = n0 + lane;
        float sl = 0.f, ss = 0.f;
        for (int b = b0; b < b1; b += 64) {
            const int bj = min(b + lane, b1 - 1);
            const int kj = k_tmp[bj * A + a];
            const float gpj = gp_tmp[bj * A + a], gsj = gs_tmp[bj * A + a];
            const int cnt = min(64, b1 - b);
            for (int j = 0; j < cnt; ++j) {
                const int kk = __shfl(kj, j, 64);
                const float p_ = __shfl(gpj, j, 64), s_ = __shfl(gsj, j, 64);
                if (kk == n) { sl += p_; ss += s_; }
            }
        }
        if (n < N) {
            partial[((size_t)c * 2 + 0) * A * N + a * N + n] = sl;
            partial[((size_t)c * 2 + 1) * A * N + a * N + n] = ss;
        }
    }
}

__global__ __launch_bounds__(1024) void k_param_grad_final(const float* __restrict__ partial, int C, int AN,
                                                           int accumulate, float* __restrict__ dloc,
                                                           float* __restrict__ dlogstd) {
    __shared__ float red[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);      // column of the [2*AN] slab
    const float s = slab_reduce_1024<false>(partial, C, (size_t)2 * AN, col, col < 2 * AN, red);
    if ((threadIdx.x >> 6) != 0 || col >= 2 * AN) return;
    float* dst = col < AN ? dloc + col : dlogstd + (col - AN);
    *dst = accumulate ? *dst + s : s;
}

static inline int stats_chunks(int B) {
    int C = (B + 63) / 64;
    if (C > 64) C = 64;
    if (C < 1) C = 1;
    return C;
}

// Gumbel noise of the relaxed one-hot draw from uniforms, in place: g = -log(-log(max(u, tiny)))
// (TFP 0.7 ExpRelaxedOneHotCategorical's sampler behind utils.py:96, :156-160).  One launch instead of the five
// elementwise passes (clamp, log, neg, log, neg) the same transform takes as tensor ops.
// All four noise tensors of a SAC learner step from ONE buffer of uniforms [4][n], in place: rows 0 and 1 become
// Gumbel noise (as k_gumbel_inplace), rows 2 and 3 a pair of independent standard normals by Box-Muller
// (z0 = r cos t, z1 = r sin t, r = sqrt(-2 ln max(u2, tiny)), t = 2 pi u3) -- tf.random.normal's role in utils.py:161.
__global__ __launch_bounds__(256) void k_sac_noise_inplace(float* __restrict__ u, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        u[i] = -logf(-logf(fmaxf(u[i], 1.17549435e-38f)));
        u[n + i] = -logf(-logf(fmaxf(u[n + i], 1.17549435e-38f)));
        const float r = sqrtf(-2.f * logf(fmaxf(u[2 * n + i], 1.17549435e-38f)));
        float sn, cs;
        sincosf(6.28318530717958647692f * u[3 * n + i], &sn, &cs);
        u[2 * n + i] = r * cs;
        u[3 * n + i] = r * sn;
    }
}

// The same four tensors WITHOUT the uniform fill: element i draws its four uniforms from one Philox4x32-10 block
// (sac_noise_element, pfpn_common.h) and writes the transformed values: one launch and 16 B per element instead of a 16-B fill
// plus a 32-B in-place pass.  `epoch` != nullptr: the offset is read from device memory (the learner step's own draw
// counter, advanced by its loss head: the same launch can then be replayed from a hipGraph).
__global__ __launch_bounds__(256) void k_sac_noise_fill(float* __restrict__ u, size_t n, unsigned long long seed,
                                                        unsigned long long offset,
                                                        const unsigned long long* __restrict__ epoch) {
    if (epoch) offset = epoch[0];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) sac_noise_element(u, n, i, seed, offset);
}

__global__ __launch_bounds__(256) void k_gumbel_inplace(float* __restrict__ u, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = fmaxf(u[i], 1.17549435e-38f);
        u[i] = -logf(-logf(v));
    }
}

}  // namespace pfpn

using namespace pfpn;

extern "C" {

size_t pfpn_rollout_workspace_bytes(int B, int A, int N) {
    // lpseg, smax, sinv, gp, gs [B*A] floats + k [B*A] ints + partial slabs [<=256][2][AN]
    return ((size_t)B * A * 6 + (size_t)256 * 2 * A * N + 64) * sizeof(float);
}

static int stats_launch(const float* logits, const float* smax, const float* sinv, int B, int A, int N,
                        float* partial, float* max_active, float* sum_active, hipStream_t st) {
    const int AN = A * N;
    const int C = stats_chunks(B);
    const int len = (B + C - 1) / C;
    if (C == 1) {
        hipLaunchKernelGGL(k_stats_partial, dim3((AN + 255) / 256, 1), dim3(256), 0, st, logits, smax, sinv, B, A,
                           N, len, partial, max_active, sum_active);
        PFPN_CHECK_LAUNCH();
        return PFPN_OK;
    }
    hipLaunchKernelGGL(k_stats_partial, dim3((AN + 255) / 256, C), dim3(256), 0, st, logits, smax, sinv, B, A,
                       N, len, partial, (float*)nullptr, (float*)nullptr);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_stats_apply, dim3((AN + 63) / 64), dim3(1024), 0, st, partial, C, AN, max_active,
                       sum_active);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_sample(const float* logits, const float* loc, const float* logstd, const double* u_cat,
                const float* eps_sel, int B, int A, int N, int32_t* k, float* action, float* logp,
                float* max_active, float* sum_active, void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!logits || !loc || !logstd || !u_cat || !eps_sel || !k || !action || B < 0 || A < 1 || N < 1)
        return PFPN_ERR_ARG;
    if ((max_active == nullptr) != (sum_active == nullptr)) return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    if (!workspace || workspace_bytes < pfpn_rollout_workspace_bytes(B, A, N)) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* lpseg = (float*)workspace;
    float* smax = lpseg + (size_t)B * A;
    float* sinv = smax + (size_t)B * A;
    float* partial = sinv + (size_t)B * A * 4;
    const int S = B * A;
    if (N <= 64)
        hipLaunchKernelGGL(k_sample_wave, dim3((S + 3) / 4), dim3(256), 0, st, logits, loc, logstd, u_cat, eps_sel, B, A,
                           N, (int*)k, action, lpseg, smax, sinv);
    else
        hipLaunchKernelGGL(k_sample_plain, dim3((S + 255) / 256), dim3(256), 0, st, logits, loc, logstd, u_cat,
                           eps_sel, B, A, N, (int*)k, action, lpseg, smax, sinv);
    PFPN_CHECK_LAUNCH();
    if (logp) {
        hipLaunchKernelGGL(k_row_sum2, dim3((B * 64 + 255) / 256), dim3(256), 0, st, lpseg, B, A, logp);
        PFPN_CHECK_LAUNCH();
    }
    if (max_active) return stats_launch(logits, smax, sinv, B, A, N, partial, max_active, sum_active, st);
    return PFPN_OK;
}

int pfpn_active_stats(const float* logits, int B, int A, int N, float* max_active, float* sum_active,
                      void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!logits || !max_active || !sum_active || B < 0 || A < 1 || N < 1) return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    if (!workspace || workspace_bytes < pfpn_rollout_workspace_bytes(B, A, N)) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* smax = (float*)workspace + (size_t)B * A;
    float* sinv = smax + (size_t)B * A;
    float* partial = sinv + (size_t)B * A * 4;
    const int S = B * A;
    hipLaunchKernelGGL(k_softmax_norm, dim3((S + 255) / 256), dim3(256), 0, st, logits, S, N, smax, sinv);
    PFPN_CHECK_LAUNCH();
    return stats_launch(logits, smax, sinv, B, A, N, partial, max_active, sum_active, st);
}

int pfpn_mean_action(const float* logits, const float* loc, int B, int A, int N, int tanh_mode,
                     int32_t* k, float* action, void* stream) {
    PFPN_BEGIN();
    if (!logits || !loc || !action || B < 0 || A < 1 || N < 1) return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    const int S = B * A;
    hipLaunchKernelGGL(k_mean_action, dim3((S + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, loc, B,
                       A, N, tanh_mode, (int*)k, action);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_entropy_fwd(const float* logits, int B, int A, int N, float* H, void* stream) {
    PFPN_BEGIN();
    if (!logits || !H || B < 0 || A < 1 || N < 1) return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    const int S = B * A;
    hipLaunchKernelGGL(k_entropy_fwd, dim3((S + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, S, N, H);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_entropy_bwd(const float* logits, const float* dH, int B, int A, int N, int accumulate,
                     float* dlogits, void* stream) {
    PFPN_BEGIN();
    if (!logits || !dH || !dlogits || B < 0 || A < 1 || N < 1) return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    const int S = B * A;
    hipLaunchKernelGGL(k_entropy_bwd, dim3((S + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, dH, S, N,
                       accumulate, dlogits);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_sac_noise_from_uniform(float* u, size_t n, void* stream) {
    PFPN_BEGIN();
    if (!u) return PFPN_ERR_ARG;
    if (n == 0) return PFPN_OK;
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_sac_noise_inplace, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, u, n);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_sac_noise_fill(float* buf, size_t n, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream) {
    PFPN_BEGIN();
    if (!buf) return PFPN_ERR_ARG;
    if (n == 0) return PFPN_OK;
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_sac_noise_fill, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, buf, n,
                       (unsigned long long)seed, (unsigned long long)offset, (const unsigned long long*)offset_dev);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_gumbel_from_uniform(float* u, size_t n, void* stream) {
    PFPN_BEGIN();
    if (!u) return PFPN_ERR_ARG;
    if (n == 0) return PFPN_OK;
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_gumbel_inplace, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, u, n);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_rsample_fwd(const float* logits, const float* loc, const float* logstd, const float* gumbel,
                     const float* eps, int B, int A, int N, int32_t* k, float* a_out, float* s_out,
                     void* stream) {
    PFPN_BEGIN();
    if (!logits || !loc || !logstd || !gumbel || !eps || !k || !a_out || !s_out || B < 0 || A < 1 || N < 1)
        return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    const int S = B * A;
    if (N <= RS_G * RS_NP && S >= RS_STAGE_MIN_SEGS)
        hipLaunchKernelGGL(k_rsample_fwd_quad<true>, dim3((S + 4 * RS_SEGS - 1) / (4 * RS_SEGS)), dim3(256),
                           (size_t)4 * 2 * ((RS_SEGS * N + 3) & ~3) * sizeof(float), (hipStream_t)stream, logits,
                           loc, logstd, gumbel, eps, B, A, N, (int*)k, a_out, s_out);
    else if (N <= RS_G * RS_NP)
        hipLaunchKernelGGL(k_rsample_fwd_quad<false>, dim3((S + 4 * RS_SEGS - 1) / (4 * RS_SEGS)), dim3(256), 0,
                           (hipStream_t)stream, logits, loc, logstd, gumbel, eps, B, A, N, (int*)k, a_out, s_out);
    else
        hipLaunchKernelGGL(k_rsample_fwd, dim3((S + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, loc,
                           logstd, gumbel, eps, B, A, N, (int*)k, a_out, s_out);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_rsample_bwd(const float* logits, const float* loc, const float* logstd, const float* gumbel,
                     const float* eps, const float* da, const float* ds, int B, int A, int N, int accumulate,
                     float* dlogits, float* dloc, float* dlogstd, int finalize, void* workspace,
                     size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!logits || !loc || !logstd || !gumbel || !eps || !da || !ds || !dlogits || (finalize && (!dloc || !dlogstd)) ||
        B < 0 || A < 1 || N < 1 || (!finalize && B == 0))
        return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    if (!workspace || workspace_bytes < pfpn_rollout_workspace_bytes(B, A, N)) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* gp = (float*)workspace;
    float* gs = gp + (size_t)B * A;
    int* kt = (int*)(gs + (size_t)B * A);
    const int S = B * A;
    if (N <= RS_G * RS_NP && S >= RS_STAGE_MIN_SEGS)
        hipLaunchKernelGGL(k_rsample_bwd_quad<true>, dim3((S + 4 * RS_SEGS - 1) / (4 * RS_SEGS)), dim3(256),
                           (size_t)4 * 4 * ((RS_SEGS * N + 3) & ~3) * sizeof(float), st, logits, loc, logstd,
                           gumbel, eps, da, ds, B, A, N, accumulate, dlogits, kt, gp, gs);
    else if (N <= RS_G * RS_NP)
        hipLaunchKernelGGL(k_rsample_bwd_quad<false>, dim3((S + 4 * RS_SEGS - 1) / (4 * RS_SEGS)), dim3(256), 0, st,
                           logits, loc, logstd, gumbel, eps, da, ds, B, A, N, accumulate, dlogits, kt, gp, gs);
    else
        hipLaunchKernelGGL(k_rsample_bwd, dim3((S + 255) / 256), dim3(256), 0, st, logits, loc, logstd, gumbel, eps,
                           da, ds, B, A, N, accumulate, dlogits, kt, gp, gs);
    PFPN_CHECK_LAUNCH();
    // parameter scatter: <= 256 row chunks (multiples of 64 rows) x A waves, then a fixed-order slab sum
    int C = (B + 63) / 64;
    if (C > 256) C = 256;
    const int len = (((B + C - 1) / C) + 63) / 64 * 64;
    C = (B + len - 1) / len;
    float* partial = (float*)workspace + (size_t)B * A * 6;
    hipLaunchKernelGGL(k_rsample_param_partial, dim3((C * A * 64 + 255) / 256), dim3(256), 0, st, kt, gp, gs, B, A, N,
                       C, len, partial);
    PFPN_CHECK_LAUNCH();
    if (finalize) {
        hipLaunchKernelGGL(k_param_grad_final, dim3((2 * A * N + 63) / 64), dim3(1024), 0, st, partial, C, A * N,
                           accumulate, dloc, dlogstd);
        PFPN_CHECK_LAUNCH();
    }
    return PFPN_OK;
}

// Slab geometry of pfpn_rsample_bwd(finalize = 0): `nslab` slabs of `stride` floats [dloc A*N | dlogstd A*N] starting
// `float_offset` floats into its workspace; the caller adds their fixed-order sum to the particle-grid gradients.
int pfpn_rsample_bwd_slabs(int B, int A, int N, int* nslab, int* stride, long long* float_offset) {
    if (!nslab || !stride || !float_offset || B < 1 || A < 1 || N < 1) return PFPN_ERR_ARG;
    int C = (B + 63) / 64;
    if (C > 256) C = 256;
    const int len = (((B + C - 1) / C) + 63) / 64 * 64;
    *nslab = (B + len - 1) / len;
    *stride = 2 * A * N;
    *float_offset = (long long)B * A * 6;
    return PFPN_OK;
}

}  // extern "C"
