// Learner-side utility kernels for gfx950: GAE scan (K6), PPO loss head (K7, unfused form),
// state normaliser + EMA moments (K8), dense-layer epilogues (bias+ReLU6 and its backward with
// the bias-gradient column sum), the scalar value head, global-norm clip + TF-Adam (K9), SAC
// soft target update.  All HBM-bound streaming kernels: float4 where rows allow, column
// reductions through fixed-order chunk slabs (bitwise reproducible).
#include "pfpn_common.h"
#include "pfpn_hip.h"     // PfpnReduceJob; also checks every definition below against its declaration

namespace pfpn {

// ---------------------------------------------------------------------------------------
// K6: GAE.  One block per sequence.  delta_t = (r_t + gamma*v_{t+1}) - v_t in fp32 (two roundings,
// the reference's NumPy op order, a2c.py:32-35), then adv_t = delta_t + c*adv_{t+1} with a float64
// accumulator walked sequentially by one lane (utils.py:5-15 under NumPy<2 promotion): the same
// operation order as the reference loop, so the fp32 output is bit-exact.
// ---------------------------------------------------------------------------------------
constexpr int GAE_CHUNK = 2048;

__global__ __launch_bounds__(256) void k_gae(const float* __restrict__ reward, const float* __restrict__ value,
                                             const int64_t* __restrict__ offsets, int E, float gamma,
                                             double gae_gamma, float* __restrict__ adv) {
    __shared__ float delta[GAE_CHUNK];
    __shared__ double carry;
    const int e = blockIdx.x;
    const int64_t r0 = offsets[e];
    const int T = (int)(offsets[e + 1] - r0);
    const float* r = reward + r0;
    const float* v = value + r0 + e;          // sequence e owns T+1 values
    float* out = adv + r0;
    if (threadIdx.x == 0) carry = 0.0;
    for (int hi = T; hi > 0; hi -= GAE_CHUNK) {
        const int lo = max(0, hi - GAE_CHUNK);
        __syncthreads();
        for (int t = lo + threadIdx.x; t < hi; t += blockDim.x) {
            const float gv = gamma * v[t + 1];
            const float s = r[t] + gv;
            delta[t - lo] = s - v[t];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // the recurrence is sequential by definition (the reference's Python loop, float64 accumulator: same order,
            // same bits); what need not be sequential are the LDS accesses -- 16 deltas per round trip instead of one
            // dependent LDS read per step (27 -> ~6 us for 512 steps)
            double acc = carry;
            int t = hi - 1 - lo;                              // index into delta
            for (; t >= 15; t -= 16) {
                float d[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) d[j] = delta[t - j];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const double m = gae_gamma * acc;
                    acc = (double)d[j] + m;
                    d[j] = (float)acc;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) delta[t - j] = d[j];
            }
            for (; t >= 0; --t) {
                const double m = gae_gamma * acc;
                acc = (double)delta[t] + m;
                delta[t] = (float)acc;
            }
            carry = acc;
        }
        __syncthreads();
        for (int t = lo + threadIdx.x; t < hi; t += blockDim.x) out[t] = delta[t - lo];
    }
}

// ---------------------------------------------------------------------------------------
// K7 (unfused): PPO loss head on [B] vectors; one block, fixed order.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_ppo_loss(
    const float* __restrict__ logp, const float* __restrict__ logp_old, const float* __restrict__ adv,
    const float* __restrict__ value, const float* __restrict__ value_old, const float* __restrict__ stats,
    int B, float clip_lo, float clip_hi, float value_coef,
    float* __restrict__ dlogp, float* __restrict__ dvalue, float* __restrict__ loss3)
{
    __shared__ float red[16];
    const float mean = stats[0], rstd = stats[1], alo = stats[2], ahi = stats[3];
    const float inv_B = 1.0f / (float)B;
    float ps = 0.f, vs = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float a = adv[b];
        const float nadv = clip_adv((a - mean) * rstd, alo, ahi);
        const float ratio = expf(logp[b] - logp_old[b]);
        const float sur = ratio * nadv;
        const float cr = fmaxf(fminf(ratio, clip_hi), clip_lo);
        const float csur = cr * nadv;
        ps += fminf(sur, csur);
        const bool pass = (sur <= csur) || (ratio >= clip_lo && ratio <= clip_hi);
        dlogp[b] = pass ? (-inv_B * nadv) * ratio : 0.f;
        const float dvv = value[b] - (a + value_old[b]);
        vs += dvv * dvv;
        dvalue[b] = 2.f * dvv * value_coef * inv_B;
    }
    ps = block_sum(ps, red);
    vs = block_sum(vs, red);
    if (threadIdx.x == 0) {
        const float pl = -ps * inv_B, vl = vs * inv_B * value_coef;
        loss3[0] = pl + vl; loss3[1] = pl; loss3[2] = vl;
    }
}

// ---------------------------------------------------------------------------------------
// V-trace (reference networks/actor_critic/vtrace.py:101-180).  E episodes of T steps, data flat in
// episode-major order (index e*T + t, the order the workers feed: vtrace.py:16-37).  Every episode's T steps are
// walked backwards exactly as the reference's reverse tf.scan does, in fp32 and in the graph's operation order:
//   rho = exp(lp_target - lp_behaviour); v_rho = min(rho_clip, rho); c = min(1, rho); disc = gamma*not_terminal
//   value_t1[t] = value[t+1], and value[t] itself at the last step / at t = seq_len-1 (concat_bootstrap_value)
//   a[t] = v_rho*((r + disc*value_t1) - value) + (disc'*c)*a[t+1],  disc' = 0 at t = seq_len-1;  vs = a + value
//   pg_adv[t] = pg_rho*((r + disc*vs_t1) - value), vs_t1 = vs[t+1] (value[t] at the last step / episode end)
// Clips <= 0 mean "None" (no clipping).  mask[t] = t < seq_len (all ones without seq_len).
// ---------------------------------------------------------------------------------------
// One WAVE per episode (round 2; before: one thread per episode, five dependent global loads per step -- 20-30 us for 64
// steps): the lanes take 64 consecutive steps, everything that does not depend on the recurrence (ratio, clips, discounts,
// the temporal difference, the coefficient of a[t+1]) is computed once per step in parallel from coalesced loads, and only
// a[t] = td[t] + k[t] * a[t+1] walks the lanes from 63 down to 0 on wave-uniform values -- the same fp32 multiply and add per
// step, in the same order, as the serial loop (and as the reference's reverse tf.scan).
__global__ __launch_bounds__(256) void k_vtrace(const float* __restrict__ lp_t, const float* __restrict__ lp_b,
                                                const float* __restrict__ value, const float* __restrict__ reward,
                                                const float* __restrict__ not_terminal,
                                                const int* __restrict__ seq_len, int E, int T, float gamma,
                                                float rho_clip, float pg_rho_clip, float* __restrict__ vs,
                                                float* __restrict__ pg_adv, float* __restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const int e = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (e >= E) return;
    const int end = seq_len ? seq_len[e] - 1 : -1;           // episodic_end_idx within the episode
    const size_t base = (size_t)e * T;
    float a_next = 0.f, vs_next = 0.f;                       // a[t+1], vs[t+1] of the chunk handled before (later in time)
    for (int hi = T; hi > 0; hi -= 64) {
        const int t = hi - 64 + lane;                        // my step (may be < 0 in the first, ragged chunk)
        const bool live = t >= 0;
        const size_t i = base + (live ? t : 0);
        const float v = value[i], r = reward[i];
        const float rho = expf(lp_t[i] - lp_b[i]);
        const float v_rho = rho_clip > 0.f ? fminf(rho_clip, rho) : rho;
        const float pg_rho = pg_rho_clip > 0.f ? fminf(pg_rho_clip, rho) : rho;
        const float c = fminf(1.0f, rho);
        const float disc = gamma * not_terminal[i];
        const bool boot = (t == T - 1) || (t == end);
        const float v_t1 = (boot || !live) ? v : value[i + 1];
        const float td = v_rho * ((r + disc * v_t1) - v);
        const float dsc = (t == end) ? 0.f : disc;
        const float kc = dsc * c;
        // the recurrence, lane 63 (latest step) first
        float a = 0.f;
        for (int j = 63; j >= 0; --j) {
            if (hi - 64 + j < 0) break;                      // (wave-uniform)
            const float td_j = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(td), j));
            const float kc_j = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kc), j));
            a_next = td_j + kc_j * a_next;
            if (lane == j) a = a_next;
        }
        const float vs_t = a + v;
        // vs[t+1]: my right neighbour's, or the previous chunk's first step for lane 63
        float vs_r = __shfl_down(vs_t, 1, 64);
        if (lane == 63) vs_r = vs_next;
        const float vs_t1 = boot ? v : vs_r;
        if (live) {
            vs[i] = vs_t;
            pg_adv[i] = pg_rho * ((r + disc * vs_t1) - v);
            if (mask) mask[i] = (!seq_len || t <= end) ? 1.f : 0.f;
        }
        // the chunk's first live step feeds the next (earlier) chunk
        const int first = max(0, 64 - hi);
        vs_next = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vs_t), first));
    }
}

// ---------------------------------------------------------------------------------------
// Policy-gradient (A2C / A3C / IMPALA) loss head on [B] vectors -- reference a2c.py:53-65,
// actor_critic.py:132-184:  policy_loss = -mean_valid(logp * adv_n),  value_loss = coef * mean_valid((v - vt)^2),
// adv_n = (adv - mean)*rstd (stats = {0,1}: off).  mask (optional, 0/1) is valid_data_mask; means run over the
// valid rows.  dlogp = -adv_n*mask/n, dvalue = 2*coef*(v - vt)*mask/n.  One block, fixed order.
// loss3 = {loss, policy_loss, value_loss*coef}; n_valid is returned in loss3[3].
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_a2c_loss(const float* __restrict__ logp, const float* __restrict__ adv,
                                                   const float* __restrict__ value,
                                                   const float* __restrict__ value_target,
                                                   const float* __restrict__ stats, const float* __restrict__ mask,
                                                   int B, float value_coef, float* __restrict__ dlogp,
                                                   float* __restrict__ dvalue, float* __restrict__ loss4) {
    __shared__ float red[16];
    float nv = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) nv += mask ? mask[b] : 1.f;
    nv = block_sum(nv, red);
    const float inv_n = 1.0f / nv;
    const float mean = stats[0], rstd = stats[1], alo = stats[2], ahi = stats[3];
    float ps = 0.f, vs = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float m = mask ? mask[b] : 1.f;
        const float nadv = clip_adv((adv[b] - mean) * rstd, alo, ahi);
        ps += m * (logp[b] * nadv);
        const float d = value[b] - value_target[b];
        vs += m * (d * d);
        dlogp[b] = -(nadv * m) * inv_n;
        dvalue[b] = (2.f * value_coef) * (d * m) * inv_n;
    }
    ps = block_sum(ps, red);
    vs = block_sum(vs, red);
    if (threadIdx.x == 0) {
        const float pl = -ps * inv_n, vl = vs * inv_n * value_coef;
        loss4[0] = pl + vl; loss4[1] = pl; loss4[2] = vl; loss4[3] = nv;
    }
}

// ---------------------------------------------------------------------------------------
// K8: state normaliser
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_state_norm(const float* __restrict__ s, const float* __restrict__ mean,
                                                    const float* __restrict__ std, int B, int S, float clip,
                                                    float* __restrict__ x) {
    const size_t n = (size_t)B * S;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % S);
        float v = (s[i] - mean[c]) / std[c];                      // actor_critic.py:229-232
        if (clip > 0.f) v = fmaxf(fminf(v, clip), -clip);         // clip_by_value: Minimum then Maximum
        x[i] = v;
    }
}

// Batch moments for the state-normaliser EMA (utils.py:60-68), one pass: per 16-row chunk the column
// sums of x and x^2 in float64 (mean and the biased variance of the batch are then exact to fp32
// rounding; the reference's two-pass fp32 form agrees to ~1e-7).  256-thread block = 64 columns x
// 4 row lanes, combined through LDS in lane order; partial[ch][0|1][S] doubles.
__global__ __launch_bounds__(256) void k_col_moments(const float* __restrict__ s, int B, int S, int chunk_len,
                                                     double* __restrict__ partial) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int ch = blockIdx.y;
    const int b0 = ch * chunk_len, b1 = min(B, b0 + chunk_len);
    double a1 = 0.0, a2 = 0.0;
    if (c < S) {
        for (int b = b0 + rl; b < b1; b += 16) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = b + 4 * j;
                v[j] = r < b1 ? s[(size_t)r * S + c] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double d = (double)v[j]; a1 += d; a2 += d * d; }
        }
    }
    red[0][rl][cl] = a1; red[1][rl][cl] = a2;
    __syncthreads();
    if (rl == 0 && c < S) {
        double s1 = red[0][0][cl], s2 = red[1][0][cl];
        for (int k = 1; k < 4; ++k) { s1 += red[0][k][cl]; s2 += red[1][k][cl]; }
        partial[((size_t)ch * 2 + 0) * S + c] = s1;
        partial[((size_t)ch * 2 + 1) * S + c] = s2;
    }
}

// Step prologue in ONE launch (the learner's first four launches were latency-bound ~5 us each):
//   blocks (x, ch < C): x = clip((s - mean)/std) for a 16-row chunk of 64 columns (k_state_norm's arithmetic) AND
//                       that chunk's float64 column moments (k_col_moments' arithmetic and layout);
//   block  (0, C)     : advantage moments {mean, 1/(std + 1e-8)} (k_adv_stats' arithmetic, 256 threads), optional.
__device__ __forceinline__ void state_prologue_block(const float* __restrict__ s, const float* __restrict__ mean,
                                                     const float* __restrict__ std, int B, int S, int chunk_len,
                                                     int C, float clip, float* __restrict__ x,
                                                     double* __restrict__ partial, const float* __restrict__ adv,
                                                     int normalize_adv, float* __restrict__ adv_stats,
                                                     const float* __restrict__ s2, float* __restrict__ x2, int bx, int by) {
    __shared__ double red[2][4][64];
    if (by == C) {                                         // the advantage block
        if (bx != 0 || !adv_stats) return;
        float* fred = reinterpret_cast<float*>(&red[0][0][0]);
        if (!normalize_adv) {
            if (threadIdx.x == 0) { adv_stats[0] = 0.f; adv_stats[1] = 1.f; }
            return;
        }
        // the first 16 values per thread (all of them up to B = 4096) are loaded at once and kept for the second pass
        float a[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const int i = threadIdx.x + j * 256; a[j] = i < B ? adv[i] : 0.f; }
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) if ((int)threadIdx.x + j * 256 < B) sm += a[j];
        for (int i = threadIdx.x + 4096; i < B; i += blockDim.x) sm += adv[i];
        const float mu = block_sum(sm, fred) / (float)B;
        float vv = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) if ((int)threadIdx.x + j * 256 < B) { const float d = a[j] - mu; vv = fmaf(d, d, vv); }
        for (int i = threadIdx.x + 4096; i < B; i += blockDim.x) { const float d = adv[i] - mu; vv = fmaf(d, d, vv); }
        const float var = block_sum(vv, fred) / (float)B;
        if (threadIdx.x == 0) { adv_stats[0] = mu; adv_stats[1] = 1.0f / (sqrtf(var) + 1e-8f); }
        return;
    }
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = bx * 64 + cl;
    const int ch = by;
    const int b0 = ch * chunk_len, b1 = min(B, b0 + chunk_len);
    double a1 = 0.0, a2 = 0.0;
    if (c < S) {
        const float mc = mean[c], sc = std[c];
        for (int b = b0 + rl; b < b1; b += 16) {
            float v[4], w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = b + 4 * j;
                v[j] = r < b1 ? s[(size_t)r * S + c] : 0.f;
                w[j] = (s2 && r < b1) ? s2[(size_t)r * S + c] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = b + 4 * j;
                const double d = (double)v[j];
                a1 += d; a2 += d * d;                                  // zero rows beyond the chunk add nothing
                if (r < b1) {
                    float xn = (v[j] - mc) / sc;                       // actor_critic.py:229-232
                    if (clip > 0.f) xn = fmaxf(fminf(xn, clip), -clip);
                    x[(size_t)r * S + c] = xn;
                    if (s2) {                                          // a second batch normalised with the same moments
                        float yn = (w[j] - mc) / sc;                   // (SAC: the next states), k_state_norm's arithmetic
                        if (clip > 0.f) yn = fmaxf(fminf(yn, clip), -clip);
                        x2[(size_t)r * S + c] = yn;
                    }
                }
            }
        }
    }
    if (!partial) return;
    red[0][rl][cl] = a1; red[1][rl][cl] = a2;
    __syncthreads();
    if (rl == 0 && c < S) {
        double s1 = red[0][0][cl], s2 = red[1][0][cl];
        for (int k = 1; k < 4; ++k) { s1 += red[0][k][cl]; s2 += red[1][k][cl]; }
        partial[((size_t)ch * 2 + 0) * S + c] = s1;
        partial[((size_t)ch * 2 + 1) * S + c] = s2;
    }
}

__global__ __launch_bounds__(256) void k_state_prologue(const float* __restrict__ s, const float* __restrict__ mean,
                                                        const float* __restrict__ std, int B, int S, int chunk_len,
                                                        int C, float clip, float* __restrict__ x,
                                                        double* __restrict__ partial, const float* __restrict__ adv,
                                                        int normalize_adv, float* __restrict__ adv_stats,
                                                        const float* __restrict__ s2, float* __restrict__ x2) {
    state_prologue_block(s, mean, std, B, S, chunk_len, C, clip, x, partial, adv, normalize_adv, adv_stats, s2, x2,
                         (int)blockIdx.x, (int)blockIdx.y);
}

// Chunk moments -> batch mean / variance -> EMA update, utils.py:60-68: d = min(0.9999, (1+step)/(10+step)).
// 1024-thread block = 64 columns x 16 chunk lanes (fixed order: deterministic).
__global__ __launch_bounds__(1024) void k_norm_ema(const double* __restrict__ partial, int C, int B, int S,
                                                   const float* __restrict__ step_dev,
                                                   const float* __restrict__ mean_in,
                                                   const float* __restrict__ std_in, float* __restrict__ mean_out,
                                                   float* __restrict__ std_out) {
    __shared__ double red[2][16][64];
    const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double a1 = 0.0, a2 = 0.0;
    if (c < S) {
        int k = kl;
        for (; k + 112 < C; k += 128) {                   // 16 loads in flight; additions stay in k order
            double u[8], w[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                u[m] = partial[((size_t)(k + 16 * m) * 2 + 0) * S + c];
                w[m] = partial[((size_t)(k + 16 * m) * 2 + 1) * S + c];
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) { a1 += u[m]; a2 += w[m]; }
        }
        for (; k < C; k += 16) {
            a1 += partial[((size_t)k * 2 + 0) * S + c];
            a2 += partial[((size_t)k * 2 + 1) * S + c];
        }
    }
    red[0][kl][cl] = a1; red[1][kl][cl] = a2;
    __syncthreads();
    if (kl != 0 || c >= S) return;
    for (int j = 1; j < 16; ++j) { a1 += red[0][j][cl]; a2 += red[1][j][cl]; }
    const double mu = a1 / (double)B;
    double var = a2 / (double)B - mu * mu;
    if (var < 0.0) var = 0.0;
    const float bmean = (float)mu, bvar = (float)var;
    const float step = step_dev[0];                                // global_step as fp32 (device scalar)
    const float d = fminf(0.9999f, (1.f + step) / (10.f + step));
    const float omd = 1.f - d;
    mean_out[c] = d * mean_in[c] + omd * bmean;
    std_out[c] = fmaxf(1e-6f, d * std_in[c] + omd * sqrtf(bvar));
}

// ---------------------------------------------------------------------------------------
// dense-layer epilogues.  y is [rows, H]; bias row = (row / group_rows) when group_rows > 0
// (batched layers stacked as [G*rows, H] with bias [G, H]).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bias_act(float* __restrict__ y, const float* __restrict__ bias,
                                                  int rows, int H, int group_rows, int relu6) {
    const int H4 = H >> 2;
    const size_t n4 = (size_t)rows * H4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // four independent float4 streams per thread so ~4 KiB per wave is in flight
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 4 * stride) {
        float4 v[4], bb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = i0 + j * stride;
            const size_t ic = i < n4 ? i : i0;
            const int row = (int)(ic / H4);
            const int c4 = (int)(ic - (size_t)row * H4);
            const int g = group_rows > 0 ? row / group_rows : 0;
            v[j] = reinterpret_cast<float4*>(y)[ic];
            bb[j] = reinterpret_cast<const float4*>(bias + (size_t)g * H)[c4];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = i0 + j * stride;
            float4 t = v[j];
            t.x += bb[j].x; t.y += bb[j].y; t.z += bb[j].z; t.w += bb[j].w;
            if (relu6) {
                t.x = fminf(fmaxf(t.x, 0.f), 6.f); t.y = fminf(fmaxf(t.y, 0.f), 6.f);
                t.z = fminf(fmaxf(t.z, 0.f), 6.f); t.w = fminf(fmaxf(t.w, 0.f), 6.f);
            }
            if (i < n4) reinterpret_cast<float4*>(y)[i] = t;
        }
    }
}

// Relu6Grad + bias gradient: dy <- dy * [0 < y < 6] (in place when relu6), partial column sums
// of the masked dy over a chunk of rows.  256-thread block = 64 float4 column groups x 4 row
// lanes (each wave streams whole 1 KiB row pieces); the 4 row lanes are combined through LDS in
// lane order, so the result is deterministic.
__global__ __launch_bounds__(256) void k_act_bwd_partial(const float* __restrict__ y, float* __restrict__ dy,
                                                         int grows, int H, int chunk_len, int relu6,
                                                         float* __restrict__ partial) {
    __shared__ float4 red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + cl;
    const int H4 = H >> 2;
    const int ch = blockIdx.y;
    const size_t goff = (size_t)blockIdx.z * grows * H4;          // group g owns rows [g*grows, (g+1)*grows)
    const int b0 = ch * chunk_len, b1 = min(grows, b0 + chunk_len);
    const float4* y4 = reinterpret_cast<const float4*>(y) + goff;
    float4* d4 = reinterpret_cast<float4*>(dy) + goff;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < H4) {
        // wave rl owns rows b0+rl, b0+rl+4, ...; four of them are loaded before any is consumed
        for (int b = b0 + rl; b < b1; b += 16) {
            float4 g[4], v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = b + 4 * j;
                const size_t i = (size_t)(r < b1 ? r : b) * H4 + c4;
                g[j] = d4[i];
                if (relu6) v[j] = y4[i];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = b + 4 * j;
                if (r < b1) {
                    float4 t = g[j];
                    if (relu6) {
                        t.x = (v[j].x > 0.f && v[j].x < 6.f) ? t.x : 0.f;
                        t.y = (v[j].y > 0.f && v[j].y < 6.f) ? t.y : 0.f;
                        t.z = (v[j].z > 0.f && v[j].z < 6.f) ? t.z : 0.f;
                        t.w = (v[j].w > 0.f && v[j].w < 6.f) ? t.w : 0.f;
                        d4[(size_t)r * H4 + c4] = t;
                    }
                    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                }
            }
        }
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c4 < H4) {
        float4 s = red[0][cl];
        for (int k = 1; k < 4; ++k) { const float4 t = red[k][cl]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
        reinterpret_cast<float4*>(partial + ((size_t)blockIdx.z * gridDim.y + ch) * H)[c4] = s;
    }
}

// db[g, c] = sum over the chunks of group g, fixed order.  Grid (ceil(H/64), G), block 1024.
// With `vec` (value head): one extra block (blockIdx.x == gridDim.x-1) emits vec_out[0] = sum(vec[0..nvec)).
__global__ __launch_bounds__(1024) void k_bias_grad_final(const float* __restrict__ partial, int chunks_per_group,
                                                          int G, int H, float* __restrict__ db,
                                                          const float* __restrict__ vec = nullptr, int nvec = 0,
                                                          float* __restrict__ vec_out = nullptr) {
    __shared__ float red[16][64];
    if (vec && blockIdx.x == gridDim.x - 1) {
        float s = 0.f;
        for (int i = threadIdx.x; i < nvec; i += blockDim.x) s += vec[i];
        s = block_sum(s, &red[0][0]);
        if (threadIdx.x == 0) vec_out[0] = s;
        return;
    }
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = blockIdx.y;
    const float s = slab_reduce_1024<false>(partial + (size_t)g * chunks_per_group * H, chunks_per_group, (size_t)H,
                                            c, c < H, red);
    if ((threadIdx.x >> 6) == 0 && c < H) db[(size_t)g * H + c] = s;
}

// scalar value head: v[b] = h[b,:] . w + bias   (critic fc3, squeeze).  One wave per row.
__global__ __launch_bounds__(256) void k_value_head_fwd(const float* __restrict__ h, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int B, int H,
                                                        float* __restrict__ v) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= B) return;
    const float4* hr = reinterpret_cast<const float4*>(h + (size_t)row * H);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    float acc = 0.f;
    for (int i = lane; i < (H >> 2); i += 64) {
        const float4 a = hr[i], b = w4[i];
        acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) v[row] = acc + bias[0];
}

// backward of the value head fused with the ReLU6 mask of the layer below:
// dh[b,c] = dv[b]*w[c]*[0<h<6];  partial dw[c] = sum_b h[b,c]*dv[b] over a chunk of rows.
__global__ __launch_bounds__(256) void k_value_head_bwd_partial(
    const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ dv,
    int B, int H, int chunk_len, int relu6, float* __restrict__ dh, float* __restrict__ partial)
{
    __shared__ float4 red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + cl;
    const int H4 = H >> 2;
    const int ch = blockIdx.y;
    const int b0 = ch * chunk_len, b1 = min(B, b0 + chunk_len);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < H4) {
        const float4 wv = reinterpret_cast<const float4*>(w)[c4];
        for (int b = b0 + rl; b < b1; b += 4) {
            const size_t i = (size_t)b * H4 + c4;
            const float4 v = reinterpret_cast<const float4*>(h)[i];
            const float g = dv[b];
            acc.x += v.x * g; acc.y += v.y * g; acc.z += v.z * g; acc.w += v.w * g;
            float4 d = make_float4(g * wv.x, g * wv.y, g * wv.z, g * wv.w);
            if (relu6) {
                d.x = (v.x > 0.f && v.x < 6.f) ? d.x : 0.f;
                d.y = (v.y > 0.f && v.y < 6.f) ? d.y : 0.f;
                d.z = (v.z > 0.f && v.z < 6.f) ? d.z : 0.f;
                d.w = (v.w > 0.f && v.w < 6.f) ? d.w : 0.f;
            }
            reinterpret_cast<float4*>(dh)[i] = d;
        }
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c4 < H4) {
        float4 s = red[0][cl];
        for (int k = 1; k < 4; ++k) { const float4 t = red[k][cl]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
        reinterpret_cast<float4*>(partial + (size_t)ch * H)[c4] = s;
    }
}

// ---------------------------------------------------------------------------------------
// K9: global-norm clip + TF-Adam on one flat parameter buffer
// ---------------------------------------------------------------------------------------
// per-thread sum of squares over the grid-stride float4 stream, eight loads in flight (a load per iteration made the
// kernel a chain of dependent HBM round trips); additions stay in stream order, so the sum is the plain loop's.
__device__ __forceinline__ float sqsum_stream(const float* __restrict__ g, size_t n4) {
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = g4[i + j * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    }
    for (; i < n4; i += stride) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    return s;
}

__global__ __launch_bounds__(256) void k_sqsum_partial(const float* __restrict__ g, size_t n,
                                                       float* __restrict__ partial) {
    __shared__ float red[4];
    const size_t n4 = n >> 2;
    float s = sqsum_stream(g, n4);
    if (blockIdx.x == 0)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// norm = sqrt(2 * sum(0.5 * |g|^2)) ; scale = c * min(1/norm, 1/c), NaN when norm is not finite
// (tf.clip_by_global_norm, reference models/workers/base_worker.py:97-102)
__global__ __launch_bounds__(256) void k_clip_scale(const float* __restrict__ partial, int nblk, float clip_norm,
                                                    float* __restrict__ norm_scale) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(2.f * (0.5f * s));
        float scale;
        if (clip_norm > 0.f) {
            scale = clip_norm * fminf(1.0f / norm, 1.0f / clip_norm);
            if (!isfinite(norm)) scale = NAN;
        } else {
            scale = 1.0f;
        }
        norm_scale[0] = norm;
        norm_scale[1] = scale;
    }
}

__global__ __launch_bounds__(256) void k_scale_inplace(float* __restrict__ g, size_t n,
                                                       const float* __restrict__ norm_scale) {
    const float sc = norm_scale[1];
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<float4*>(g)[i];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        reinterpret_cast<float4*>(g)[i] = v;
    }
    if (blockIdx.x == 0)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) g[i] *= sc;
}

// clip scale from the partial sums (k_clip_scale's arithmetic, in every block) + g *= scale, one launch
__global__ __launch_bounds__(256) void k_clip_scale_inplace(float* __restrict__ g, size_t n,
                                                            const float* __restrict__ partial, int nblk,
                                                            float clip_norm, float* __restrict__ norm_scale) {
    __shared__ float red[4];
    float sq = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) sq += partial[i];
    sq = block_sum(sq, red);
    const float norm = sqrtf(2.f * (0.5f * sq));
    float sc = 1.0f;
    if (clip_norm > 0.f) {
        sc = clip_norm * fminf(1.0f / norm, 1.0f / clip_norm);
        if (!isfinite(norm)) sc = NAN;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { norm_scale[0] = norm; norm_scale[1] = sc; }
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<float4*>(g)[i];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        reinterpret_cast<float4*>(g)[i] = v;
    }
    if (blockIdx.x == 0)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) g[i] *= sc;
}

// TF ApplyAdam (CPU kernel form): alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1);
// v += (g*g-v)(1-b2); p -= m*alpha/(sqrt(v)+eps).  g is first multiplied by the optional device
// scalar norm_scale[1] (clip) and by grad_mul (1/W after an all-reduce sum).
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float alpha, float omb1, float omb2,
                                      float eps) {
    m = m + (g - m) * omb1;
    v = v + (g * g - v) * omb2;
    p = p - (m * alpha) / (sqrtf(v) + eps);
}

__global__ __launch_bounds__(256) void k_tf_adam(float* __restrict__ p, const float* __restrict__ g,
                                                 float* __restrict__ m, float* __restrict__ v, size_t n,
                                                 float lr, float b1, float b2, float eps,
                                                 const float* __restrict__ beta_pow,
                                                 const float* __restrict__ norm_scale, float grad_mul) {
    const float alpha = lr * sqrtf(1.f - beta_pow[1]) / (1.f - beta_pow[0]);
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    const float gs = norm_scale ? norm_scale[1] : 1.f;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        if (norm_scale) { gg.x *= gs; gg.y *= gs; gg.z *= gs; gg.w *= gs; }
        if (grad_mul != 1.f) { gg.x *= grad_mul; gg.y *= grad_mul; gg.z *= grad_mul; gg.w *= grad_mul; }
        adam1(pp.x, gg.x, mm.x, vv.x, alpha, omb1, omb2, eps);
        adam1(pp.y, gg.y, mm.y, vv.y, alpha, omb1, omb2, eps);
        adam1(pp.z, gg.z, mm.z, vv.z, alpha, omb1, omb2, eps);
        adam1(pp.w, gg.w, mm.w, vv.w, alpha, omb1, omb2, eps);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0) {
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
            float gi = g[i];
            if (norm_scale) gi *= gs;
            if (grad_mul != 1.f) gi *= grad_mul;
            adam1(p[i], gi, m[i], v[i], alpha, omb1, omb2, eps);
        }
    }
}

// ---- one-replica optimiser step in TWO launches (clip folded into Adam) ---------------------------------------
// beta_pow[g] = {b1^t, b2^t for the NEXT apply ("next": the value of TF's beta*_power variables), and the pair the
// CURRENT apply uses ("used")}.  Launch 1 (k_sqsum_partial_rot) rotates used <- next, next <- next*beta in block 0
// -- nobody else touches the powers in that launch -- so launch 2 can read `used` from every block without racing
// the advance (one kernel doing both would).
struct AdamGroups {
    int n;                       // <= 4
    long long off[4], len[4];    // ranges of the flat buffers (floats, 4-aligned)
    float lr[4];
};

__global__ __launch_bounds__(256) void k_sqsum_partial_rot(const float* __restrict__ g, size_t n,
                                                           float* __restrict__ partial, float* __restrict__ beta_pow,
                                                           int n_groups, float b1, float b2) {
    __shared__ float red[4];
    const size_t n4 = n >> 2;
    float s = sqsum_stream(g, n4);
    if (blockIdx.x == 0)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s;
        if (blockIdx.x == 0)
            for (int k = 0; k < n_groups; ++k) {
                float* bp = beta_pow + 4 * k;
                bp[2] = bp[0]; bp[3] = bp[1];
                bp[0] *= b1; bp[1] *= b2;
            }
    }
}

__device__ __forceinline__ void clip_adam_block(float* __restrict__ p, const float* __restrict__ g,
                                                float* __restrict__ m, float* __restrict__ v, const AdamGroups& gr,
                                                float b1, float b2, float eps, const float* __restrict__ beta_pow,
                                                const float* __restrict__ partial, int nblk, float clip_norm,
                                                float grad_mul, float* __restrict__ norm_scale,
                                                float* __restrict__ step, float* __restrict__ scale_buf,
                                                size_t scale_n, float scale_mul, float* __restrict__ soft_t,
                                                size_t soft_o4, size_t soft_n4, float tau, unsigned bid, unsigned nblocks) {
    __shared__ float red[4];
    // data parallel: the replicated statistics behind the gradients in the all-reduce bucket turn from sums into
    // replica means here (sync_model.py:37-46 push_weights) instead of in a launch of their own
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < scale_n; i += (size_t)nblocks * blockDim.x)
        scale_buf[i] *= scale_mul;
    float gs = 1.0f;
    if (nblk > 0) {
        // every block forms the same scale from the same partial sums, in k_clip_scale's order
        float sq = 0.f;
        for (int i = threadIdx.x; i < nblk; i += blockDim.x) sq += partial[i];
        sq = block_sum(sq, red);
        const float norm = sqrtf(2.f * (0.5f * sq));
        if (clip_norm > 0.f) {
            gs = clip_norm * fminf(1.0f / norm, 1.0f / clip_norm);
            if (!isfinite(norm)) gs = NAN;
        }
        if (bid == 0 && threadIdx.x == 0 && norm_scale) { norm_scale[0] = norm; norm_scale[1] = gs; }
    }
    if (grad_mul != 1.f) gs *= grad_mul;        // 1/W after an all-reduce sum
    if (bid == 0 && threadIdx.x == 0 && step)
        step[0] += 1.0f;                        // float(global_step) of the state-normaliser decay; no reader in this launch
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    for (int k = 0; k < gr.n; ++k) {
        const float alpha = gr.lr[k] * sqrtf(1.f - beta_pow[4 * k + 3]) / (1.f - beta_pow[4 * k + 2]);
        const size_t o4 = (size_t)gr.off[k] >> 2, n4 = (size_t)gr.len[k] >> 2;
        for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < n4; i += (size_t)nblocks * blockDim.x) {
            float4 pp = reinterpret_cast<float4*>(p)[o4 + i];
            float4 gg = reinterpret_cast<const float4*>(g)[o4 + i];
            float4 mm = reinterpret_cast<float4*>(m)[o4 + i];
            float4 vv = reinterpret_cast<float4*>(v)[o4 + i];
            gg.x *= gs; gg.y *= gs; gg.z *= gs; gg.w *= gs;
            adam1(pp.x, gg.x, mm.x, vv.x, alpha, omb1, omb2, eps);
            adam1(pp.y, gg.y, mm.y, vv.y, alpha, omb1, omb2, eps);
            adam1(pp.z, gg.z, mm.z, vv.z, alpha, omb1, omb2, eps);
            adam1(pp.w, gg.w, mm.w, vv.w, alpha, omb1, omb2, eps);
            reinterpret_cast<float4*>(p)[o4 + i] = pp;
            reinterpret_cast<float4*>(m)[o4 + i] = mm;
            reinterpret_cast<float4*>(v)[o4 + i] = vv;
            // SAC soft target update (sac.py:67-73) of the parameters just written: t <- (1-tau) t + tau p, k_soft_update's
            // arithmetic, for the floats [4 soft_o4, 4 (soft_o4 + soft_n4)) of the buffer (soft_t holds their targets)
            const size_t e = o4 + i - soft_o4;
            if (soft_t && o4 + i >= soft_o4 && e < soft_n4) {
                const float omt = 1.f - tau;
                float4 t = reinterpret_cast<float4*>(soft_t)[e];
                t.x = omt * t.x + tau * pp.x; t.y = omt * t.y + tau * pp.y;
                t.z = omt * t.z + tau * pp.z; t.w = omt * t.w + tau * pp.w;
                reinterpret_cast<float4*>(soft_t)[e] = t;
            }
        }
    }
}


__global__ __launch_bounds__(256) void k_clip_adam(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, AdamGroups gr,
                                                   float b1, float b2, float eps, const float* __restrict__ beta_pow,
                                                   const float* __restrict__ partial, int nblk, float clip_norm,
                                                   float grad_mul, float* __restrict__ norm_scale,
                                                   float* __restrict__ step, float* __restrict__ scale_buf = nullptr,
                                                   size_t scale_n = 0, float scale_mul = 1.f,
                                                   float* __restrict__ soft_t = nullptr, size_t soft_o4 = 0,
                                                   size_t soft_n4 = 0, float tau = 0.f) {
    clip_adam_block(p, g, m, v, gr, b1, b2, eps, beta_pow, partial, nblk, clip_norm, grad_mul, norm_scale, step, scale_buf,
                    scale_n, scale_mul, soft_t, soft_o4, soft_n4, tau, blockIdx.x, gridDim.x);
}

// The optimiser launch of learner step k with the FIRST launch of step k+1 riding in it (inside a multi-step capture,
// pfpn_amd/graphs.py): blocks [0, adam_blocks) are k_clip_adam, the rest are k_state_prologue's blocks on the next step's
// batch.  The two touch disjoint data (the prologue reads the normaliser statistics the reduction launch BEFORE this one
// updated, and writes buffers -- normalised state, moment workspace, advantage moments -- whose last readers of step k ran
// before that launch too), so one launch less per step in a single chain of kernels.
// SAC (round 4): the rider may also carry a second batch for the prologue (the next states, normalised with the same
// moments), the soft target update rides in the optimiser part as in k_clip_adam, and a third block role draws the NEXT
// step's four noise tensors (sac_noise_element; the offset is the step's device-resident draw counter, which this step's
// loss head advanced in an earlier launch: no reader races its writer).  pro_blocks = 0: noise only (the last step of a replay).
struct PrologueRider {
    const float* s; const float* mean; const float* std;
    int B, S, chunk_len, C, gx;
    float clip;
    float* x; double* partial; const float* adv; int normalize_adv; float* adv_stats;
    const float* s2; float* x2;
    unsigned pro_blocks;
    float* noise; size_t noise_n; unsigned long long noise_seed; const unsigned long long* noise_epoch; unsigned noise_blocks;
};
__global__ __launch_bounds__(256) void k_clip_adam_prologue(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, AdamGroups gr,
                                                            float b1, float b2, float eps,
                                                            const float* __restrict__ beta_pow,
                                                            const float* __restrict__ partial, int nblk, float clip_norm,
                                                            float* __restrict__ norm_scale, float* __restrict__ step,
                                                            unsigned adam_blocks, PrologueRider r,
                                                            float* __restrict__ soft_t, size_t soft_o4, size_t soft_n4,
                                                            float tau) {
    // The draw's blocks come FIRST in the grid: they are arithmetic (Philox, logs, sin / cos: ~1,300 instructions per thread)
    // next to an optimiser pass that waits on memory; dispatched last they ran as its tail (the grid is larger than what
    // the chip holds at once), dispatched first they run under it.
    const unsigned nz = r.noise_blocks;
    if (blockIdx.x < nz) {
        const unsigned long long offset = r.noise_epoch[0];
        const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (size_t i = first; i < r.noise_n; i += (size_t)nz * blockDim.x)
            sac_noise_element(r.noise, r.noise_n, i, r.noise_seed, offset);
    } else if (blockIdx.x < nz + adam_blocks) {
        clip_adam_block(p, g, m, v, gr, b1, b2, eps, beta_pow, partial, nblk, clip_norm, 1.0f, norm_scale, step, nullptr, 0,
                        1.f, soft_t, soft_o4, soft_n4, tau, blockIdx.x - nz, adam_blocks);
    } else {
        const int lin = (int)(blockIdx.x - nz - adam_blocks);
        state_prologue_block(r.s, r.mean, r.std, r.B, r.S, r.chunk_len, r.C, r.clip, r.x, r.partial, r.adv, r.normalize_adv,
                             r.adv_stats, r.s2, r.x2, lin % r.gx, lin / r.gx);
    }
}

// ---- sharded exchange (data parallel): the owner of a 1/W slice of the bucket forms the MEAN of the per-replica CLIPPED
// gradients itself -- sum_w s_w g_w / W with ONE scalar s_w per replica (its clip factor) -- from the W unscaled slices an
// all-to-all delivered, applies TF-Adam to its slice of the parameters only, and turns the statistics behind the
// gradients into replica means (sync_model.py:64-95: per-worker clip_by_global_norm, accumulator mean, 14 x ApplyAdam,
// push_weights).  The bytes of a gradient can leave a replica before its norm is known; Adam's traffic is 1/W per replica.
__global__ __launch_bounds__(256) void k_shard_adam(float* __restrict__ px, float* __restrict__ m, float* __restrict__ v,
                                                    const float* __restrict__ recv, const float* __restrict__ scales,
                                                    int W, size_t lo4, size_t L4, size_t size4, size_t nb4, AdamGroups gr,
                                                    float b1, float b2, float eps, const float* __restrict__ beta_pow,
                                                    float* __restrict__ step) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && step) step[0] += 1.0f;
    const float omb1 = 1.f - b1, omb2 = 1.f - b2, invw = 1.0f / (float)W;
    float alpha[4];
    for (int k = 0; k < 4; ++k)
        alpha[k] = k < gr.n ? gr.lr[k] * sqrtf(1.f - beta_pow[4 * k + 3]) / (1.f - beta_pow[4 * k + 2]) : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < L4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t idx = lo4 + i;                          // float4 index in bucket space
        if (idx >= nb4) continue;                            // padding behind the bucket
        const float4* r = reinterpret_cast<const float4*>(recv) + i;
        if (idx < size4) {
            float4 g = r[0];
            const float s0 = scales[0];
            g.x *= s0; g.y *= s0; g.z *= s0; g.w *= s0;
            for (int w = 1; w < W; ++w) {                    // replica order: every owner sums the same way
                const float4 u = r[(size_t)w * L4];
                const float sw = scales[w];
                g.x += u.x * sw; g.y += u.y * sw; g.z += u.z * sw; g.w += u.w * sw;
            }
            g.x *= invw; g.y *= invw; g.z *= invw; g.w *= invw;
            int k = -1;
            for (int q = 0; q < gr.n; ++q)
                if (idx >= ((size_t)gr.off[q] >> 2) && idx < ((size_t)(gr.off[q] + gr.len[q]) >> 2)) k = q;
            if (k < 0) continue;
            float4 pp = reinterpret_cast<float4*>(px)[idx];
            float4 mm = reinterpret_cast<float4*>(m)[idx];
            float4 vv = reinterpret_cast<float4*>(v)[idx];
            const float a = alpha[k];
            adam1(pp.x, g.x, mm.x, vv.x, a, omb1, omb2, eps);
            adam1(pp.y, g.y, mm.y, vv.y, a, omb1, omb2, eps);
            adam1(pp.z, g.z, mm.z, vv.z, a, omb1, omb2, eps);
            adam1(pp.w, g.w, mm.w, vv.w, a, omb1, omb2, eps);
            reinterpret_cast<float4*>(px)[idx] = pp;
            reinterpret_cast<float4*>(m)[idx] = mm;
            reinterpret_cast<float4*>(v)[idx] = vv;
        } else {                                             // replicated statistics: plain mean over the replicas
            float4 t = r[0];
            for (int w = 1; w < W; ++w) {
                const float4 u = r[(size_t)w * L4];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            t.x *= invw; t.y *= invw; t.z *= invw; t.w *= invw;
            reinterpret_cast<float4*>(px)[idx] = t;
        }
    }
}


// ---- calibration: the build's own streaming copy (float4 grid-stride, U loads in flight per thread before the stores).
// The HBM roofline fractions of the mixture kernels are quoted against the 8 TB/s spec AND against what this access
// pattern moves on the box at hand (bench.py roofline_stress.device_copy / frac_of_box_copy).
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_stream_copy(float* __restrict__ dst, const float* __restrict__ src, size_t n4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const f4* s = reinterpret_cast<const f4*>(src);
    f4* d = reinterpret_cast<f4*>(dst);
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride);
            else d[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) d[i] = s[i];
}
// read-only variant (the forward kernels' pattern): the stream is folded into one float per thread so the loads cannot be
// dropped; sink [gridDim.x * 256]
template <bool NT>
__global__ __launch_bounds__(256) void k_stream_read(float* __restrict__ sink, const float* __restrict__ src, size_t n4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f4* s = reinterpret_cast<const f4*>(src);
    f4 a = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        a += NT ? __builtin_nontemporal_load(s + i) : s[i];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (a.x + a.y) + (a.z + a.w);
}

// beta powers advance after the applies (Adam/mul, Adam/mul_1 in the saved graph)
__global__ void k_beta_pow_step(float* __restrict__ beta_pow, float b1, float b2, float* __restrict__ step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        beta_pow[0] *= b1;
        beta_pow[1] *= b2;
        if (step) step[0] += 1.0f;          // float(global_step) used by the state-normaliser decay
    }
}

// Minibatch slicing of a training round (models/distributed_model.py flat_train :320-352): dst_c[b, :] = src_c[rows[b], :]
// for up to 8 experience columns in ONE launch (state, action, value, log-prob, advantage of a DPPO minibatch).
struct GatherCols {
    const float* src[8];
    float* dst[8];
    int width[8];
    int start[9];      // prefix sum of the widths: floats of one row over all columns
    int ncol;
};

__global__ void k_gather_columns(GatherCols c, const int64_t* __restrict__ rows, long n_src, int B) {
    const int per_row = c.start[c.ncol];
    const long total = (long)B * per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_row), j = (int)(i - (long)b * per_row);
        int col = 0;
        while (j >= c.start[col + 1]) ++col;
        const int k = j - c.start[col];
        long r = rows[b];
        r = r < 0 ? 0 : (r >= n_src ? n_src - 1 : r);          // never read outside the staged rollout
        c.dst[col][(long)b * c.width[col] + k] = c.src[col][r * c.width[col] + k];
    }
}


// SAC soft target update (sac.py:67-73): t <- (1-tau)*t + tau*s
__global__ __launch_bounds__(256) void k_soft_update(float* __restrict__ t, const float* __restrict__ s, size_t n,
                                                     float tau) {
    const float omt = 1.f - tau;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        t[i] = omt * t[i] + tau * s[i];
}

// SAC loss head (reference networks/actor_critic/sac.py:123-139, 160-173) on [B] vectors, one block:
//   vf' = min(q1', q2') - alpha*logp'          (target net, stop-grad)
//   q_target = r + gamma*not_terminal*vf'
//   value_loss = coef * mean((q_target-q1r)^2 + (q_target-q2r)^2)
//   policy_loss = mean(alpha*logp - min(q1p,q2p) - log_alpha*sg(logp + target_entropy))
// q_* are [2,B] group-major.  Outputs the gradients the backward pass needs.
__global__ __launch_bounds__(1024) void k_sac_loss(
    const float* __restrict__ q_pi, const float* __restrict__ q_run, const float* __restrict__ q_tgt,
    const float* __restrict__ logp, const float* __restrict__ logp_t, const float* __restrict__ reward,
    const float* __restrict__ not_terminal, const float* __restrict__ log_alpha, int B, float gamma,
    float value_coef, float target_entropy,
    float* __restrict__ dq_pi, float* __restrict__ dq_run, float* __restrict__ dlogp,
    float* __restrict__ dlog_alpha, float* __restrict__ loss3, int ld,      // ld: row stride of q_pi, q_run, dq_pi, dq_run
    unsigned long long* __restrict__ noise_epoch)   // the step's draw counter: += 1 (its readers sit in other launches)
{
    __shared__ float red[16];
    if (noise_epoch && threadIdx.x == 0) noise_epoch[0] += 1ull;
    const float la = log_alpha[0];
    const float alpha = expf(la);
    const float inv_B = 1.0f / (float)B;
    float ps = 0.f, vs = 0.f, es = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float vft = fminf(q_tgt[b], q_tgt[B + b]) - alpha * logp_t[b];
        const float qt = reward[b] + gamma * not_terminal[b] * vft;
        const float d1 = qt - q_run[b], d2 = qt - q_run[ld + b];
        vs += d1 * d1 + d2 * d2;
        dq_run[b] = -2.f * d1 * value_coef * inv_B;
        dq_run[ld + b] = -2.f * d2 * value_coef * inv_B;
        const float q1 = q_pi[b], q2 = q_pi[ld + b];
        const bool sel1 = q1 <= q2;                       // tf.minimum routes its gradient to x where x <= y
        const float lp = logp[b];
        ps += alpha * lp - fminf(q1, q2) - la * (lp + target_entropy);
        es += lp + target_entropy;
        dq_pi[b] = sel1 ? -inv_B : 0.f;
        dq_pi[ld + b] = sel1 ? 0.f : -inv_B;
        dlogp[b] = alpha * inv_B;
    }
    ps = block_sum(ps, red);
    vs = block_sum(vs, red);
    es = block_sum(es, red);
    if (threadIdx.x == 0) {
        const float pl = ps * inv_B, vl = vs * inv_B * value_coef;
        loss3[0] = pl + vl; loss3[1] = pl; loss3[2] = vl;
        dlog_alpha[0] = -es * inv_B;
    }
}


// BiasAdd + Relu6 that also leaves the Relu6Grad mask as BITS: for float4 index i (row-major over [rows, H/4]) and
// component k, bit (i & 63) of mask[(i >> 6) * 4 + k] = (0 < y + bias < 6).  The backward pass then reads 1 bit per
// element instead of re-reading the activations (33.5 MB -> 1 MB for the merged first layer).  A wave's 64 lanes own 64
// consecutive float4s, so the four words of a chunk are four wave ballots, stored by every lane to the same address
// (one transaction, no branch).
// (body: block `blk` of `nblk`; mask == nullptr: the plain BiasAdd + Relu6 -- k_bias_act's results, bit for bit)
__device__ __forceinline__ void bias_act_mask_body(float* __restrict__ y, const float* __restrict__ bias, int rows, int H,
                                                   int group_rows, unsigned long long* __restrict__ mask, int blk,
                                                   int nblk) {
    const int H4 = H >> 2;
    const size_t n4 = (size_t)rows * H4;
    const size_t stride = (size_t)nblk * blockDim.x;
    const size_t nround = (n4 + 63) & ~(size_t)63;             // whole waves: every lane of a live chunk takes part
    for (size_t i0 = (size_t)blk * blockDim.x + threadIdx.x; i0 < nround; i0 += 4 * stride) {
        float4 v[4], bb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = i0 + j * stride;
            const size_t ic = i < n4 ? i : 0;
            const int row = (int)(ic / H4);
            const int c4 = (int)(ic - (size_t)row * H4);
            const int g = group_rows > 0 ? row / group_rows : 0;
            v[j] = reinterpret_cast<float4*>(y)[ic];
            bb[j] = reinterpret_cast<const float4*>(bias + (size_t)g * H)[c4];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = i0 + j * stride;
            const bool live = i < n4;
            float4 t = v[j];
            t.x += bb[j].x; t.y += bb[j].y; t.z += bb[j].z; t.w += bb[j].w;
            const unsigned long long m0 = __ballot(live && t.x > 0.f && t.x < 6.f);
            const unsigned long long m1 = __ballot(live && t.y > 0.f && t.y < 6.f);
            const unsigned long long m2 = __ballot(live && t.z > 0.f && t.z < 6.f);
            const unsigned long long m3 = __ballot(live && t.w > 0.f && t.w < 6.f);
            t.x = fminf(fmaxf(t.x, 0.f), 6.f); t.y = fminf(fmaxf(t.y, 0.f), 6.f);
            t.z = fminf(fmaxf(t.z, 0.f), 6.f); t.w = fminf(fmaxf(t.w, 0.f), 6.f);
            if (live) reinterpret_cast<float4*>(y)[i] = t;
            if (mask && i < nround) {                           // wave-uniform: the whole chunk or nothing
                unsigned long long* mw = mask + (i >> 6) * 4;
                mw[0] = m0; mw[1] = m1; mw[2] = m2; mw[3] = m3;
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_bias_act_mask(float* __restrict__ y, const float* __restrict__ bias, int rows,
                                                       int H, int group_rows, unsigned long long* __restrict__ mask) {
    bias_act_mask_body(y, bias, rows, H, group_rows, mask, blockIdx.x, gridDim.x);
}
// Two epilogue passes of one width in ONE launch (SAC: the live critics' layer over 2B rows with mask bits and the target
// critics' layer over B rows without): blocks [0, nb_a) are pass a's grid, the rest pass b's.  A dependent launch costs
// ~5 us on this chip whatever it does; the two passes have nothing to do with each other.
__global__ __launch_bounds__(256) void k_bias_act_mask_pair(PfpnEpilogueJob a, PfpnEpilogueJob b, int H, int nb_a) {
    if ((int)blockIdx.x < nb_a)
        bias_act_mask_body(a.y, a.bias, a.rows, H, a.group_rows, (unsigned long long*)a.mask, blockIdx.x, nb_a);
    else
        bias_act_mask_body(b.y, b.bias, b.rows, H, b.group_rows, (unsigned long long*)b.mask, blockIdx.x - nb_a,
                           gridDim.x - nb_a);
}

// k_act_bwd_partial reading the mask bits of k_bias_act_mask instead of the activations (H/4 a multiple of 64, so a
// wave's 64 float4s are one mask chunk: its four words arrive as one wave-uniform 32-byte load).
__global__ __launch_bounds__(256) void k_act_bwd_partial_mask(float* __restrict__ dy,
                                                              const unsigned long long* __restrict__ mask, int grows,
                                                              int H, int chunk_len, float* __restrict__ partial) {
    __shared__ float4 red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + cl;
    const int H4 = H >> 2;
    const int ch = blockIdx.y;
    const size_t goff = (size_t)blockIdx.z * grows * H4;
    const int b0 = ch * chunk_len, b1 = min(grows, b0 + chunk_len);
    float4* d4 = reinterpret_cast<float4*>(dy) + goff;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long bit = 1ull << cl;
    for (int b = b0 + rl; b < b1; b += 16) {
        float4 g[4];
        unsigned long long m[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = b + 4 * j;
            const int rc = r < b1 ? r : b;
            g[j] = d4[(size_t)rc * H4 + c4];
            // chunk index of this wave's 64 float4s: wave-uniform
            const size_t chunk = (goff + (size_t)rc * H4 + (size_t)blockIdx.x * 64) >> 6;
            const unsigned long long* mw = mask + chunk * 4;
            m[j][0] = mw[0]; m[j][1] = mw[1]; m[j][2] = mw[2]; m[j][3] = mw[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = b + 4 * j;
            if (r < b1) {
                float4 t = g[j];
                t.x = (m[j][0] & bit) ? t.x : 0.f;
                t.y = (m[j][1] & bit) ? t.y : 0.f;
                t.z = (m[j][2] & bit) ? t.z : 0.f;
                t.w = (m[j][3] & bit) ? t.w : 0.f;
                d4[(size_t)r * H4 + c4] = t;
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
        }
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0) {
        float4 s = red[0][cl];
        for (int k = 1; k < 4; ++k) { const float4 t = red[k][cl]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
        reinterpret_cast<float4*>(partial + ((size_t)blockIdx.z * gridDim.y + ch) * H)[c4] = s;
    }
}

// Last trunk layer: y <- act(y + bias) AND the scalar critic head v[b] = y[b,:] . w + vb on the rows of group
// `vgroup` (critic fc3, actor_critic.py:206-210; vgroup < 0: every group has its own head, vw [G,H], vb [G], v [G*B] --
// the twin Q networks of SAC, sac.py:123-139) in the same pass -- the head's own launch re-read 8 MB to do
// 2 flops per element.  One wave per row, four rows in flight; the dot product is accumulated exactly like
// k_value_head_fwd (lane partial over i = lane, lane+64, ..; wave_sum), so both paths agree bit for bit.
#ifndef PFPN_BAV_ROWS
#define PFPN_BAV_ROWS 2
#endif
__device__ __forceinline__ void bias_act_value_body(float* __restrict__ y, const float* __restrict__ bias, int rows,
                                                    int H, int group_rows, int relu6, int vgroup,
                                                    const float* __restrict__ vw, const float* __restrict__ vb,
                                                    int vb_stride, float* __restrict__ v, int blk) {
    const int lane = threadIdx.x & 63;
    const int wave = (blk * blockDim.x + threadIdx.x) >> 6;
    constexpr int RW = PFPN_BAV_ROWS;                             // rows per wave (group_rows % 4 == 0 covers 2 and 4)
    const int r0 = wave * RW;
    if (r0 >= rows) return;
    const int H4 = H >> 2;
    const int g = group_rows > 0 ? r0 / group_rows : 0;          // group_rows % 4 == 0: the four rows share a group
    const bool all = vgroup < 0;                                  // every group carries its own head (twin Q networks)
    const bool head = all || g == vgroup;
    const float4* b4 = reinterpret_cast<const float4*>(bias + (size_t)g * H);
    const float4* w4 = reinterpret_cast<const float4*>(vw + (all ? (size_t)g * H : 0));
    float acc[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) acc[j] = 0.f;
    for (int i = lane; i < H4; i += 64) {
        float4 t[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int r = r0 + j < rows ? r0 + j : r0;
            t[j] = reinterpret_cast<const float4*>(y)[(size_t)r * H4 + i];
        }
        const float4 bb = b4[i];
        float4 ww = make_float4(0.f, 0.f, 0.f, 0.f);
        if (head) ww = w4[i];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            float4 u = t[j];
            u.x += bb.x; u.y += bb.y; u.z += bb.z; u.w += bb.w;
            if (relu6) {
                u.x = fminf(fmaxf(u.x, 0.f), 6.f); u.y = fminf(fmaxf(u.y, 0.f), 6.f);
                u.z = fminf(fmaxf(u.z, 0.f), 6.f); u.w = fminf(fmaxf(u.w, 0.f), 6.f);
            }
            if (r0 + j < rows) reinterpret_cast<float4*>(y)[(size_t)(r0 + j) * H4 + i] = u;
            acc[j] += u.x * ww.x + u.y * ww.y + u.z * ww.z + u.w * ww.w;
        }
    }
    if (!head) return;
    const float b0 = vb[all ? g * vb_stride : 0];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const float sacc = wave_sum(acc[j]);
        const int r = r0 + j;
        if (lane == 0 && r < rows) v[all ? r : r - vgroup * group_rows] = sacc + b0;
    }
}
__global__ __launch_bounds__(256) void k_bias_act_value(float* __restrict__ y, const float* __restrict__ bias, int rows,
                                                        int H, int group_rows, int relu6, int vgroup,
                                                        const float* __restrict__ vw, const float* __restrict__ vb,
                                                        int vb_stride, float* __restrict__ v) {
    bias_act_value_body(y, bias, rows, H, group_rows, relu6, vgroup, vw, vb, vb_stride, v, blockIdx.x);
}
// two last-layer passes with a head on every group (the live and the target twin critics of a SAC step) in ONE launch
__global__ __launch_bounds__(256) void k_bias_act_value_pair(PfpnEpilogueJob a, PfpnEpilogueJob b, int H, int nb_a) {
    if ((int)blockIdx.x < nb_a)
        bias_act_value_body(a.y, a.bias, a.rows, H, a.group_rows, 1, -1, a.value_w, a.value_b, a.value_b_stride, a.v,
                            blockIdx.x);
    else
        bias_act_value_body(b.y, b.bias, b.rows, H, b.group_rows, 1, -1, b.value_w, b.value_b, b.value_b_stride, b.v,
                            blockIdx.x - nb_a);
}

// k_act_bwd_partial with the critic head's backward folded in: on group `vgroup` the incoming gradient is not read
// but formed as dv[b] * w[c] (k_value_head_bwd_partial's arithmetic), masked and written like any other row, and a
// second accumulator collects the head's weight-gradient partials sum_b y[b,c] dv[b] into vpartial[chunk][H].
// SAC = the SAC loss head rides along (PfpnSacLoss, one pass of the twin critics over [(s, a_pi); (s, a_hist)]: grows = 2B,
// a head on both groups): the upstream dQ of a row is a function of that row alone (k_sac_loss's expressions, same order ->
// same bits), so every block forms the values it needs instead of reading them from a launch of their own; the blocks of
// the first column tile write them to dq_both (the head-bias job reads them), block (0, 0, 0) adds what a loss head
// leaves besides: the three losses, d log_alpha, dlogp [B] and the step's draw counter.
__device__ __forceinline__ float sac_dq(const PfpnSacLoss& L, int z, int b, float alpha, float inv_B) {
    if (b < L.B) {                                                // Q_z(s, a_pi): tf.minimum routes its gradient to x where x <= y
        const bool sel1 = L.q_both[b] <= L.q_both[2 * L.B + b];
        return (z == 0) == sel1 ? -inv_B : 0.f;
    }
    const int r = b - L.B;                                        // Q_z(s, a_hist): the value loss
    const float vft = fminf(L.q_tgt[r], L.q_tgt[L.B + r]) - alpha * L.logp_t[r];
    const float qt = L.reward[r] + L.gamma * L.not_terminal[r] * vft;
    const float d = qt - L.q_both[(size_t)z * 2 * L.B + L.B + r];
    return -2.f * d * L.value_coef * inv_B;
}

template <bool SAC>
__global__ __launch_bounds__(256) void k_act_bwd_partial_value(const float* __restrict__ y, float* __restrict__ dy,
                                                               int grows, int H, int chunk_len,
                                                               float* __restrict__ partial, int vgroup,
                                                               const float* __restrict__ vw,
                                                               const float* __restrict__ dv,
                                                               float* __restrict__ vpartial, PfpnSacLoss L) {
    __shared__ float4 red[4][64];
    float sac_alpha = 0.f, sac_inv_B = 0.f;
    if (SAC) { sac_alpha = expf(L.log_alpha[0]); sac_inv_B = 1.0f / (float)L.B; }
    const int nchunk = SAC ? gridDim.y - 1 : gridDim.y;           // SAC: one more row of blocks, of which ONE is the loss head
    if (SAC && (int)blockIdx.y == nchunk) {
        if (blockIdx.x || blockIdx.z) return;
        // what k_sac_loss leaves besides dQ, next to (not after) the blocks that do the pass (sums over 256 threads here,
        // 1024 there: the losses agree to rounding)
        float* redf = reinterpret_cast<float*>(red);
        if (L.noise_epoch && threadIdx.x == 0) L.noise_epoch[0] += 1ull;
        const float la = L.log_alpha[0];
        float ps = 0.f, vs = 0.f, es = 0.f;
        for (int b = threadIdx.x; b < L.B; b += blockDim.x) {
            const float vft = fminf(L.q_tgt[b], L.q_tgt[L.B + b]) - sac_alpha * L.logp_t[b];
            const float qt = L.reward[b] + L.gamma * L.not_terminal[b] * vft;
            const float d1 = qt - L.q_both[L.B + b], d2 = qt - L.q_both[3 * L.B + b];
            vs += d1 * d1 + d2 * d2;
            const float lp = L.logp[b];
            ps += sac_alpha * lp - fminf(L.q_both[b], L.q_both[2 * L.B + b]) - la * (lp + L.target_entropy);
            es += lp + L.target_entropy;
            L.dlogp[b] = sac_alpha * sac_inv_B;
        }
        ps = block_sum(ps, redf);
        vs = block_sum(vs, redf);
        es = block_sum(es, redf);
        if (threadIdx.x == 0) {
            const float pl = ps * sac_inv_B, vl = vs * sac_inv_B * L.value_coef;
            L.loss3[0] = pl + vl; L.loss3[1] = pl; L.loss3[2] = vl;
            L.dlog_alpha[0] = -es * sac_inv_B;
        }
        return;
    }
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + cl;
    const int H4 = H >> 2;
    const int ch = blockIdx.y;
    const bool all = vgroup < 0;                                  // every group has a head: vw [G,H], dv [G*grows],
    const bool head = all || (int)blockIdx.z == vgroup;           // vpartial [G][chunks][H]
    if (all) { vw += (size_t)blockIdx.z * H; dv += (size_t)blockIdx.z * grows; vpartial += (size_t)blockIdx.z * nchunk * H; }
    const size_t goff = (size_t)blockIdx.z * grows * H4;
    const int b0 = ch * chunk_len, b1 = min(grows, b0 + chunk_len);
    const float4* y4 = reinterpret_cast<const float4*>(y) + goff;
    float4* d4 = reinterpret_cast<float4*>(dy) + goff;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < H4) {
        float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (head) wv = reinterpret_cast<const float4*>(vw)[c4];
        for (int b = b0 + rl; b < b1; b += 16) {
            float4 g[4], v[4];
            float gs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = b + 4 * j;
                const int rc = r < b1 ? r : b;
                const size_t i = (size_t)rc * H4 + c4;
                v[j] = y4[i];
                if (head) {
                    gs[j] = SAC ? sac_dq(L, (int)blockIdx.z, rc, sac_alpha, sac_inv_B) : dv[rc];
                    if (SAC && blockIdx.x == 0 && cl == 0 && r < b1) L.dq_both[(size_t)blockIdx.z * grows + r] = gs[j];
                    g[j] = make_float4(gs[j] * wv.x, gs[j] * wv.y, gs[j] * wv.z, gs[j] * wv.w);
                } else {
                    gs[j] = 0.f;
                    g[j] = d4[i];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = b + 4 * j;
                if (r < b1) {
                    float4 t = g[j];
                    t.x = (v[j].x > 0.f && v[j].x < 6.f) ? t.x : 0.f;
                    t.y = (v[j].y > 0.f && v[j].y < 6.f) ? t.y : 0.f;
                    t.z = (v[j].z > 0.f && v[j].z < 6.f) ? t.z : 0.f;
                    t.w = (v[j].w > 0.f && v[j].w < 6.f) ? t.w : 0.f;
                    d4[(size_t)r * H4 + c4] = t;
                    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                    acc2.x += v[j].x * gs[j]; acc2.y += v[j].y * gs[j]; acc2.z += v[j].z * gs[j]; acc2.w += v[j].w * gs[j];
                }
            }
        }
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c4 < H4) {
        float4 t = red[0][cl];
        for (int k = 1; k < 4; ++k) { const float4 u = red[k][cl]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        reinterpret_cast<float4*>(partial + ((size_t)blockIdx.z * nchunk + ch) * H)[c4] = t;
    }
    if (!head) return;
    __syncthreads();
    red[rl][cl] = acc2;
    __syncthreads();
    if (rl == 0 && c4 < H4) {
        float4 t = red[0][cl];
        for (int k = 1; k < 4; ++k) { const float4 u = red[k][cl]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        reinterpret_cast<float4*>(vpartial + (size_t)ch * H)[c4] = t;
    }
}

// ---------------------------------------------------------------------------------------
// Deferred reductions: every slab producer of a learner step (fused mixture kernel, bias-gradient partials, the
// critic head, the state-normaliser moments) leaves its per-block partials in its own workspace, and ONE launch at
// the end of the backward pass reduces them all in fixed order.  Each of the five ~5 us latency-bound finalize
// launches this replaces sat on the step's critical path between two GEMMs.
// ---------------------------------------------------------------------------------------
struct ReduceJobTable {
    PfpnReduceJob job[PFPN_MAX_REDUCE_JOBS];
    int first_block[PFPN_MAX_REDUCE_JOBS + 1];
    int n_jobs;
};

// With `na.sq_partial` the launch also yields the global gradient norm's partial sums (tf.clip_by_global_norm,
// base_worker.py:97-102): every gradient that a job of this launch produces is squared where it is produced, every other
// gradient is final before the launch (the last GEMM precedes it) and is covered by PFPN_JOB_SQSUM jobs; block b
// writes sq_partial[b] (0 for jobs that produce no gradient).  Block 0 then also rotates the Adam powers, i.e. the
// launch replaces k_sqsum_partial_rot.
struct NormArgs {
    float* sq_partial;
    float* beta_pow;
    int n_groups;
    float b1, b2;
};

__global__ __launch_bounds__(1024) void k_reduce_jobs(ReduceJobTable t, NormArgs na) {
    __shared__ double dred[2][16][64];
    float (*red)[64] = reinterpret_cast<float (*)[64]>(&dred[0][0][0]);
    int j = 0;
    while (j + 1 < t.n_jobs && (int)blockIdx.x >= t.first_block[j + 1]) ++j;
    const PfpnReduceJob& q = t.job[j];
    const int blk = blockIdx.x - t.first_block[j];
    const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
    if (na.sq_partial && threadIdx.x == 0) {
        if (q.kind == PFPN_JOB_PPO_LOSS || q.kind == PFPN_JOB_NORM_EMA) na.sq_partial[blockIdx.x] = 0.f;
        if (blockIdx.x == 0 && na.beta_pow)
            for (int k = 0; k < na.n_groups; ++k) {
                float* bp = na.beta_pow + 4 * k;
                bp[2] = bp[0]; bp[3] = bp[1];
                bp[0] *= na.b1; bp[1] *= na.b2;
            }
    }
    if (q.kind == PFPN_JOB_COLSUM) {                      // out[i] = sum_k src[k*stride + i]       (k_bias_grad_final)
        const int i = blk * 64 + cl;
        float s = slab_reduce_1024<false>((const float*)q.src, q.nslab, (size_t)q.stride, i, i < q.n, red);
        if (q.in0)                                          // a second slab set: out = sum(first) + sum(second)
            s = s + slab_reduce_1024<false>(q.in0, (int)q.p0, (size_t)q.p1, i, i < q.n, red);
        if (kl == 0 && i < q.n) q.out[i] = s;
        if (na.sq_partial && kl == 0) {                    // wave 0 holds the block's 64 outputs
            const float sq = wave_sum((q.norm && i < q.n) ? s * s : 0.f);
            if (cl == 0) na.sq_partial[blockIdx.x] = sq;
        }
    } else if (q.kind == PFPN_JOB_VECSUM) {               // out[0] = sum_i src[i]                  (value-head bias)
        float s = 0.f;
        for (int i = threadIdx.x; i < q.n; i += blockDim.x) s += ((const float*)q.src)[i];
        s = block_sum(s, &red[0][0]);
        if (threadIdx.x == 0) {
            q.out[0] = s;
            if (na.sq_partial) na.sq_partial[blockIdx.x] = q.norm ? s * s : 0.f;
        }
    } else if (q.kind == PFPN_JOB_SQSUM) {                // sum of squares of a finished gradient segment, block `blk`'s share
        const float4* g4 = reinterpret_cast<const float4*>(q.src);
        const size_t n4 = (size_t)q.n >> 2;
        const size_t share = (n4 + q.nslab - 1) / q.nslab;
        const size_t lo = (size_t)blk * share, hi = lo + share < n4 ? lo + share : n4;
        float s = 0.f;
        size_t i = lo + threadIdx.x;
        for (; i + 3 * 1024 < hi; i += 4 * 1024) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = g4[i + u * 1024];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
        }
        for (; i < hi; i += 1024) { const float4 v = g4[i]; s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
        s = block_sum(s, &red[0][0]);
        if (threadIdx.x == 0 && na.sq_partial) na.sq_partial[blockIdx.x] = s;
    } else if (q.kind == PFPN_JOB_PPO_LOSS) {             // two loss columns -> loss3              (k_finalize)
        __shared__ float lsum[2];
        const float s = slab_reduce_1024<false>((const float*)q.src, q.nslab, (size_t)q.stride, cl & 1, cl < 2, red);
        if (kl == 0 && cl < 2) lsum[cl] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float pl = -lsum[0] * q.p0, vl = lsum[1] * q.p0 * q.p1;
            q.out[0] = pl + vl;
            q.out[1] = pl;
            q.out[2] = vl;
        }
    } else if (q.kind == PFPN_JOB_NORM_EMA) {             // chunk moments -> EMA of mean / std      (k_norm_ema)
        // 16 columns per block (first 256 threads = 16 columns x 16 chunk lanes; the rest only keep the barriers): with
        // 64 columns per block the whole 800 KB of moments went through FOUR CUs' load paths (~3 us); 13 blocks spread
        // it.  The arithmetic order per column (chunk lane kl sums k = kl, kl+16, ..; then lanes 0..15 in order) is
        // k_norm_ema's, so both give the same bits.
        const double* partial = (const double*)q.src;
        const int S = q.n, C = q.nslab;
        const int ecl = threadIdx.x & 15, ekl = (threadIdx.x >> 4) & 15;
        const int c = blk * 16 + ecl;
        const bool active = threadIdx.x < 256 && c < S;
        double a1 = 0.0, a2 = 0.0;
        if (active) {
            int k = ekl;
            for (; k + 112 < C; k += 128) {               // 16 loads in flight per thread
                double u[8], w[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    u[m] = partial[((size_t)(k + 16 * m) * 2 + 0) * S + c];
                    w[m] = partial[((size_t)(k + 16 * m) * 2 + 1) * S + c];
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) { a1 += u[m]; a2 += w[m]; }
            }
            for (; k < C; k += 16) {
                a1 += partial[((size_t)k * 2 + 0) * S + c];
                a2 += partial[((size_t)k * 2 + 1) * S + c];
            }
        }
        if (threadIdx.x < 256) { dred[0][ekl][ecl] = a1; dred[1][ekl][ecl] = a2; }
        __syncthreads();
        if (!active || ekl != 0) return;
        for (int m = 1; m < 16; ++m) { a1 += dred[0][m][ecl]; a2 += dred[1][m][ecl]; }
        const double Bd = (double)q.stride;               // batch rows
        const double mu = a1 / Bd;
        double var = a2 / Bd - mu * mu;
        if (var < 0.0) var = 0.0;
        const float bmean = (float)mu, bvar = (float)var;
        const float step = q.in0[0];
        const float d = fminf(0.9999f, (1.f + step) / (10.f + step));
        const float omd = 1.f - d;
        q.out[c] = d * q.in1[c] + omd * bmean;
        q.out2[c] = fmaxf(1e-6f, d * q.in2[c] + omd * sqrtf(bvar));
    }
}

// Row-chunk count of the column-reduction kernels: 16-row chunks (4 rows per wave in flight),
// at most PFPN_MAX_CHUNKS slabs in the workspace.
#define PFPN_MAX_CHUNKS 1024
static inline int row_chunks(int rows) {
    int C = (rows + 15) / 16;
    if (C > PFPN_MAX_CHUNKS) C = PFPN_MAX_CHUNKS;
    if (C < 1) C = 1;
    return C;
}
static inline int stream_grid(size_t n_items) {
    size_t g = (n_items + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// The three critic inputs of a SAC step -- concat([state, action]) for (s, a_pi), (s, a_hist), (s', a') (reference
// networks/actor_critic/sac.py:108-121, three tf.concat) -- in ONE launch: three 5-us launches of a latency-bound step.
__global__ __launch_bounds__(256) void k_sac_q_inputs(const float* __restrict__ x, const float* __restrict__ x_next,
                                                      const float* __restrict__ a_pi, const float* __restrict__ a_hist,
                                                      const float* __restrict__ a_next, int B, int S, int A,
                                                      float* __restrict__ xa_pi, float* __restrict__ xa_hist,
                                                      float* __restrict__ xa_next) {
    const int W = S + A;
    const long long per = (long long)B * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * per) return;
    const int which = (int)(i / per);
    const long long j = i - which * per;
    const int b = (int)(j / W), c = (int)(j - (long long)b * W);
    const float* xs = which == 2 ? x_next : x;
    const float* as = which == 0 ? a_pi : which == 1 ? a_hist : a_next;
    float* dst = which == 0 ? xa_pi : which == 1 ? xa_hist : xa_next;
    dst[j] = c < S ? xs[(size_t)b * S + c] : as[(size_t)b * A + (c - S)];
}

}  // namespace pfpn

using namespace pfpn;

extern "C" {

int pfpn_gae(const float* reward, const float* value, const int64_t* offsets, int E, float gamma,
             double gae_gamma, float* adv, void* stream) {
    PFPN_BEGIN();
    if (!reward || !value || !offsets || !adv || E < 0) return PFPN_ERR_ARG;
    if (E == 0) return PFPN_OK;
    hipLaunchKernelGGL(k_gae, dim3(E), dim3(256), 0, (hipStream_t)stream, reward, value, offsets, E, gamma,
                       gae_gamma, adv);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_ppo_loss(const float* logp, const float* logp_old, const float* adv, const float* value,
                  const float* value_old, const float* adv_stats, int B, float clip_eps, float value_coef,
                  float* dlogp, float* dvalue, float* loss3, void* stream) {
    PFPN_BEGIN();
    if (!logp || !logp_old || !adv || !value || !value_old || !adv_stats || !dlogp || !dvalue || !loss3 || B < 1)
        return PFPN_ERR_ARG;
    hipLaunchKernelGGL(k_ppo_loss, dim3(1), dim3(1024), 0, (hipStream_t)stream, logp, logp_old, adv, value,
                       value_old, adv_stats, B, 1.0f - clip_eps, 1.0f + clip_eps, value_coef, dlogp, dvalue, loss3);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_vtrace(const float* logp_target, const float* logp_behavior, const float* value, const float* reward,
                const float* not_terminal, const int32_t* seq_len, int E, int T, float gamma, float rho_clip,
                float pg_rho_clip, float* vs, float* pg_adv, float* mask, void* stream) {
    PFPN_BEGIN();
    if (!logp_target || !logp_behavior || !value || !reward || !not_terminal || !vs || !pg_adv || E < 0 || T < 1)
        return PFPN_ERR_ARG;
    if (E == 0) return PFPN_OK;
    hipLaunchKernelGGL(k_vtrace, dim3((E + 3) / 4), dim3(256), 0, (hipStream_t)stream, logp_target, logp_behavior,
                       value, reward, not_terminal, (const int*)seq_len, E, T, gamma, rho_clip, pg_rho_clip, vs,
                       pg_adv, mask);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_a2c_loss(const float* logp, const float* adv, const float* value, const float* value_target,
                  const float* adv_stats, const float* mask, int B, float value_coef, float* dlogp, float* dvalue,
                  float* loss4, void* stream) {
    PFPN_BEGIN();
    if (!logp || !adv || !value || !value_target || !adv_stats || !dlogp || !dvalue || !loss4 || B < 1)
        return PFPN_ERR_ARG;
    hipLaunchKernelGGL(k_a2c_loss, dim3(1), dim3(1024), 0, (hipStream_t)stream, logp, adv, value, value_target,
                       adv_stats, mask, B, value_coef, dlogp, dvalue, loss4);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_state_norm(const float* s, const float* mean, const float* std, int B, int S, float clip, float* x,
                    void* stream) {
    PFPN_BEGIN();
    if (!s || !mean || !std || !x || B < 0 || S < 1) return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    hipLaunchKernelGGL(k_state_norm, dim3(stream_grid((size_t)B * S)), dim3(256), 0, (hipStream_t)stream, s, mean,
                       std, B, S, clip, x);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

size_t pfpn_learner_workspace_bytes(int rows, int H) {
    // chunk slabs [PFPN_MAX_CHUNKS][H] + 2 column vectors + norm partials
    return ((size_t)(PFPN_MAX_CHUNKS + 2) * (size_t)(H > 0 ? H : 1) + 4096) * sizeof(float) + (size_t)rows * 0;
}

int pfpn_state_norm_update(const float* s, int B, int S, const float* step, const float* mean_in, const float* std_in,
                           float* mean_out, float* std_out, void* workspace, size_t workspace_bytes,
                           void* stream) {
    PFPN_BEGIN();
    if (!s || !step || !mean_in || !std_in || !mean_out || !std_out || B < 1 || S < 1) return PFPN_ERR_ARG;
    if (!workspace || workspace_bytes < pfpn_learner_workspace_bytes(B, S)) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int C = row_chunks(B);
    if (C > 256) C = 256;                      // 2*C*S doubles fit the learner workspace
    const int len = (B + C - 1) / C;
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(k_col_moments, dim3((S + 63) / 64, C), dim3(256), 0, st, s, B, S, len, partial);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_norm_ema, dim3((S + 63) / 64), dim3(1024), 0, st, (const double*)partial, C, B, S, step,
                       mean_in, std_in, mean_out, std_out);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_state_prologue(const float* s, const float* mean, const float* std, int B, int S, float clip, float* x,
                        int want_moments, const float* adv, int normalize_adv, float* adv_stats,
                        const float* s2, float* x2, void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!s || !mean || !std || !x || B < 1 || S < 1 || (adv_stats && !adv) || (s2 && !x2)) return PFPN_ERR_ARG;
    if (want_moments && (!workspace || workspace_bytes < pfpn_learner_workspace_bytes(B, S))) return PFPN_ERR_WORKSPACE;
    int C = row_chunks(B);
    if (C > 256) C = 256;
    const int len = (B + C - 1) / C;
    hipLaunchKernelGGL(k_state_prologue, dim3((S + 63) / 64, C + 1), dim3(256), 0, (hipStream_t)stream, s, mean, std, B,
                       S, len, C, clip, x, want_moments ? (double*)workspace : (double*)nullptr, adv, normalize_adv,
                       adv_stats, s2, x2);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_state_norm_ema(const void* workspace, int B, int S, const float* step, const float* mean_in,
                        const float* std_in, float* mean_out, float* std_out, void* stream) {
    PFPN_BEGIN();
    if (!workspace || !step || !mean_in || !std_in || !mean_out || !std_out || B < 1 || S < 1) return PFPN_ERR_ARG;
    int C = row_chunks(B);
    if (C > 256) C = 256;
    hipLaunchKernelGGL(k_norm_ema, dim3((S + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const double*)workspace, C,
                       B, S, step, mean_in, std_in, mean_out, std_out);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_bias_act(float* y, const float* bias, int rows, int H, int group_rows, int relu6, void* stream) {
    PFPN_BEGIN();
    if (!y || !bias || rows < 0 || H < 4 || (H & 3) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15))
        return PFPN_ERR_ARG;
    if (rows == 0) return PFPN_OK;
    hipLaunchKernelGGL(k_bias_act, dim3(stream_grid((size_t)rows * (H >> 2))), dim3(256), 0, (hipStream_t)stream,
                       y, bias, rows, H, group_rows, relu6);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_act_bwd_bias_grad(const float* y, float* dy, int rows, int H, int group_rows, int relu6, float* db,
                           void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!dy || !db || (relu6 && !y) || rows < 1 || H < 4 || (H & 3) || ((uintptr_t)dy & 15)) return PFPN_ERR_ARG;
    if (!workspace || workspace_bytes < pfpn_learner_workspace_bytes(rows, H)) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int G = group_rows > 0 ? rows / group_rows : 1;
    const int grows = group_rows > 0 ? group_rows : rows;
    if (group_rows > 0 && rows % group_rows) return PFPN_ERR_ARG;
    int cpg = row_chunks(grows);                 // chunks per group; chunks never straddle groups
    if (cpg * G > PFPN_MAX_CHUNKS) cpg = PFPN_MAX_CHUNKS / G;
    if (cpg < 1) return PFPN_ERR_ARG;
    const int len = (grows + cpg - 1) / cpg;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(k_act_bwd_partial, dim3(((H >> 2) + 63) / 64, cpg, G), dim3(256), 0, st, y, dy, grows, H, len,
                       relu6, partial);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bias_grad_final, dim3((H + 63) / 64, G), dim3(1024), 0, st, partial, cpg, G, H, db);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_state_moment_chunks(int B) {
    int C = row_chunks(B);
    return C > 256 ? 256 : C;
}

size_t pfpn_relu6_mask_bytes(int rows, int H) {
    const size_t n4 = (size_t)rows * (size_t)(H >> 2);
    return ((n4 + 63) >> 6) * 4 * sizeof(unsigned long long);
}

int pfpn_bias_act_mask(float* y, const float* bias, int rows, int H, int group_rows, void* mask, size_t mask_bytes,
                       void* stream) {
    PFPN_BEGIN();
    if (!y || !bias || !mask || rows < 1 || H < 4 || (H & 3) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) ||
        ((uintptr_t)mask & 7) || (group_rows > 0 && rows % group_rows))
        return PFPN_ERR_ARG;
    if (mask_bytes < pfpn_relu6_mask_bytes(rows, H)) return PFPN_ERR_WORKSPACE;
    hipLaunchKernelGGL(k_bias_act_mask, dim3(stream_grid((size_t)rows * (H >> 2))), dim3(256), 0, (hipStream_t)stream, y,
                       bias, rows, H, group_rows, (unsigned long long*)mask);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_act_bwd_partial_mask(float* dy, const void* mask, int rows, int H, int group_rows, float* partial,
                              size_t partial_floats, int* chunks_per_group, void* stream) {
    PFPN_BEGIN();
    if (!dy || !mask || !partial || !chunks_per_group || rows < 1 || H < 256 || (H & 255) || ((uintptr_t)dy & 15) ||
        ((uintptr_t)mask & 7))
        return PFPN_ERR_ARG;                         // H/4 must be a multiple of 64: a wave's float4s are one mask chunk
    const int G = group_rows > 0 ? rows / group_rows : 1;
    const int grows = group_rows > 0 ? group_rows : rows;
    if (group_rows > 0 && rows % group_rows) return PFPN_ERR_ARG;
    int cpg = row_chunks(grows);
    if (cpg * G > PFPN_MAX_CHUNKS) cpg = PFPN_MAX_CHUNKS / G;
    if (cpg < 1) return PFPN_ERR_ARG;
    if (partial_floats < (size_t)G * cpg * H) return PFPN_ERR_WORKSPACE;
    const int len = (grows + cpg - 1) / cpg;
    *chunks_per_group = cpg;
    hipLaunchKernelGGL(k_act_bwd_partial_mask, dim3((H >> 2) / 64, cpg, G), dim3(256), 0, (hipStream_t)stream, dy,
                       (const unsigned long long*)mask, grows, H, len, partial);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_bias_act_value(float* y, const float* bias, int rows, int H, int group_rows, int relu6, int value_group,
                        const float* value_w, const float* value_b, int value_b_stride, float* v, void* stream) {
    PFPN_BEGIN();
    if (!y || !bias || !value_w || !value_b || !v || rows < 1 || H < 4 || (H & 3) || ((uintptr_t)y & 15) ||
        ((uintptr_t)bias & 15) || ((uintptr_t)value_w & 15) || group_rows < 4 || (group_rows & 3) || rows % group_rows ||
        value_group >= rows / group_rows || value_b_stride < 1)
        return PFPN_ERR_ARG;
    const int waves = (rows + PFPN_BAV_ROWS - 1) / PFPN_BAV_ROWS;
    hipLaunchKernelGGL(k_bias_act_value, dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream, y, bias, rows, H,
                       group_rows, relu6, value_group, value_w, value_b, value_b_stride, v);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

static bool epilogue_job_ok(const PfpnEpilogueJob* j, int H, bool value) {
    if (!j || !j->y || !j->bias || j->rows < 1 || ((uintptr_t)j->y & 15) || ((uintptr_t)j->bias & 15) ||
        (j->group_rows > 0 && j->rows % j->group_rows))
        return false;
    if (j->mask && (((uintptr_t)j->mask & 7) || j->mask_bytes < pfpn_relu6_mask_bytes(j->rows, H))) return false;
    if (value && (!j->value_w || !j->value_b || !j->v || ((uintptr_t)j->value_w & 15) || j->group_rows < 4 ||
                  (j->group_rows & 3) || j->value_b_stride < 1))
        return false;
    return true;
}

int pfpn_bias_act_mask_pair(const PfpnEpilogueJob* a, const PfpnEpilogueJob* b, int H, void* stream) {
    PFPN_BEGIN();
    if (H < 4 || (H & 3) || !epilogue_job_ok(a, H, false) || !epilogue_job_ok(b, H, false)) return PFPN_ERR_ARG;
    const int na = stream_grid((size_t)a->rows * (H >> 2)), nb = stream_grid((size_t)b->rows * (H >> 2));
    hipLaunchKernelGGL(k_bias_act_mask_pair, dim3(na + nb), dim3(256), 0, (hipStream_t)stream, *a, *b, H, na);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_bias_act_value_pair(const PfpnEpilogueJob* a, const PfpnEpilogueJob* b, int H, void* stream) {
    PFPN_BEGIN();
    if (H < 4 || (H & 3) || !epilogue_job_ok(a, H, true) || !epilogue_job_ok(b, H, true)) return PFPN_ERR_ARG;
    const int na = (((a->rows + PFPN_BAV_ROWS - 1) / PFPN_BAV_ROWS) + 3) / 4;
    const int nb = (((b->rows + PFPN_BAV_ROWS - 1) / PFPN_BAV_ROWS) + 3) / 4;
    hipLaunchKernelGGL(k_bias_act_value_pair, dim3(na + nb), dim3(256), 0, (hipStream_t)stream, *a, *b, H, na);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_act_bwd_partial(const float* y, float* dy, int rows, int H, int group_rows, int relu6, int value_group,
                         const float* value_w, const float* dvalue, float* value_partial, float* partial,
                         size_t partial_floats, int* chunks_per_group, void* stream) {
    PFPN_BEGIN();
    if (!dy || !partial || !chunks_per_group || (relu6 && !y) || rows < 1 || H < 4 || (H & 3) || ((uintptr_t)dy & 15))
        return PFPN_ERR_ARG;
    const int G = group_rows > 0 ? rows / group_rows : 1;
    const int grows = group_rows > 0 ? group_rows : rows;
    if (group_rows > 0 && rows % group_rows) return PFPN_ERR_ARG;
    int cpg = row_chunks(grows);
    if (cpg * G > PFPN_MAX_CHUNKS) cpg = PFPN_MAX_CHUNKS / G;
    if (cpg < 1) return PFPN_ERR_ARG;
    if (partial_floats < (size_t)G * cpg * H) return PFPN_ERR_WORKSPACE;
    const int len = (grows + cpg - 1) / cpg;
    *chunks_per_group = cpg;
    hipStream_t st = (hipStream_t)stream;
    if (value_w) {
        if (!dvalue || !value_partial || !relu6 || value_group >= G || ((uintptr_t)value_w & 15))
            return PFPN_ERR_ARG;
        hipLaunchKernelGGL(k_act_bwd_partial_value<false>, dim3(((H >> 2) + 63) / 64, cpg, G), dim3(256), 0, st, y, dy,
                           grows, H, len, partial, value_group, value_w, dvalue, value_partial, PfpnSacLoss{});
    } else {
        hipLaunchKernelGGL(k_act_bwd_partial, dim3(((H >> 2) + 63) / 64, cpg, G), dim3(256), 0, st, y, dy, grows, H, len,
                           relu6, partial);
    }
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_act_bwd_partial_sac(const float* y, float* dy, int H, const float* value_w, float* value_partial, float* partial,
                             size_t partial_floats, int* chunks_per_group, const PfpnSacLoss* loss, void* stream) {
    PFPN_BEGIN();
    if (!y || !dy || !value_w || !value_partial || !partial || !chunks_per_group || !loss || H < 4 || (H & 3) ||
        ((uintptr_t)dy & 15) || ((uintptr_t)value_w & 15))
        return PFPN_ERR_ARG;
    const PfpnSacLoss& L = *loss;
    if (!L.q_both || !L.q_tgt || !L.logp || !L.logp_t || !L.reward || !L.not_terminal || !L.log_alpha || !L.dq_both ||
        !L.dlogp || !L.dlog_alpha || !L.loss3 || L.B < 1)
        return PFPN_ERR_ARG;
    const int G = 2, grows = 2 * L.B;
    int cpg = row_chunks(grows);
    if (cpg * G > PFPN_MAX_CHUNKS) cpg = PFPN_MAX_CHUNKS / G;
    if (cpg < 1) return PFPN_ERR_ARG;
    if (partial_floats < (size_t)G * cpg * H) return PFPN_ERR_WORKSPACE;
    const int len = (grows + cpg - 1) / cpg;
    *chunks_per_group = cpg;
    hipLaunchKernelGGL(k_act_bwd_partial_value<true>, dim3(((H >> 2) + 63) / 64, cpg + 1, G), dim3(256), 0, (hipStream_t)stream,
                       y, dy, grows, H, len, partial, -1, value_w, (const float*)nullptr, value_partial, L);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

static int reduce_jobs_launch(const PfpnReduceJob* jobs, int n_jobs, NormArgs na, size_t sq_capacity, int* n_blocks,
                              void* stream) {
    if (!jobs || n_jobs < 1 || n_jobs > PFPN_MAX_REDUCE_JOBS) return PFPN_ERR_ARG;
    ReduceJobTable t{};
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const PfpnReduceJob& q = jobs[j];
        if (!q.src || (!q.out && q.kind != PFPN_JOB_SQSUM) || q.n < 1 || q.nslab < 1) return PFPN_ERR_ARG;
        int nb = 1;
        if (q.kind == PFPN_JOB_COLSUM) {
            if (q.in0 && (q.p0 < 1.f || q.p1 < (float)q.n)) return PFPN_ERR_ARG;
            nb = (q.n + 63) / 64;
        } else if (q.kind == PFPN_JOB_NORM_EMA) {
            if (!q.out2 || !q.in0 || !q.in1 || !q.in2 || q.stride < 1) return PFPN_ERR_ARG;
            nb = (q.n + 15) / 16;
        } else if (q.kind == PFPN_JOB_SQSUM) {
            if (!na.sq_partial || (q.n & 3) || ((uintptr_t)q.src & 15)) return PFPN_ERR_ARG;
            nb = q.nslab;
        } else if (q.kind != PFPN_JOB_VECSUM && q.kind != PFPN_JOB_PPO_LOSS) return PFPN_ERR_ARG;
        t.job[j] = q;
        t.first_block[j] = blocks;
        blocks += nb;
    }
    t.first_block[n_jobs] = blocks;
    t.n_jobs = n_jobs;
    if (na.sq_partial && (size_t)blocks > sq_capacity) return PFPN_ERR_WORKSPACE;
    if (n_blocks) *n_blocks = blocks;
    hipLaunchKernelGGL(k_reduce_jobs, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, t, na);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_reduce_jobs(const PfpnReduceJob* jobs, int n_jobs, void* stream) {
    PFPN_BEGIN();
    return reduce_jobs_launch(jobs, n_jobs, NormArgs{nullptr, nullptr, 0, 0.f, 0.f}, 0, nullptr, stream);
}

int pfpn_reduce_jobs_norm(const PfpnReduceJob* jobs, int n_jobs, float* sq_partial, size_t sq_capacity, int* n_partials,
                          float* beta_pow, int n_groups, float beta1, float beta2, void* stream) {
    PFPN_BEGIN();
    if (!sq_partial || !n_partials || (beta_pow && n_groups < 1)) return PFPN_ERR_ARG;
    return reduce_jobs_launch(jobs, n_jobs, NormArgs{sq_partial, beta_pow, n_groups, beta1, beta2}, sq_capacity, n_partials,
                              stream);
}

int pfpn_value_head_fwd(const float* h, const float* w, const float* bias, int B, int H, float* v, void* stream) {
    PFPN_BEGIN();
    if (!h || !w || !bias || !v || B < 0 || H < 4 || (H & 3)) return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    hipLaunchKernelGGL(k_value_head_fwd, dim3((B * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, h, w, bias,
                       B, H, v);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_value_head_bwd(const float* h, const float* w, const float* dv, int B, int H, int relu6, float* dh,
                        float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!h || !w || !dv || !dh || !dw || !db || B < 1 || H < 4 || (H & 3)) return PFPN_ERR_ARG;
    if (!workspace || workspace_bytes < pfpn_learner_workspace_bytes(B, H)) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int C = row_chunks(B);
    const int len = (B + C - 1) / C;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(k_value_head_bwd_partial, dim3(((H >> 2) + 63) / 64, C), dim3(256), 0, st, h, w, dv, B, H, len,
                       relu6, dh, partial);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bias_grad_final, dim3((H + 63) / 64 + 1, 1), dim3(1024), 0, st, partial, C, 1, H, dw, dv, B,
                       db);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_clip_by_global_norm(float* grads, size_t n, float clip_norm, int apply, float* norm_scale,
                             void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!grads || !norm_scale || n < 1 || ((uintptr_t)grads & 15)) return PFPN_ERR_ARG;
    if (!workspace || workspace_bytes < 1024 * sizeof(float)) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int nblk = stream_grid(n >> 2);
    if (nblk > 1024) nblk = 1024;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(k_sqsum_partial, dim3(nblk), dim3(256), 0, st, grads, n, partial);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_clip_scale, dim3(1), dim3(256), 0, st, partial, nblk, clip_norm, norm_scale);
    PFPN_CHECK_LAUNCH();
    if (apply) {
        hipLaunchKernelGGL(k_scale_inplace, dim3(stream_grid(n >> 2)), dim3(256), 0, st, grads, n, norm_scale);
        PFPN_CHECK_LAUNCH();
    }
    return PFPN_OK;
}

int pfpn_tf_adam(float* params, const float* grads, float* m, float* v, size_t n, float lr, float beta1,
                 float beta2, float eps, float* beta_pow, int advance_beta_pow, const float* norm_scale,
                 float grad_mul, float* step_counter, void* stream) {
    PFPN_BEGIN();
    if (!params || !grads || !m || !v || !beta_pow || n < 1) return PFPN_ERR_ARG;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) return PFPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_tf_adam, dim3(stream_grid(n >> 2)), dim3(256), 0, st, params, grads, m, v, n, lr, beta1,
                       beta2, eps, beta_pow, norm_scale, grad_mul);
    PFPN_CHECK_LAUNCH();
    if (advance_beta_pow) {
        hipLaunchKernelGGL(k_beta_pow_step, dim3(1), dim3(64), 0, st, beta_pow, beta1, beta2, step_counter);
        PFPN_CHECK_LAUNCH();
    }
    return PFPN_OK;
}

int pfpn_clip_adam(float* params, const float* grads, float* m, float* v, size_t n, int n_groups,
                   const long long* group_off, const long long* group_len, const float* group_lr, float beta1,
                   float beta2, float eps, float* beta_pow, float clip_norm, float* norm_scale, float* step_counter,
                   void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!params || !grads || !m || !v || !beta_pow || !norm_scale || !group_off || !group_len || !group_lr ||
        n < 1 || n_groups < 1 || n_groups > 4)
        return PFPN_ERR_ARG;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) return PFPN_ERR_ARG;
    if (!workspace || workspace_bytes < 256 * sizeof(float)) return PFPN_ERR_WORKSPACE;
    AdamGroups gr{};
    gr.n = n_groups;
    size_t longest = 0;
    for (int k = 0; k < n_groups; ++k) {
        if (group_off[k] < 0 || group_len[k] < 0 || (group_off[k] & 3) || (group_len[k] & 3) ||
            (size_t)(group_off[k] + group_len[k]) > n)
            return PFPN_ERR_ARG;
        gr.off[k] = group_off[k]; gr.len[k] = group_len[k]; gr.lr[k] = group_lr[k];
        if ((size_t)group_len[k] > longest) longest = (size_t)group_len[k];
    }
    hipStream_t st = (hipStream_t)stream;
    int nblk = stream_grid(n >> 2);
    if (nblk > 256) nblk = 256;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(k_sqsum_partial_rot, dim3(nblk), dim3(256), 0, st, grads, n, partial, beta_pow, n_groups, beta1,
                       beta2);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_clip_adam, dim3(stream_grid(longest >> 2)), dim3(256), 0, st, params, grads, m, v, gr, beta1,
                       beta2, eps, (const float*)beta_pow, (const float*)partial, nblk, clip_norm, 1.0f, norm_scale,
                       step_counter);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

// ---- data-parallel optimiser step around the all-reduce: clip per replica (2 launches), Adam on the sum (1) ----
int pfpn_clip_adam_prenorm(float* params, const float* grads, float* m, float* v, size_t n, int n_groups,
                           const long long* group_off, const long long* group_len, const float* group_lr, float beta1,
                           float beta2, float eps, const float* beta_pow, float clip_norm, float* norm_scale,
                           float* step_counter, const float* sq_partial, int n_partials, float* soft_target,
                           long long soft_off, long long soft_len, float soft_tau, void* stream) {
    PFPN_BEGIN();
    if (!params || !grads || !m || !v || !beta_pow || !group_off || !group_len || !group_lr || !sq_partial || n < 1 ||
        n_groups < 1 || n_groups > 4 || n_partials < 1)
        return PFPN_ERR_ARG;
    if (soft_target && (soft_off < 0 || soft_len < 0 || (soft_off & 3) || (soft_len & 3) ||
                        (size_t)(soft_off + soft_len) > n || ((uintptr_t)soft_target & 15)))
        return PFPN_ERR_ARG;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) return PFPN_ERR_ARG;
    AdamGroups gr{};
    gr.n = n_groups;
    size_t longest = 0;
    for (int k = 0; k < n_groups; ++k) {
        if (group_off[k] < 0 || group_len[k] < 0 || (group_off[k] & 3) || (group_len[k] & 3) ||
            (size_t)(group_off[k] + group_len[k]) > n)
            return PFPN_ERR_ARG;
        gr.off[k] = group_off[k]; gr.len[k] = group_len[k]; gr.lr[k] = group_lr[k];
        if ((size_t)group_len[k] > longest) longest = (size_t)group_len[k];
    }
    hipLaunchKernelGGL(k_clip_adam, dim3(stream_grid(longest >> 2)), dim3(256), 0, (hipStream_t)stream, params, grads, m,
                       v, gr, beta1, beta2, eps, beta_pow, sq_partial, n_partials, clip_norm, 1.0f, norm_scale,
                       step_counter, (float*)nullptr, (size_t)0, 1.f, soft_target, (size_t)soft_off >> 2,
                       (size_t)soft_len >> 2, soft_tau);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_clip_adam_prenorm_prologue(float* params, const float* grads, float* m, float* v, size_t n, int n_groups,
                                    const long long* group_off, const long long* group_len, const float* group_lr,
                                    float beta1, float beta2, float eps, const float* beta_pow, float clip_norm,
                                    float* norm_scale, float* step_counter, const float* sq_partial, int n_partials,
                                    const PfpnStatePrologue* next, float* soft_target, long long soft_off,
                                    long long soft_len, float soft_tau, void* stream) {
    PFPN_BEGIN();
    if (!params || !grads || !m || !v || !beta_pow || !group_off || !group_len || !group_lr || !sq_partial || n < 1 ||
        n_groups < 1 || n_groups > 4 || n_partials < 1 || !next)
        return PFPN_ERR_ARG;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) return PFPN_ERR_ARG;
    if (soft_target && (soft_off < 0 || soft_len < 0 || (soft_off & 3) || (soft_len & 3) ||
                        (size_t)(soft_off + soft_len) > n || ((uintptr_t)soft_target & 15)))
        return PFPN_ERR_ARG;
    const bool has_pro = next->s != nullptr, has_noise = next->noise != nullptr;
    if (!has_pro && !has_noise) return PFPN_ERR_ARG;
    if (has_pro && (!next->mean || !next->std || !next->x || next->B < 1 || next->S < 1 || (next->adv_stats && !next->adv) ||
                    ((next->s2 != nullptr) != (next->x2 != nullptr))))
        return PFPN_ERR_ARG;
    if (has_pro && (!next->workspace || next->workspace_bytes < pfpn_learner_workspace_bytes(next->B, next->S)))
        return PFPN_ERR_WORKSPACE;
    if (has_noise && (!next->noise_epoch || next->noise_n < 1)) return PFPN_ERR_ARG;
    AdamGroups gr{};
    gr.n = n_groups;
    size_t longest = 0;
    for (int k = 0; k < n_groups; ++k) {
        if (group_off[k] < 0 || group_len[k] < 0 || (group_off[k] & 3) || (group_len[k] & 3) ||
            (size_t)(group_off[k] + group_len[k]) > n)
            return PFPN_ERR_ARG;
        gr.off[k] = group_off[k]; gr.len[k] = group_len[k]; gr.lr[k] = group_lr[k];
        if ((size_t)group_len[k] > longest) longest = (size_t)group_len[k];
    }
    PrologueRider r{};
    if (has_pro) {
        r.s = next->s; r.mean = next->mean; r.std = next->std; r.B = next->B; r.S = next->S; r.clip = next->clip;
        r.x = next->x; r.partial = (double*)next->workspace; r.adv = next->adv; r.normalize_adv = next->normalize_adv;
        r.adv_stats = next->adv_stats; r.s2 = next->s2; r.x2 = next->x2;
        int C = row_chunks(next->B);
        if (C > 256) C = 256;
        r.C = C; r.chunk_len = (next->B + C - 1) / C; r.gx = (next->S + 63) / 64;
        r.pro_blocks = (unsigned)(r.gx * (C + 1));
    }
    if (has_noise) {
        r.noise = next->noise; r.noise_n = next->noise_n; r.noise_seed = next->noise_seed;
        r.noise_epoch = (const unsigned long long*)next->noise_epoch;
        size_t nb = (next->noise_n + 255) / 256;
        r.noise_blocks = (unsigned)(nb > 1024 ? 1024 : nb);
    }
    const unsigned adam_blocks = (unsigned)stream_grid(longest >> 2);
    hipLaunchKernelGGL(k_clip_adam_prologue, dim3(adam_blocks + r.pro_blocks + r.noise_blocks), dim3(256), 0,
                       (hipStream_t)stream, params, grads, m, v, gr, beta1, beta2, eps, beta_pow, sq_partial, n_partials,
                       clip_norm, norm_scale, step_counter, adam_blocks, r, soft_target, (size_t)soft_off >> 2,
                       (size_t)soft_len >> 2, soft_tau);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_clip_scale_inplace(float* grads, size_t n, float clip_norm, float* norm_scale, float* beta_pow, int n_groups,
                            float beta1, float beta2, void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!grads || !norm_scale || n < 1 || ((uintptr_t)grads & 15) || (beta_pow && (n_groups < 1 || n_groups > 4)))
        return PFPN_ERR_ARG;
    if (!workspace || workspace_bytes < 256 * sizeof(float)) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int nblk = stream_grid(n >> 2);
    if (nblk > 256) nblk = 256;
    float* partial = (float*)workspace;
    if (beta_pow)
        hipLaunchKernelGGL(k_sqsum_partial_rot, dim3(nblk), dim3(256), 0, st, (const float*)grads, n, partial, beta_pow,
                           n_groups, beta1, beta2);
    else
        hipLaunchKernelGGL(k_sqsum_partial, dim3(nblk), dim3(256), 0, st, (const float*)grads, n, partial);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_clip_scale_inplace, dim3(stream_grid(n >> 2)), dim3(256), 0, st, grads, n,
                       (const float*)partial, nblk, clip_norm, norm_scale);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_clip_scale_inplace_prenorm(float* grads, size_t n, float clip_norm, float* norm_scale, const float* sq_partial,
                                    int n_partials, void* stream) {
    PFPN_BEGIN();
    if (!grads || !norm_scale || !sq_partial || n < 1 || n_partials < 1 || ((uintptr_t)grads & 15)) return PFPN_ERR_ARG;
    hipLaunchKernelGGL(k_clip_scale_inplace, dim3(stream_grid(n >> 2)), dim3(256), 0, (hipStream_t)stream, grads, n,
                       sq_partial, n_partials, clip_norm, norm_scale);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_stream_copy(float* dst, const float* src, size_t n, int unroll, int nontemporal, int blocks, void* stream) {
    PFPN_BEGIN();
    if (!dst || !src || n < 4 || (n & 3) || (((uintptr_t)dst | (uintptr_t)src) & 15) || blocks < 1 || blocks > 65536)
        return PFPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t n4 = n >> 2;
    if (unroll == 0) {                 // read-only stream: dst is the [blocks * 256] sink
        if (nontemporal) hipLaunchKernelGGL((k_stream_read<true>), dim3(blocks), dim3(256), 0, st, dst, src, n4);
        else hipLaunchKernelGGL((k_stream_read<false>), dim3(blocks), dim3(256), 0, st, dst, src, n4);
        PFPN_CHECK_LAUNCH();
        return PFPN_OK;
    }
#define PFPN_COPY_CASE(U_, NT_)                                                                              \
    hipLaunchKernelGGL((k_stream_copy<U_, NT_>), dim3(blocks), dim3(256), 0, st, dst, src, n4)
    if (unroll == 1) { if (nontemporal) PFPN_COPY_CASE(1, true); else PFPN_COPY_CASE(1, false); }
    else if (unroll == 2) { if (nontemporal) PFPN_COPY_CASE(2, true); else PFPN_COPY_CASE(2, false); }
    else if (unroll == 4) { if (nontemporal) PFPN_COPY_CASE(4, true); else PFPN_COPY_CASE(4, false); }
    else if (unroll == 8) { if (nontemporal) PFPN_COPY_CASE(8, true); else PFPN_COPY_CASE(8, false); }
    else return PFPN_ERR_ARG;
#undef PFPN_COPY_CASE
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_clip_scale_only(const float* grads, size_t n, float clip_norm, float* norm_scale, const float* sq_partial,
                         int n_partials, float* beta_pow, int n_groups, float beta1, float beta2, void* workspace,
                         size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!norm_scale || (beta_pow && (n_groups < 1 || n_groups > 4))) return PFPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (sq_partial) {                    // partial sums (and rotated powers) left by pfpn_reduce_jobs_norm
        if (n_partials < 1) return PFPN_ERR_ARG;
        hipLaunchKernelGGL(k_clip_scale, dim3(1), dim3(256), 0, st, sq_partial, n_partials, clip_norm, norm_scale);
        PFPN_CHECK_LAUNCH();
        return PFPN_OK;
    }
    if (!grads || n < 1 || ((uintptr_t)grads & 15)) return PFPN_ERR_ARG;
    if (!workspace || workspace_bytes < 256 * sizeof(float)) return PFPN_ERR_WORKSPACE;
    int nblk = stream_grid(n >> 2);
    if (nblk > 256) nblk = 256;
    float* partial = (float*)workspace;
    if (beta_pow)
        hipLaunchKernelGGL(k_sqsum_partial_rot, dim3(nblk), dim3(256), 0, st, grads, n, partial, beta_pow, n_groups, beta1,
                           beta2);
    else
        hipLaunchKernelGGL(k_sqsum_partial, dim3(nblk), dim3(256), 0, st, grads, n, partial);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_clip_scale, dim3(1), dim3(256), 0, st, (const float*)partial, nblk, clip_norm, norm_scale);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_shard_adam(float* params_x, float* m, float* v, const float* recv, const float* scales, int world,
                    long long shard_lo, long long shard_len, long long n_params, long long n_bucket, int n_groups,
                    const long long* group_off, const long long* group_len, const float* group_lr, float beta1,
                    float beta2, float eps, const float* beta_pow, float* step_counter, void* stream) {
    PFPN_BEGIN();
    if (!params_x || !m || !v || !recv || !scales || !beta_pow || !group_off || !group_len || !group_lr || world < 1 ||
        n_groups < 1 || n_groups > 4 || shard_lo < 0 || shard_len < 4 || n_params < 4 || n_bucket < n_params)
        return PFPN_ERR_ARG;
    if ((shard_lo | shard_len | n_params | n_bucket) & 3) return PFPN_ERR_ARG;
    if (((uintptr_t)params_x | (uintptr_t)m | (uintptr_t)v | (uintptr_t)recv) & 15) return PFPN_ERR_ARG;
    AdamGroups gr{};
    gr.n = n_groups;
    for (int k = 0; k < n_groups; ++k) {
        if (group_off[k] < 0 || group_len[k] < 0 || (group_off[k] & 3) || (group_len[k] & 3) ||
            group_off[k] + group_len[k] > n_params)
            return PFPN_ERR_ARG;
        gr.off[k] = group_off[k]; gr.len[k] = group_len[k]; gr.lr[k] = group_lr[k];
    }
    hipLaunchKernelGGL(k_shard_adam, dim3(stream_grid((size_t)shard_len >> 2)), dim3(256), 0, (hipStream_t)stream,
                       params_x, m, v, recv, scales, world, (size_t)shard_lo >> 2, (size_t)shard_len >> 2,
                       (size_t)n_params >> 2, (size_t)n_bucket >> 2, gr, beta1, beta2, eps, beta_pow, step_counter);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_adam_groups(float* params, const float* grads, float* m, float* v, size_t n, int n_groups,
                     const long long* group_off, const long long* group_len, const float* group_lr, float beta1,
                     float beta2, float eps, const float* beta_pow, float grad_mul, float* step_counter,
                     float* scale_buf, size_t scale_n, float scale_mul, float* soft_target, long long soft_off,
                     long long soft_len, float soft_tau, void* stream) {
    PFPN_BEGIN();
    if (!params || !grads || !m || !v || !beta_pow || !group_off || !group_len || !group_lr || n < 1 || n_groups < 1 ||
        n_groups > 4 || (scale_n > 0 && !scale_buf))
        return PFPN_ERR_ARG;
    if (soft_target && (soft_off < 0 || soft_len < 0 || (soft_off & 3) || (soft_len & 3) ||
                        (size_t)(soft_off + soft_len) > n || ((uintptr_t)soft_target & 15)))
        return PFPN_ERR_ARG;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) return PFPN_ERR_ARG;
    AdamGroups gr{};
    gr.n = n_groups;
    size_t longest = 0;
    for (int k = 0; k < n_groups; ++k) {
        if (group_off[k] < 0 || group_len[k] < 0 || (group_off[k] & 3) || (group_len[k] & 3) ||
            (size_t)(group_off[k] + group_len[k]) > n)
            return PFPN_ERR_ARG;
        gr.off[k] = group_off[k]; gr.len[k] = group_len[k]; gr.lr[k] = group_lr[k];
        if ((size_t)group_len[k] > longest) longest = (size_t)group_len[k];
    }
    hipLaunchKernelGGL(k_clip_adam, dim3(stream_grid(longest >> 2)), dim3(256), 0, (hipStream_t)stream, params, grads, m,
                       v, gr, beta1, beta2, eps, beta_pow, (const float*)nullptr, 0, 0.f, grad_mul, (float*)nullptr,
                       step_counter, scale_buf, scale_n, scale_mul, soft_target, (size_t)soft_off >> 2,
                       (size_t)soft_len >> 2, soft_tau);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_sac_loss(const float* q_pi, const float* q_run, const float* q_tgt, const float* logp, const float* logp_t,
                  const float* reward, const float* not_terminal, const float* log_alpha, int B, float gamma,
                  float value_coef, float target_entropy, float* dq_pi, float* dq_run, float* dlogp,
                  float* dlog_alpha, float* loss3, int q_ld, uint64_t* noise_epoch, void* stream) {
    PFPN_BEGIN();
    if (!q_pi || !q_run || !q_tgt || !logp || !logp_t || !reward || !not_terminal || !log_alpha || !dq_pi ||
        !dq_run || !dlogp || !dlog_alpha || !loss3 || B < 1 || (q_ld != 0 && q_ld < B))
        return PFPN_ERR_ARG;
    hipLaunchKernelGGL(k_sac_loss, dim3(1), dim3(1024), 0, (hipStream_t)stream, q_pi, q_run, q_tgt, logp, logp_t,
                       reward, not_terminal, log_alpha, B, gamma, value_coef, target_entropy, dq_pi, dq_run, dlogp,
                       dlog_alpha, loss3, q_ld ? q_ld : B, (unsigned long long*)noise_epoch);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_sac_q_inputs(const float* x, const float* x_next, const float* a_pi, const float* a_hist, const float* a_next,
                      int B, int S, int A, float* xa_pi, float* xa_hist, float* xa_next, void* stream) {
    PFPN_BEGIN();
    if (!x || !x_next || !a_pi || !a_hist || !a_next || !xa_pi || !xa_hist || !xa_next || B < 1 || S < 1 || A < 1)
        return PFPN_ERR_ARG;
    const long long n = 3LL * B * (S + A);
    hipLaunchKernelGGL(k_sac_q_inputs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_next,
                       a_pi, a_hist, a_next, B, S, A, xa_pi, xa_hist, xa_next);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_gather_columns(const void* const* srcs, void* const* dsts, const int* widths, int ncol, const int64_t* rows,
                        long long n_src, int B, void* stream) {
    PFPN_BEGIN();
    if (!srcs || !dsts || !widths || !rows || ncol < 1 || ncol > 8 || B < 1 || n_src < 1) return PFPN_ERR_ARG;
    GatherCols c;
    c.ncol = ncol;
    c.start[0] = 0;
    for (int k = 0; k < ncol; ++k) {
        if (!srcs[k] || !dsts[k] || widths[k] < 1) return PFPN_ERR_ARG;
        c.src[k] = (const float*)srcs[k];
        c.dst[k] = (float*)dsts[k];
        c.width[k] = widths[k];
        c.start[k + 1] = c.start[k] + widths[k];
    }
    const size_t total = (size_t)B * (size_t)c.start[ncol];
    hipLaunchKernelGGL(k_gather_columns, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, c, rows,
                       (long)n_src, B);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_soft_update(float* target, const float* source, size_t n, float tau, void* stream) {
    PFPN_BEGIN();
    if (!target || !source) return PFPN_ERR_ARG;
    if (n == 0) return PFPN_OK;
    hipLaunchKernelGGL(k_soft_update, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, target, source, n,
                       tau);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

}  // extern "C"
