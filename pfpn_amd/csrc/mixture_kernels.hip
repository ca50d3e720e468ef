// Particle-mixture log-prob kernels for gfx950 (MI355X): forward, backward and the fused
// PPO forward+loss+backward, i.e. HIP units K1 / K2 / K7 of DESIGN.md.
//
// Replaces the TF op subgraph of MixtureGaussianDistribution.log_prob
// (reference networks/utils.py:108-134, op order in SURVEY.md Appendix A) and its autodiff,
// plus the clipped-PPO loss head (reference networks/actor_critic/ppo.py:31-54,
// actor_critic.py:151-156).
//
// Two data paths:
//  * "packed" (N <= 64, templated on N; the shipped N=35 / default N=50): a 256-thread block
//    stages R whole batch rows (R*A*N contiguous floats, float4-coalesced) in LDS, one THREAD owns
//    one (row, action-dim) segment and walks its N particles out of LDS with everything in
//    registers; the particle bank {loc, 1/scale, 0.5log2pi+log scale} is LDS-resident as float4
//    (conflict-free b128 reads at stride 35).  Because a block owns complete rows, the row-level
//    log-prob, the PPO ratio/clip and the backward run in ONE pass: logits are read from HBM
//    once and dlogits written once (2L traffic instead of 3L).
//  * "wave" (any N): one 64-lane wavefront owns one segment, lanes stride the particles; a
//    wave keeps a fixed action dim so dloc/dlogstd accumulate in registers.
// Reductions over the batch (dloc, dlogstd, loss sums) go through per-block partial slabs and
// a fixed-order finalize kernel: results are bitwise reproducible (no float atomics).
#include "pfpn_common.h"

namespace pfpn {

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_PPO = 2 };

struct PpoArgs {
    const float* logp_old;   // [B]
    const float* adv;        // [B]
    const float* value;      // [B]
    const float* value_old;  // [B]
    const float* adv_stats;  // [2] mean, 1/(std+1e-8)   (device)
    float inv_B;             // 1/B (rank-local batch, reference per-worker minibatch mean)
    float clip_lo, clip_hi;  // 1-eps, 1+eps
    float value_coef;        // 0.5
    float* dvalue;           // [B] out
    float* ratio_out;        // [B] out (optional, may be null)
};

constexpr int PACK_G = 4;                            // lanes per (row, action-dim) segment
constexpr int PACK_BLOCK = 1024;                     // 16 waves share one tile
constexpr int PACK_SEGS = PACK_BLOCK / PACK_G;       // 256 segments per tile
constexpr int PACK_MAXV = 3;                         // float4 per thread per tile
constexpr int PACK_TILE_BYTES = PACK_BLOCK * PACK_MAXV * 16;   // 48 KiB of logits per tile

// ---------------------------------------------------------------------------------------
// packed path
// ---------------------------------------------------------------------------------------
template <int N, int MODE>
__global__ __launch_bounds__(PACK_BLOCK) void k_mix_packed(
    const float* __restrict__ logits, const float* __restrict__ x,
    const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ dlogp_in,      // MODE_BWD: [B]
    int B, int A, int R, int tanh_mode, int vec_ok,
    float* __restrict__ logp_out,            // [B]   (FWD, PPO)
    float* __restrict__ pmix_out,            // [B,A] optional
    float* __restrict__ dlogits,             // [B,A,N] (BWD, PPO)
    float* __restrict__ dx_out,              // [B,A] optional (BWD)
    float* __restrict__ partial,             // [grid][2*A*N + 4] (BWD, PPO)
    PpoArgs ppo, int dbg)
{
    constexpr int NP = (N + PACK_G - 1) / PACK_G;         // particles per lane (9 for N=35)
    extern __shared__ float4 smem4[];
    const int AN = A * N;
    const int tile_floats = R * AN;
    const int tile_f4 = (tile_floats + 3) >> 2;
    float* tile = reinterpret_cast<float*>(smem4);
    float4* bank = smem4 + tile_f4;                       // [AN] {loc, isc2, cterm2, isc*c}  (see below)
    float* segbuf = reinterpret_cast<float*>(bank + AN);  // [R*A] per-segment log p
    float* rowbuf = segbuf + R * A;                       // [R]   per-row dlogp
    float* lossbuf = rowbuf + R;                          // [2R]

    const int t = threadIdx.x;
    // particle bank -> LDS.  scale = exp(logstd), cterm = 0.5log(2pi) + log(scale) as the reference
    // graph computes them; then everything is pre-scaled to base 2 so the per-particle work is
    //   z' = (x - loc) * isc2,  phi = exp2(-z'^2 - cterm2)      (3 VALU + 1 transcendental)
    // with isc2 = sqrt(0.5*log2 e)/scale and cterm2 = cterm*log2 e; the true z = c*z', c = sqrt(2 ln 2).
    for (int i = t; i < AN; i += PACK_BLOCK) {
        const float sc = expf(logstd[i]);
        const float isc = 1.0f / sc;
        bank[i] = make_float4(loc[i], isc * PFPN_SQRT_HALF_LOG2E, (PFPN_HALF_LOG_2PI + logf(sc)) * PFPN_LOG2E,
                              isc * PFPN_SQRT_2LN2);
    }

    const int sidx = t >> 2;                // segment slot in the tile
    const int g = t & 3;                    // lane in the segment: particles g, g+4, g+8, ...
    const int nseg = R * A;                 // <= PACK_SEGS
    const bool seg_thread = sidx < nseg;
    const int r = seg_thread ? sidx / A : 0;
    const int a = seg_thread ? sidx - r * A : 0;
    const float4* bk = bank + a * N + g;    // particle g+4i at bk[4i]
    float* seg = tile + sidx * N + g;       // (r*A + a)*N == sidx*N
    const bool last_ok = (g + PACK_G * (NP - 1)) < N;     // does my NP-th particle exist?

    float acc_loc[NP], acc_ls[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) { acc_loc[i] = 0.f; acc_ls[i] = 0.f; }
    float pol_sum = 0.f, val_sum = 0.f;
    float adv_mean = 0.f, adv_rstd = 1.f;
    if (MODE == MODE_PPO) { adv_mean = ppo.adv_stats[0]; adv_rstd = ppo.adv_stats[1]; }

    const int ntiles = (B + R - 1) / R;
    const int wave = t >> 6, lane = t & 63;
    // Software pipeline: the NEXT tile's logits are fetched into registers (3 float4 per thread)
    // while the current tile is processed, so the HBM latency hides behind the compute phases.
    float4 buf[PACK_MAXV];
    auto fetch = [&](int tileid) {
        const int row0 = tileid * R;
        const int nv = (min(R, B - row0) * AN) >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(logits + (size_t)row0 * AN);
#pragma unroll
        for (int j = 0; j < PACK_MAXV; ++j)        // clamped index: all loads unconditional
            buf[j] = g4[min(t + j * PACK_BLOCK, nv - 1)];
    };
    if (vec_ok && (int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tileid = blockIdx.x; tileid < ntiles; tileid += gridDim.x) {
        const int row0 = tileid * R;
        const int rows = min(R, B - row0);
        const int nfl = rows * AN;
        __syncthreads();   // previous tile's LDS reads (store-out) are done; bank is written
        if (vec_ok) {
            const int nv = nfl >> 2;            // AN % 4 == 0 guaranteed by vec_ok
#pragma unroll
            for (int j = 0; j < PACK_MAXV; ++j) {
                const int i = t + j * PACK_BLOCK;
                if (i < nv) reinterpret_cast<float4*>(tile)[i] = buf[j];
            }
            if (tileid + (int)gridDim.x < ntiles) fetch(tileid + gridDim.x);
        } else {
            const float* gsrc = logits + (size_t)row0 * AN;
            for (int i = t; i < nfl; i += PACK_BLOCK) tile[i] = gsrc[i];
        }
        const bool active = seg_thread && r < rows;
        float xv = 0.f;
        if (active) xv = x[(size_t)row0 * A + sidx];
        // row scalars for the row stage: wave w owns row w of the tile; fetched now, used later
        float r_dlogp = 0.f, r_adv = 0.f, r_lpold = 0.f, r_val = 0.f, r_vold = 0.f;
        if (wave < rows && lane == 0) {
            const int b = row0 + wave;
            if (MODE == MODE_BWD) r_dlogp = dlogp_in[b];
            if (MODE == MODE_PPO) {
                r_adv = ppo.adv[b]; r_lpold = ppo.logp_old[b]; r_val = ppo.value[b]; r_vold = ppo.value_old[b];
            }
        }
        __syncthreads();

        // ---- forward on my quarter of the segment ---------------------------------------
        // exp(l - max) goes back to the LDS slot (the logits are not needed again), e*phi
        // stays in registers for the backward.
        float ephi[NP];
        float inv_sum = 0.f, p = 0.f;
        if (active && !(dbg & 1)) {
            float mx = seg[0];
#pragma unroll
            for (int i = 1; i < NP; ++i)
                if (i < NP - 1 || last_ok) mx = fmaxf(mx, seg[PACK_G * i]);
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
            const float mxl = mx * PFPN_LOG2E;
            float sum = 0.f, num = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                ephi[i] = 0.f;
                if (i < NP - 1 || last_ok) {
                    const float4 k4 = bk[PACK_G * i];
                    // softmax numerator: raw v_exp_f32 (tiny values may flush; they carry no mass)
                    const float en = __builtin_amdgcn_exp2f(fmaf(seg[PACK_G * i], PFPN_LOG2E, -mxl));
                    const float zq = (xv - k4.x) * k4.y;
                    const float ph = exp2f(fmaf(-zq, zq, -k4.z));      // denormal-correct
                    if (MODE != MODE_FWD) seg[PACK_G * i] = en;
                    ephi[i] = en * ph;
                    sum += en;
                    num += ephi[i];
                }
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            num += __shfl_xor(num, 1, 64);
            num += __shfl_xor(num, 2, 64);
            inv_sum = 1.0f / sum;
            p = num * inv_sum;
            if (g == 0) {
                float lp = logf(p);
                if (tanh_mode) lp -= tanh_log_det(xv);
                if (pmix_out) pmix_out[(size_t)row0 * A + sidx] = p;
                segbuf[sidx] = lp;
            }
        }
        __syncthreads();

        // ---- row stage: wave w reduces the A segment log-probs of rows w, w+16, ... (shuffle tree,
        //      fixed order) and its lane 0 runs the PPO head (first row: prefetched scalars)
        for (int rw = wave; rw < rows && !(dbg & 8); rw += PACK_BLOCK / 64) {
            float sacc = 0.f;
            for (int i = lane; i < A; i += 64) sacc += segbuf[rw * A + i];
            const float s = wave_sum(sacc);
            if (lane == 0) {
                const int b = row0 + rw;
                if (rw != wave) {           // only when a tile holds more than 16 rows (small A)
                    if (MODE == MODE_BWD) r_dlogp = dlogp_in[b];
                    if (MODE == MODE_PPO) {
                        r_adv = ppo.adv[b]; r_lpold = ppo.logp_old[b]; r_val = ppo.value[b]; r_vold = ppo.value_old[b];
                    }
                }
                if (MODE != MODE_BWD) logp_out[b] = s;
                if (MODE == MODE_BWD) rowbuf[rw] = r_dlogp;
                if (MODE == MODE_PPO) {
                    const float nadv = (r_adv - adv_mean) * adv_rstd;
                    const float ratio = expf(s - r_lpold);
                    const float sur = ratio * nadv;
                    const float cr = fmaxf(fminf(ratio, ppo.clip_hi), ppo.clip_lo);
                    const float csur = cr * nadv;
                    pol_sum += fminf(sur, csur);
                    // grad of -mean(min(sur, csur)): tf.minimum routes to `sur` where sur <= csur,
                    // clip_by_value passes where lo <= ratio <= hi.
                    const bool pass = (sur <= csur) || (ratio >= ppo.clip_lo && ratio <= ppo.clip_hi);
                    rowbuf[rw] = pass ? (-ppo.inv_B * nadv) * ratio : 0.f;
                    const float dvv = r_val - (r_adv + r_vold);
                    val_sum += dvv * dvv;
                    ppo.dvalue[b] = 2.f * dvv * ppo.value_coef * ppo.inv_B;
                    if (ppo.ratio_out) ppo.ratio_out[b] = ratio;
                }
            }
        }
        if (MODE == MODE_FWD) continue;
        __syncthreads();

        // ---- backward on my quarter (dlogits overwrite the LDS slot) ---------------------
        if (active && !(dbg & 2)) {
            const float gup = rowbuf[r];
            const float q = guard_finite(gup / p);
            const float qs = q * inv_sum;
            const float qp = qs * p;
            float dxa = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (i < NP - 1 || last_ok) {
                    const float4 k4 = bk[PACK_G * i];
                    const float tt = qs * ephi[i];                          // q * w_n * phi_n
                    seg[PACK_G * i] = fmaf(-seg[PACK_G * i], qp, tt);      // q * w_n * (phi_n - p)
                    const float zq = (xv - k4.x) * k4.y;                    // z / c
                    acc_loc[i] = fmaf(tt, zq, acc_loc[i]);                  // scaled by isc*c in the tail
                    acc_ls[i] = fmaf(tt, fmaf(zq * zq, PFPN_2LN2, -1.f), acc_ls[i]);   // tt*(z^2 - 1)
                    if (dx_out) dxa = fmaf(tt * zq, k4.w, dxa);
                }
            }
            if (dx_out) {
                dxa += __shfl_xor(dxa, 1, 64);
                dxa += __shfl_xor(dxa, 2, 64);
                if (g == 0) {
                    float d = -dxa;
                    if (tanh_mode) d += gup * 2.f * tanhf(xv);
                    dx_out[(size_t)row0 * A + sidx] = d;
                }
            }
        }
        __syncthreads();
        float* gdst = dlogits + (size_t)row0 * AN;
        if (dbg & 4) continue;
        if (vec_ok) {
            const int nv = nfl >> 2;
#pragma unroll
            for (int j = 0; j < PACK_MAXV; ++j) {
                const int i = t + j * PACK_BLOCK;
                if (i < nv) reinterpret_cast<float4*>(gdst)[i] = reinterpret_cast<const float4*>(tile)[i];
            }
        } else {
            for (int i = t; i < nfl; i += PACK_BLOCK) gdst[i] = tile[i];
        }
    }

    if (MODE == MODE_FWD) return;
    // ---- block partials: dloc / dlogstd summed over the R segments that share an action dim,
    //      in row order; loss sums in row-slot order.  Slab layout [2*AN + 4] per block.
    float* slab = partial + (size_t)blockIdx.x * (2 * AN + 4);
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
        if (seg_thread) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (i < NP - 1 || last_ok) seg[PACK_G * i] = which ? acc_ls[i] : acc_loc[i];
        }
        __syncthreads();
        for (int i = t; i < AN; i += PACK_BLOCK) {
            float s = 0.f;
            for (int rr = 0; rr < R; ++rr) s += tile[rr * AN + i];
            slab[which * AN + i] = which ? s : s * bank[i].w;      // dloc: sum tt*z' times c/scale
        }
    }
    if (MODE == MODE_PPO) {
        const int nwr = min(R, PACK_BLOCK / 64);          // waves that own rows
        if (wave < nwr && lane == 0) { lossbuf[wave] = pol_sum; lossbuf[nwr + wave] = val_sum; }
        __syncthreads();
        if (t == 0) {
            float ps = 0.f, vs = 0.f;
            for (int i = 0; i < nwr; ++i) { ps += lossbuf[i]; vs += lossbuf[nwr + i]; }
            slab[2 * AN + 0] = ps;
            slab[2 * AN + 1] = vs;
        }
    } else if (t == 0) {
        slab[2 * AN + 0] = 0.f;
        slab[2 * AN + 1] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------
// wave path (any N): wave (a, chunk c) walks rows b = c, c+C, ...
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_mix_wave(
    const float* __restrict__ logits, const float* __restrict__ x,
    const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ dlogp_in,
    int B, int A, int N, int C, int tanh_mode,
    float* __restrict__ lpseg_out,           // [B,A] per-segment log p (FWD)
    float* __restrict__ pmix_out,            // [B,A] optional
    float* __restrict__ dlogits,             // BWD
    float* __restrict__ dx_out,              // BWD optional
    float* __restrict__ partial)             // [C][2*A*N]  (BWD)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= A * C) return;
    const int a = wave % A;
    const int c = wave / A;
    const int AN = A * N;
    const float* locr = loc + a * N;
    const float* lsr = logstd + a * N;
    // dloc/dlogstd accumulators for particles lane, lane+64, ...  (written per 64-chunk below)
    for (int n0 = 0; n0 < N; n0 += 64) {
        if (MODE == MODE_BWD && n0 + lane < N) {
            partial[(size_t)c * 2 * AN + a * N + n0 + lane] = 0.f;
            partial[(size_t)c * 2 * AN + AN + a * N + n0 + lane] = 0.f;
        }
    }
    for (int b = c; b < B; b += C) {
        const float* row = logits + (size_t)b * AN + a * N;
        const float xv = x[(size_t)b * A + a];
        float mx = -INFINITY;
        for (int n = lane; n < N; n += 64) mx = fmaxf(mx, row[n]);
        mx = wave_max(mx);
        float sum = 0.f, num = 0.f;
        for (int n = lane; n < N; n += 64) {
            const float sc = expf(lsr[n]);
            const float en = expf(row[n] - mx);
            const float z = (xv - locr[n]) * (1.0f / sc);
            const float ph = expf(fmaf(-0.5f * z, z, -(PFPN_HALF_LOG_2PI + logf(sc))));
            sum += en;
            num = fmaf(en, ph, num);
        }
        sum = wave_sum(sum);
        num = wave_sum(num);
        const float inv_sum = 1.0f / sum;
        const float p = num * inv_sum;
        if (MODE == MODE_FWD) {
            if (lane == 0) {
                float lp = logf(p);
                if (tanh_mode) lp -= tanh_log_det(xv);
                lpseg_out[(size_t)b * A + a] = lp;
                if (pmix_out) pmix_out[(size_t)b * A + a] = p;
            }
        } else {
            const float g = dlogp_in[b];
            const float q = guard_finite(g / p);
            const float qs = q * inv_sum;
            float dxa = 0.f;
            float* drow = dlogits + (size_t)b * AN + a * N;
            for (int n = lane; n < N; n += 64) {
                const float sc = expf(lsr[n]);
                const float isc = 1.0f / sc;
                const float en = expf(row[n] - mx);
                const float z = (xv - locr[n]) * isc;
                const float ph = expf(fmaf(-0.5f * z, z, -(PFPN_HALF_LOG_2PI + logf(sc))));
                const float tt = qs * en * ph;
                drow[n] = tt - (qs * en) * p;
                const float zi = z * isc;
                partial[(size_t)c * 2 * AN + a * N + n] += tt * zi;
                partial[(size_t)c * 2 * AN + AN + a * N + n] += tt * fmaf(z, z, -1.f);
                dxa = fmaf(tt, zi, dxa);
            }
            if (dx_out) {
                dxa = wave_sum(dxa);
                if (lane == 0) {
                    float d = -dxa;
                    if (tanh_mode) d += g * 2.f * tanhf(xv);
                    dx_out[(size_t)b * A + a] = d;
                }
            }
        }
    }
}

// logp[b] = sum_a lpseg[b,a], left to right (deterministic)
__global__ void k_row_sum(const float* __restrict__ lpseg, int B, int A, float* __restrict__ logp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int a = 0; a < A; ++a) s += lpseg[(size_t)b * A + a];
    logp[b] = s;
}

// Fixed-order reduction of per-block slabs: out[i] = sum_k partial[k*stride + i], i < n.
// Also emits the loss scalars of the fused PPO kernel when `loss_out` is given:
// loss_out = {loss, policy_loss, value_loss*coef}.  Grid: ceil((n+2)/64) blocks of 1024.
__global__ __launch_bounds__(1024) void k_finalize(const float* __restrict__ partial, int nslab, int stride, int n,
                                                   float* __restrict__ out0, float* __restrict__ out1, int n0,
                                                   float* __restrict__ loss_out, float inv_B, float value_coef) {
    __shared__ float red[16][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ncol = loss_out ? n + 2 : n;           // the two loss sums sit right behind the n columns
    const float s = slab_reduce_1024<false>(partial, nslab, (size_t)stride, i, i < ncol, red);
    if ((threadIdx.x >> 6) != 0 || i >= ncol) return;
    if (i < n) {
        if (i < n0) out0[i] = s; else out1[i - n0] = s;
    } else if (i == n) {
        loss_out[1] = -s * inv_B;                     // policy loss
    } else {
        loss_out[2] = s * inv_B * value_coef;         // value loss * coef
    }
}

__global__ void k_loss_total(float* __restrict__ loss3) {
    if (threadIdx.x == 0 && blockIdx.x == 0) loss3[0] = loss3[1] + loss3[2];
}

// adv_stats = {mean, 1/(sqrt(var)+1e-8)} with population variance (reference
// networks/actor_critic/actor_critic.py:151-156).  One block, fixed order.
__global__ __launch_bounds__(1024) void k_adv_stats(const float* __restrict__ adv, int B, int normalize,
                                                    float* __restrict__ stats) {
    __shared__ float red[16];
    if (!normalize) {
        if (threadIdx.x == 0) { stats[0] = 0.f; stats[1] = 1.f; }
        return;
    }
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) s += adv[i];
    const float mean = block_sum(s, red) / (float)B;
    float v = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { const float d = adv[i] - mean; v = fmaf(d, d, v); }
    const float var = block_sum(v, red) / (float)B;
    if (threadIdx.x == 0) { stats[0] = mean; stats[1] = 1.0f / (sqrtf(var) + 1e-8f); }
}

// ---------------------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------------------
struct PackPlan { int R; int grid; size_t lds; int vec_ok; bool ok; };
static int g_dbg = 0;        // diagnostic phase mask (pfpn_debug_set); 0 in production

static PackPlan plan_packed(int B, int A, int N, const void* logits, const void* dlogits) {
    PackPlan pl{};
    pl.ok = false;
    if (N > 64 || A > PACK_SEGS || A < 1 || B < 1) return pl;
    const int AN = A * N;
    int R = PACK_SEGS / A;
    const int rcap = PACK_TILE_BYTES / (AN * 4);
    if (rcap < 1) return pl;
    if (R > rcap) R = rcap;
    if (R > B) R = B;
    pl.R = R;
    const int tile_f4 = (R * AN + 3) / 4;
    pl.lds = (size_t)tile_f4 * 16 + (size_t)AN * 16 + (size_t)(R * A + R + 2 * R + 4) * 4;
    pl.vec_ok = (AN % 4 == 0) && ((uintptr_t)logits % 16 == 0) && (dlogits == nullptr || (uintptr_t)dlogits % 16 == 0);
    const int ntiles = (B + R - 1) / R;
    // one resident 1024-thread block per CU, tiles strided over the blocks (max load = ceil(ntiles/256))
    pl.grid = ntiles < 256 ? ntiles : 256;
    pl.ok = pl.lds <= 64 * 1024;
    return pl;
}

template <int MODE>
static int launch_packed(int N, const PackPlan& pl, hipStream_t st, const float* logits, const float* x,
                         const float* loc, const float* logstd, const float* dlogp, int B, int A, int tanh_mode,
                         float* logp, float* pmix, float* dlogits, float* dx, float* partial, PpoArgs ppo) {
#define PFPN_LAUNCH_PACKED(NN)                                                                          \
    hipLaunchKernelGGL((k_mix_packed<NN, MODE>), dim3(pl.grid), dim3(PACK_BLOCK), pl.lds, st, logits, x, \
                       loc, logstd, dlogp, B, A, pl.R, tanh_mode, pl.vec_ok, logp, pmix, dlogits, dx,   \
                       partial, ppo, g_dbg)
    switch (N) {
        case 35: PFPN_LAUNCH_PACKED(35); break;
        case 50: PFPN_LAUNCH_PACKED(50); break;
        case 16: PFPN_LAUNCH_PACKED(16); break;
        case 8: PFPN_LAUNCH_PACKED(8); break;
        default: return -1;
    }
#undef PFPN_LAUNCH_PACKED
    return 0;
}

static bool packed_has(int N) { return N == 35 || N == 50 || N == 16 || N == 8; }

}  // namespace pfpn

using namespace pfpn;

// =======================================================================================
// C ABI (declared in include/pfpn_hip.h)
// =======================================================================================
extern "C" {

void pfpn_debug_set(int mask) { g_dbg = mask; }

size_t pfpn_mixture_workspace_bytes(int B, int A, int N) {
    // slabs: packed path <= 768 blocks x (2AN+4); wave path C<=64 chunks x 2AN; + lpseg [B*A]
    const size_t AN = (size_t)A * N;
    const size_t slabs = (size_t)768 * (2 * AN + 4);
    return (slabs + (size_t)B * A + 64) * sizeof(float);
}

int pfpn_logprob_fwd(const float* logits, const float* x, const float* loc, const float* logstd,
                     int B, int A, int N, int tanh_mode, float* logp, float* pmix,
                     void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!logits || !x || !loc || !logstd || !logp || B < 0 || A < 1 || N < 1) return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    hipStream_t st = (hipStream_t)stream;
    PackPlan pl = plan_packed(B, A, N, logits, nullptr);
    if (pl.ok && packed_has(N)) {
        PpoArgs none{};
        launch_packed<MODE_FWD>(N, pl, st, logits, x, loc, logstd, nullptr, B, A, tanh_mode, logp, pmix,
                                nullptr, nullptr, nullptr, none);
        PFPN_CHECK_LAUNCH();
        return PFPN_OK;
    }
    if (!workspace || workspace_bytes < (size_t)B * A * sizeof(float)) return PFPN_ERR_WORKSPACE;
    float* lpseg = (float*)workspace;
    int C = (B < 64) ? B : 64;
    while ((size_t)A * C < 2048 && C < B) C = (C * 2 < B) ? C * 2 : B;
    const int waves = A * C;
    const int blocks = (waves * 64 + 255) / 256;
    hipLaunchKernelGGL((k_mix_wave<MODE_FWD>), dim3(blocks), dim3(256), 0, st, logits, x, loc, logstd,
                       (const float*)nullptr, B, A, N, C, tanh_mode, lpseg, pmix, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_row_sum, dim3((B + 255) / 256), dim3(256), 0, st, lpseg, B, A, logp);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_logprob_bwd(const float* logits, const float* x, const float* loc, const float* logstd,
                     const float* dlogp, int B, int A, int N, int tanh_mode,
                     float* dlogits, float* dloc, float* dlogstd, float* dx,
                     void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!logits || !x || !loc || !logstd || !dlogp || !dlogits || !dloc || !dlogstd || B < 0 || A < 1 || N < 1)
        return PFPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int AN = A * N;
    if (B == 0) {
        (void)hipMemsetAsync(dloc, 0, AN * sizeof(float), st);
        (void)hipMemsetAsync(dlogstd, 0, AN * sizeof(float), st);
        return PFPN_OK;
    }
    if (!workspace) return PFPN_ERR_WORKSPACE;
    float* partial = (float*)workspace;
    PackPlan pl = plan_packed(B, A, N, logits, dlogits);
    if (pl.ok && packed_has(N)) {
        if (workspace_bytes < (size_t)pl.grid * (2 * AN + 4) * sizeof(float)) return PFPN_ERR_WORKSPACE;
        PpoArgs none{};
        launch_packed<MODE_BWD>(N, pl, st, logits, x, loc, logstd, dlogp, B, A, tanh_mode, nullptr, nullptr,
                                dlogits, dx, partial, none);
        PFPN_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_finalize, dim3((2 * AN + 63) / 64), dim3(1024), 0, st, partial, pl.grid,
                           2 * AN + 4, 2 * AN, dloc, dlogstd, AN, (float*)nullptr, 0.f, 0.f);
        PFPN_CHECK_LAUNCH();
        return PFPN_OK;
    }
    int C = (B < 64) ? B : 64;
    if (workspace_bytes < (size_t)C * 2 * AN * sizeof(float)) return PFPN_ERR_WORKSPACE;
    const int waves = A * C;
    const int blocks = (waves * 64 + 255) / 256;
    hipLaunchKernelGGL((k_mix_wave<MODE_BWD>), dim3(blocks), dim3(256), 0, st, logits, x, loc, logstd, dlogp,
                       B, A, N, C, tanh_mode, (float*)nullptr, (float*)nullptr, dlogits, dx, partial);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_finalize, dim3((2 * AN + 63) / 64), dim3(1024), 0, st, partial, C, 2 * AN, 2 * AN,
                       dloc, dlogstd, AN, (float*)nullptr, 0.f, 0.f);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_adv_stats(const float* adv, int B, int normalize, float* stats, void* stream) {
    PFPN_BEGIN();
    if (!adv || !stats || B < 1) return PFPN_ERR_ARG;
    hipLaunchKernelGGL(k_adv_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, adv, B, normalize, stats);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_ppo_mixture_fwd_bwd(const float* logits, const float* action, const float* loc, const float* logstd,
                             const float* logp_old, const float* adv, const float* value,
                             const float* value_old, const float* adv_stats,
                             int B, int A, int N, float clip_eps, float value_coef,
                             float* logp, float* dlogits, float* dloc, float* dlogstd, float* dvalue,
                             float* loss3, void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!logits || !action || !loc || !logstd || !logp_old || !adv || !value || !value_old || !adv_stats ||
        !logp || !dlogits || !dloc || !dlogstd || !dvalue || !loss3 || B < 1 || A < 1 || N < 1)
        return PFPN_ERR_ARG;
    if (!workspace) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int AN = A * N;
    PackPlan pl = plan_packed(B, A, N, logits, dlogits);
    if (!(pl.ok && packed_has(N))) return PFPN_ERR_ARG;   // caller falls back to fwd + ppo_loss + bwd
    if (workspace_bytes < (size_t)pl.grid * (2 * AN + 4) * sizeof(float)) return PFPN_ERR_WORKSPACE;
    PpoArgs pa{};
    pa.logp_old = logp_old; pa.adv = adv; pa.value = value; pa.value_old = value_old; pa.adv_stats = adv_stats;
    pa.inv_B = 1.0f / (float)B; pa.clip_lo = 1.0f - clip_eps; pa.clip_hi = 1.0f + clip_eps;
    pa.value_coef = value_coef; pa.dvalue = dvalue; pa.ratio_out = nullptr;
    float* partial = (float*)workspace;
    launch_packed<MODE_PPO>(N, pl, st, logits, action, loc, logstd, nullptr, B, A, 0, logp, nullptr, dlogits,
                            nullptr, partial, pa);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_finalize, dim3((2 * AN + 2 + 63) / 64), dim3(1024), 0, st, partial, pl.grid, 2 * AN + 4,
                       2 * AN, dloc, dlogstd, AN, loss3, pa.inv_B, value_coef);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_loss_total, dim3(1), dim3(64), 0, st, loss3);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

}  // extern "C"
