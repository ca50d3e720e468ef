// SAC-PFPN actor side of the learner step as TWO launches (reference networks/actor_critic/sac.py:108-175 with
// networks/utils.py:108-134, :156-186): the reparameterised sample, its log-probability and the three critic inputs in one
// forward launch; the log-prob backward, the sample backward (custom gradients `mask` / `mask2`), the particle-grid
// gradients and the head-bias gradient in one backward launch.  The step is launch-bound at replay batches of 256-1024 rows
// (~5 us per dependent launch): this pair stands for seven launches of the piece-wise form (k_rsample_fwd_quad,
// k_mix_tile<FWD>, k_sac_q_inputs | k_mix_tile<BWD>, k_rsample_bwd_quad, k_rsample_param_partial, k_act_bwd_partial).
//
// Layout: 4 lanes per (row, action dim) segment, lane g owns particles g, g + 4, ... in registers (N <= 64).  A block owns
// Rb WHOLE rows (Rb * A quads), so (i) its logits / Gumbel / normal draws are one contiguous span per tensor, staged through
// LDS with float4 wave loads, (ii) the row sum of the per-segment log p is a block-local matter, (iii) every lane keeps the
// SAME (action dim, particles) from row group to row group: the particle-grid sums over rows are register accumulators,
// folded over the block's Rb quads of a dim through LDS in fixed order and left as ONE slab per block for the step's
// deferred reduction launch (fixed order -> bitwise reproducible run to run).
#include "pfpn_common.h"
#include "pfpn_hip.h"

namespace pfpn {

constexpr int SA_G = 4;                                  // lanes per segment
constexpr int SA_MAX_LDS_FLOATS = 16384;                 // 64 KiB of dynamic LDS without an attribute call

__device__ __forceinline__ void sa_quad_argmax(float& v, int& i) {
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
        const float ov = __shfl_xor(v, m, 64);
        const int oi = __shfl_xor(i, m, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }   // the FIRST maximum wins, as in the serial rule
    }
}

// floats [base, base + count) of src -> lds[0, count): float4 loads when the span starts on 16 bytes, scalar tail
__device__ __forceinline__ void sa_stage_in(float* __restrict__ lds, const float* __restrict__ src, size_t base, int count,
                                            int tid, int nthr) {
    const int nq = (((uintptr_t)(src + base) & 15) == 0) ? count >> 2 : 0;
    for (int q = tid; q < nq; q += nthr)
        *reinterpret_cast<float4*>(lds + 4 * q) = *reinterpret_cast<const float4*>(src + base + 4 * q);
    for (int i = 4 * nq + tid; i < count; i += nthr) lds[i] = src[base + i];
}
__device__ __forceinline__ void sa_stage_out(float* __restrict__ dst, const float* __restrict__ lds, size_t base, int count,
                                             int tid, int nthr) {
    const int nq = (((uintptr_t)(dst + base) & 15) == 0) ? count >> 2 : 0;
    for (int q = tid; q < nq; q += nthr)
        *reinterpret_cast<float4*>(dst + base + 4 * q) = *reinterpret_cast<const float4*>(lds + 4 * q);
    for (int i = 4 * nq + tid; i < count; i += nthr) dst[base + i] = lds[i];
}

// The particle bank of one lane: loc, logstd, scale, isc = sqrt(.5 log2 e) / scale, ct = 64 - (0.5 log 2 pi + log scale) log2 e
// (k_mix_tile's constants: e * phi = exp2(l log2 e - max - z'^2 + ct) * 2^-64 in ONE v_exp_f32).  The scale comes from one
// v_exp_f32 and its reciprocal from one v_rcp_f32 (2e-7 relative; log scale = logstd as it stands): a block is alive for
// ONE or two row groups, so the bank is not amortised -- with libm's expf / logf / division it was a third of the launch.
__device__ __forceinline__ float sa_fast_tanh(float v) {      // 1 - 2 / (1 + e^(2v)): v_exp_f32 + v_rcp_f32, |error| ~1e-7
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * (2.f * PFPN_LOG2E)));
}
template <int NP>
struct SaBank {
    float loc[NP], ls[NP], sc[NP], isc[NP], ct[NP];
    __device__ __forceinline__ void load(const float* __restrict__ locp, const float* __restrict__ lsp, int a, int N, int g) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int n = g + SA_G * i;
            const int o = a * N + (n < N ? n : 0);
            loc[i] = locp[o];
            ls[i] = lsp[o];
            sc[i] = __builtin_amdgcn_exp2f(ls[i] * PFPN_LOG2E);
            isc[i] = __builtin_amdgcn_rcpf(sc[i]) * PFPN_SQRT_HALF_LOG2E;
            ct[i] = 64.f - (PFPN_HALF_LOG_2PI + ls[i]) * PFPN_LOG2E;
        }
    }
};

// ---- forward ------------------------------------------------------------------------------------------------------------
// rows [0, R) of logits / gumbel / eps ([R, A*N]); R = 2B in the learner step ([s; s']).  k, a = tanh(p_k), s = p_k per
// segment, logp [R] = sum_a (log p_mix(p_k) - log(1 - tanh(p_k)^2)).  With xq_both != nullptr the critic inputs are
// assembled too: row r < B -> xq_both[r] = [x_r | a_r], xq_both[B + r] = [x_r | a_hist_r]; row r >= B -> xq_next[r - B] =
// [x_r | a_r]  (x [R, S]: the normalised states of the same rows).
template <int NP, int MAXT>
__global__ __launch_bounds__(MAXT) void k_sac_actor_fwd(
    const float* __restrict__ logits, const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ gumbel, const float* __restrict__ eps, int R, int A, int N, int Rb,
    int* __restrict__ k_out, float* __restrict__ a_out, float* __restrict__ s_out, float* __restrict__ logp,
    const float* __restrict__ x, const float* __restrict__ a_hist, int B, int S,
    float* __restrict__ xq_both, float* __restrict__ xq_next)
{
    extern __shared__ __align__(16) float sa_lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int AN = A * N;
    const int r0 = blockIdx.x * Rb;
    const int rows = min(Rb, R - r0);
    const int span = (Rb * AN + 3) & ~3;
    float* ll = sa_lds;
    float* lg = ll + span;
    float* s_lp = lg + span;                             // [Rb][A] per-segment log p
    const int W = S + A;
    const int q = tid >> 2, g = tid & 3;
    const int rq_ = q / A, a = q - rq_ * A;
    // Everything that depends on nothing computed here is REQUESTED first -- the state columns of the critic inputs (up to
    // XR per thread in registers), then the staged tensors, then the bank: one memory round trip, not three.
    constexpr int XR = 4;
    float xr[XR];
    const int nx = xq_both ? rows * S : 0;
#pragma unroll
    for (int j = 0; j < XR; ++j) {
        const int idx = tid + j * nthr;
        xr[j] = idx < nx ? x[(size_t)r0 * S + idx] : 0.f;  // (the block's rows of x are one contiguous span)
    }
    sa_stage_in(ll, logits, (size_t)r0 * AN, rows * AN, tid, nthr);
    sa_stage_in(lg, gumbel, (size_t)r0 * AN, rows * AN, tid, nthr);
    SaBank<NP> bk;
    bk.load(loc, logstd, a, N, g);
    if (xq_both) {                                       // the replayed actions: the action columns of Q(s, action_hist)
        for (int idx = tid; idx < rows * A; idx += nthr) {
            const int rr = idx / A, c = idx - rr * A;
            const int r = r0 + rr;
            if (r < B) xq_both[(size_t)(B + r) * W + S + c] = a_hist[(size_t)r * A + c];
        }
    }
    for (int idx0 = 0; idx0 < nx; idx0 += nthr * XR) {    // (one trip unless Rb * S > XR * blockDim)
#pragma unroll
        for (int j = 0; j < XR; ++j) {
            const int idx = idx0 + tid + j * nthr;
            if (idx < nx) {
                const float v = idx0 == 0 ? xr[j] : x[(size_t)r0 * S + idx];
                const int rr = idx / S, c = idx - rr * S;
                const int r = r0 + rr;
                if (r < B) {
                    xq_both[(size_t)r * W + c] = v;
                    xq_both[(size_t)(B + r) * W + c] = v;
                } else {
                    xq_next[(size_t)(r - B) * W + c] = v;
                }
            }
        }
    }
    __syncthreads();
    if (rq_ < rows) {                                    // (a quad is live or dead as a whole: the DPP steps stay inside it)
        const int rq = rq_;
        const float* row = ll + (rq * A + a) * N;
        const float* gr = lg + (rq * A + a) * N;
        float l[NP];
        float mx = -INFINITY, mxl = -INFINITY;
        int k = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int n = g + SA_G * i;
            l[i] = 0.f;
            if (n < N) {
                l[i] = row[n] * PFPN_LOG2E;
                const float y = row[n] + gr[n];
                if (y > mx) { mx = y; k = n; }
                mxl = fmaxf(mxl, l[i]);
            }
        }
        sa_quad_argmax(mx, k);
        const size_t seg = (size_t)(r0 + rq) * A + a;
        const float ek = eps[seg * N + k];               // the one draw this segment uses: requested as soon as k is known,
        mxl = quad_max(mxl);                             // the softmax denominator runs under its round trip
        float e[NP];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            e[i] = l[i] - mxl;
            if (g + SA_G * i < N) sum += __builtin_amdgcn_exp2f(e[i]);
        }
        sum = quad_sum(sum);
        // p_k = loc_k + exp(logstd_k) * eps_k from the lane that owns particle k (libm's expf: the bits of the piece-wise
        // kernel, which the backward recomputes); x + 0 + 0 + 0 is exact
        float lock = 0.f, lsk = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (g + SA_G * i == k) { lock = bk.loc[i]; lsk = bk.ls[i]; }
        float pk = (k & 3) == g ? lock + expf(lsk) * ek : 0.f;
        pk = quad_sum(pk);
        float num = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (g + SA_G * i < N) {
                const float z = (pk - bk.loc[i]) * bk.isc[i];
                num += __builtin_amdgcn_exp2f(e[i] + fmaf(-z, z, bk.ct[i]));
            }
        }
        num = quad_sum(num);
        float p = (num * 5.421010862427522e-20f) * __builtin_amdgcn_rcpf(sum);
        if (p < 1.401298464324817e-45f) p = 0.f;         // below the smallest fp32 denormal: the reference's p is 0
        const float lp = __builtin_amdgcn_logf(p) * 0.6931471805599453f - tanh_log_det(pk);
        const float ta = tanhf(pk);
        if (g == 0) {
            k_out[seg] = k;
            s_out[seg] = pk;
            a_out[seg] = ta;
            s_lp[rq * A + a] = lp;
            if (xq_both) {                                // the sample's column of its critic input
                const int r = r0 + rq;
                if (r < B) xq_both[(size_t)r * W + S + a] = ta;
                else xq_next[(size_t)(r - B) * W + S + a] = ta;
            }
        }
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
    for (int rr = wv; rr < rows; rr += nwv) {             // row sums: one wave per row, fixed-order butterfly (k_row_sum's)
        float s = 0.f;
        for (int aa = lane; aa < A; aa += 64) s += s_lp[rr * A + aa];
        s = wave_sum(s);
        if (lane == 0) logp[r0 + rr] = s;
    }
}

// ---- backward -----------------------------------------------------------------------------------------------------------
// rows [0, B): x [B, A] = the pre-tanh samples p_k, dlogp [B] (upstream of the row's log-prob), da [B, A] (upstream of
// a = tanh(p_k), from the critics); the upstream of s = p_k is the log-prob's own dx (sac.py: the log-prob is evaluated at
// the sample).  dlogits [B, A*N] is WRITTEN (both parts), slabs [gridDim][3 * A*N] = {dloc | dlogstd | d head bias} summed
// over the block's rows.
template <int NP, int MAXT>
__global__ __launch_bounds__(MAXT) void k_sac_actor_bwd(
    const float* __restrict__ logits, const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ gumbel, const float* __restrict__ eps, const float* __restrict__ x,
    const float* __restrict__ dlogp, const float* __restrict__ da, int B, int A, int N, int Rb, int iters,
    float* __restrict__ dlogits, float* __restrict__ slabs)
{
    extern __shared__ __align__(16) float sa_lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int AN = A * N;
    const int span = (Rb * AN + 3) & ~3;
    float* ll = sa_lds;                                  // logits in, dlogits out
    float* lg = ll + span;
    float* le_ = lg + span;
    const int q = tid >> 2, g = tid & 3;
    const int rq_ = q / A, a = q - rq_ * A;
    SaBank<NP> bk;
    bk.load(loc, logstd, a, N, g);
    float acc_loc[NP], acc_ls[NP], acc_b[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) acc_loc[i] = acc_ls[i] = acc_b[i] = 0.f;
    const float two_m64 = 5.421010862427522e-20f;
    for (int it = 0; it < iters; ++it) {
        const int r0 = (blockIdx.x * iters + it) * Rb;
        const int rows = max(0, min(Rb, B - r0));
        float xval = 0.f, gv = 0.f, gda = 0.f;           // (issued before the staging: one round trip, not two)
        if (rq_ < rows) {
            const size_t seg = (size_t)(r0 + rq_) * A + a;
            xval = x[seg];
            gv = dlogp[r0 + rq_];
            gda = da[seg];
        }
        if (it) __syncthreads();                         // the previous group's dlogits have left the LDS
        if (rows > 0) {
            sa_stage_in(ll, logits, (size_t)r0 * AN, rows * AN, tid, nthr);
            sa_stage_in(lg, gumbel, (size_t)r0 * AN, rows * AN, tid, nthr);
            sa_stage_in(le_, eps, (size_t)r0 * AN, rows * AN, tid, nthr);
        }
        __syncthreads();
        if (rq_ < rows) {                                // (a quad is live or dead as a whole: the DPP steps stay inside it)
            const int rq = rq_;
            float* row = ll + (rq * A + a) * N;
            const float* gr = lg + (rq * A + a) * N;
            const float* er = le_ + (rq * A + a) * N;
            // -- log-prob part (k_mix_tile's arithmetic) and the softmax of logits + Gumbel, one pass over the registers
            float lp[NP], y[NP];
            float mxl = -INFINITY, mxy = -INFINITY;
            int k = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int n = g + SA_G * i;
                lp[i] = 0.f; y[i] = -INFINITY;
                if (n < N) {
                    const float lv = row[n];
                    lp[i] = lv * PFPN_LOG2E;
                    mxl = fmaxf(mxl, lp[i]);
                    y[i] = lv + gr[n];
                    if (y[i] > mxy) { mxy = y[i]; k = n; }
                }
            }
            mxl = quad_max(mxl);
            sa_quad_argmax(mxy, k);
            float e[NP], ef[NP];
            float sum = 0.f, num = 0.f, sumy = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                e[i] = ef[i] = 0.f;
                if (g + SA_G * i < N) {
                    const float le = lp[i] - mxl;
                    const float z = (xval - bk.loc[i]) * bk.isc[i];
                    e[i] = __builtin_amdgcn_exp2f(le);
                    ef[i] = __builtin_amdgcn_exp2f(le + fmaf(-z, z, bk.ct[i]));
                    sum += e[i];
                    num += ef[i];
                    y[i] = __builtin_amdgcn_exp2f((y[i] - mxy) * PFPN_LOG2E);   // y <- softmax numerator of logits + Gumbel
                    sumy += y[i];
                } else {
                    y[i] = 0.f;
                }
            }
            sum = quad_sum(sum);
            num = quad_sum(num);
            sumy = quad_sum(sumy);
            const float inv_sum = __builtin_amdgcn_rcpf(sum);
            float p = (num * two_m64) * inv_sum;
            if (p < 1.401298464324817e-45f) p = 0.f;
            const float q_ = guard_finite(gv * __builtin_amdgcn_rcpf(p));
            const float qs = q_ * inv_sum;
            const float qp = qs * p;
            const float qs64 = qs * two_m64;
            float d[NP];
            float dxp = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const float tt = ef[i] * qs64;
                const float z = (xval - bk.loc[i]) * bk.isc[i];       // (recomputed: nine registers less across the sums)
                d[i] = fmaf(-e[i], qp, tt);
                acc_loc[i] = fmaf(tt, z, acc_loc[i]);
                acc_ls[i] = fmaf(tt, fmaf(z * z, PFPN_2LN2, -1.f), acc_ls[i]);
                dxp = fmaf(tt * z, bk.isc[i] * PFPN_2LN2, dxp);
            }
            // -- sample part (k_rsample_bwd_quad's arithmetic).  tanh(p_k) from libm at x = p_k (the forward's value); the
            // other particles' candidates through v_exp / v_rcp (|error| ~1e-7 against gaps of >= 0.05); scale_k * eps_k
            // from the lane that owns particle k.  (eps is read HERE, not with the logits: nine registers fewer live across
            // the log-prob part, which otherwise spills)
            const float tk = tanhf(xval);
            const float gds = -quad_sum(dxp) + gv * 2.f * tk;         // d L / d p_k through the log-prob
            const float inv = __builtin_amdgcn_rcpf(sumy);
            const float one_m = 1.f - tk * tk;
            const float cs = gds / fmaxf(1e-6f, one_m);
            float th[NP];
            float sek = 0.f, dot = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int n = g + SA_G * i;
                const float se = n < N ? bk.sc[i] * er[n] : 0.f;      // scale * eps
                if (n == k) sek = se;
                const float gap = sa_fast_tanh(bk.loc[i] + se) - tk;
                th[i] = gap * gda + gap * cs;                         // gw_n
                y[i] *= inv;                                          // y <- w_n
                dot += th[i] * y[i];
            }
            sek = quad_sum(sek);
            dot = quad_sum(dot);
            const float gp = gda * one_m + gds;                       // d L / d p_k
            const float gs = gp * sek;                                // d L / d logstd_k
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int n = g + SA_G * i;
                if (n < N) {
                    const float dt = d[i] + y[i] * (th[i] - dot);
                    row[n] = dt;
                    acc_b[i] += dt;
                    // (the sample's own d / d loc_k joins the log-prob's sum pre-divided by the factor that sum is finished with)
                    if (n == k) { acc_loc[i] = fmaf(gp, bk.sc[i] * (1.0f / PFPN_SQRT_2LN2), acc_loc[i]); acc_ls[i] += gs; }
                }
            }
        }
        __syncthreads();
        if (rows > 0) sa_stage_out(dlogits, ll, (size_t)r0 * AN, rows * AN, tid, nthr);
    }
    __syncthreads();
    // fold the Rb quads of every dim: planes [Rb][3 * AN] over the staging spans, then the slab in quad order
    if (rq_ < Rb) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int n = g + SA_G * i;
            if (n < N) {
                float* pl = sa_lds + (size_t)rq_ * 3 * AN + a * N + n;
                pl[0] = acc_loc[i] * (bk.isc[i] * PFPN_2LN2);
                pl[AN] = acc_ls[i];
                pl[2 * AN] = acc_b[i];
            }
        }
    }
    __syncthreads();
    float* slab = slabs + (size_t)blockIdx.x * 3 * AN;
    for (int idx = tid; idx < 3 * AN; idx += nthr) {
        float s = sa_lds[idx];
        for (int r = 1; r < Rb; ++r) s += sa_lds[(size_t)r * 3 * AN + idx];
        slab[idx] = s;
    }
}

// rows per block: whole rows, <= 144 quads (576 threads at A = 36), three staged tensors within 64 KiB of LDS
static inline int sa_rows_per_block(int A, int N) {
    if (A < 1 || N < 1 || N > 64 || A > 256) return 0;
    int rb = 144 / A;
    if (rb < 1) rb = 1;
    if (rb > 8) rb = 8;
    const int cap = (SA_MAX_LDS_FLOATS - 16) / (3 * A * N);
    if (rb > cap) rb = cap;
    return rb;                                            // 0: unsupported shape (the caller keeps the piece-wise launches)
}
// The launch is latency-bound (one block = one pass: stage, compute, write): small batches take FEWER rows per block so
// that the blocks spread over more CUs and a SIMD hosts one wave instead of three (backward at 256 rows: 10.8 -> 8.0 us;
// forward over 512 rows: 7.9 -> 6.7 us; forward over 2048 rows: 11.2 us at 4 rows per block, 10.9 at 2, 11.8 at 1).
// Backward: at most 256 blocks = slabs; beyond Rb_max * 256 rows a block walks `it` row groups.
static inline void sa_bwd_geom(int B, int A, int N, int* Rb, int* it) {
    const int rbmax = sa_rows_per_block(A, N);
    int rb = (B + 255) / 256;
    if (rb > rbmax) rb = rbmax;
    if (rb < 1) rb = rbmax > 0 ? 1 : 0;
    *Rb = rb;
    *it = rb > 0 ? (B + rb * 256 - 1) / (rb * 256) : 0;
    if (*it < 1) *it = 1;
}
static inline int sa_fwd_rows(int R, int A, int N) {
    const int rbmax = sa_rows_per_block(A, N);
    int rb = (R + 1023) / 1024;
    if (rb > rbmax) rb = rbmax;
    if (rb < 1) rb = rbmax > 0 ? 1 : 0;
    return rb;
}

}  // namespace pfpn

using namespace pfpn;

extern "C" {

int pfpn_sac_actor_supported(int A, int N) { return sa_rows_per_block(A, N) > 0 ? 1 : 0; }

int pfpn_sac_actor_fwd(const float* logits, const float* loc, const float* logstd, const float* gumbel,
                       const float* eps, int R, int A, int N, int32_t* k, float* a_out, float* s_out, float* logp,
                       const float* x, const float* a_hist, int B, int S, float* xq_both, float* xq_next,
                       void* stream) {
    PFPN_BEGIN();
    if (!logits || !loc || !logstd || !gumbel || !eps || !k || !a_out || !s_out || !logp || R < 0) return PFPN_ERR_ARG;
    if (xq_both && (!x || !a_hist || !xq_next || B < 0 || B > R || S < 1)) return PFPN_ERR_ARG;
    const int Rb = sa_fwd_rows(R, A, N);
    if (Rb < 1) return PFPN_ERR_ARG;
    if (R == 0) return PFPN_OK;
    const int threads = (Rb * A * SA_G + 63) & ~63;
    const size_t lds = ((size_t)2 * ((Rb * A * N + 3) & ~3) + Rb * A) * sizeof(float);
    const dim3 grid((R + Rb - 1) / Rb), block(threads);
#define PFPN_SA_FWD(NP_, MT_)                                                                                        \
    hipLaunchKernelGGL((k_sac_actor_fwd<NP_, MT_>), grid, block, lds, (hipStream_t)stream, logits, loc, logstd, gumbel, \
                       eps, R, A, N, Rb, (int*)k, a_out, s_out, logp, x, a_hist, B, S, xq_both, xq_next)
    if (N <= 36 && threads <= 640) PFPN_SA_FWD(9, 640);
    else if (N <= 36) PFPN_SA_FWD(9, 1024);
    else if (threads <= 640) PFPN_SA_FWD(16, 640);
    else PFPN_SA_FWD(16, 1024);
#undef PFPN_SA_FWD
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_sac_actor_bwd_slabs(int B, int A, int N, int* nslab, int* stride) {
    int Rb = 0, it = 0;
    sa_bwd_geom(B, A, N, &Rb, &it);
    if (!nslab || !stride || B < 1 || Rb < 1) return PFPN_ERR_ARG;
    *nslab = (B + Rb * it - 1) / (Rb * it);
    *stride = 3 * A * N;
    return PFPN_OK;
}

size_t pfpn_sac_actor_workspace_bytes(int B, int A, int N) {
    int n = 0, s = 0;
    if (pfpn_sac_actor_bwd_slabs(B, A, N, &n, &s) != PFPN_OK) return 0;
    return (size_t)n * s * sizeof(float);
}

int pfpn_sac_actor_bwd(const float* logits, const float* loc, const float* logstd, const float* gumbel,
                       const float* eps, const float* x, const float* dlogp, const float* da, int B, int A, int N,
                       float* dlogits, void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!logits || !loc || !logstd || !gumbel || !eps || !x || !dlogp || !da || !dlogits || B < 1) return PFPN_ERR_ARG;
    int Rb = 0, it = 0;
    sa_bwd_geom(B, A, N, &Rb, &it);
    if (Rb < 1) return PFPN_ERR_ARG;
    if (!workspace || workspace_bytes < pfpn_sac_actor_workspace_bytes(B, A, N)) return PFPN_ERR_WORKSPACE;
    const int threads = (Rb * A * SA_G + 63) & ~63;
    const size_t lds = (size_t)3 * ((Rb * A * N + 3) & ~3) * sizeof(float);
    const dim3 grid((B + Rb * it - 1) / (Rb * it)), block(threads);
#define PFPN_SA_BWD(NP_, MT_)                                                                                        \
    hipLaunchKernelGGL((k_sac_actor_bwd<NP_, MT_>), grid, block, lds, (hipStream_t)stream, logits, loc, logstd, gumbel, \
                       eps, x, dlogp, da, B, A, N, Rb, it, dlogits, (float*)workspace)
    if (N <= 36 && threads <= 640) PFPN_SA_BWD(9, 640);
    else if (N <= 36) PFPN_SA_BWD(9, 1024);
    else if (threads <= 640) PFPN_SA_BWD(16, 640);
    else PFPN_SA_BWD(16, 1024);
#undef PFPN_SA_BWD
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

}  // extern "C"
