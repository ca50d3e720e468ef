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
#include <stdlib.h>

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
    int want_db;             // also emit column sums of dlogits (fc_policy bias gradient) into the slab
};

constexpr int PACK_G = 4;                            // lanes per (row, action-dim) segment
constexpr int PACK_BLOCK = 1024;                     // launch bound; actual block = waves*64
constexpr int PACK_WSEG = 64 / PACK_G;               // 16 segments per wave

// ---------------------------------------------------------------------------------------
// packed path ("wave-sliced")
//
// A tile is Rt whole batch rows = Rt*A segments, laid out in LDS exactly as in HBM.  Wave w owns
// the 16 consecutive segments [16w, 16w+16) of the tile -- a contiguous 64*N-byte slice that the
// wave itself loads (float4, prefetched one tile ahead into registers), transforms in place and
// stores back: NO block barrier in steady state.  4 lanes share a segment (particles g, g+4, ...),
// their softmax/mixture reductions are two DPP shuffles.  The particle bank of a lane's (a, n) set
// is constant across tiles (same slot -> same action dim), so it lives in REGISTERS.
// The only cross-wave dependency is the row log-prob (a row's A segments span 2-4 waves): waves
// publish per-segment log p to LDS and bump a per-row arrival counter (monotonic, never reset);
// a wave waits only for the rows it touches.
// ---------------------------------------------------------------------------------------
// Prefetch registers as NAMED float4s (an indexed array that is conditionally written across loop
// iterations gets demoted to scratch by hipcc).
struct Pref { float4 v0, v1, v2, v3; };

template <int NV>
__device__ __forceinline__ void fetch_slice(Pref& pf, const float* __restrict__ logits, int row0, int nfl,
                                            int slice, int AN, int soff, int lane) {
    const int nv = max(0, min(nfl, slice)) >> 2;
    const int last = nv > 0 ? nv - 1 : 0;
    const float4* g4 = reinterpret_cast<const float4*>(logits + (size_t)row0 * AN + (size_t)soff);
    pf.v0 = g4[min(lane, last)];
    if (NV > 1) pf.v1 = g4[min(lane + 64, last)];
    if (NV > 2) pf.v2 = g4[min(lane + 128, last)];
    if (NV > 3) pf.v3 = g4[min(lane + 192, last)];
}

template <int NV>
__device__ __forceinline__ void put_slice(const Pref& pf, float* slice, int nv, int lane) {
    float4* s4 = reinterpret_cast<float4*>(slice);
    if (lane < nv) s4[lane] = pf.v0;
    if (NV > 1 && lane + 64 < nv) s4[lane + 64] = pf.v1;
    if (NV > 2 && lane + 128 < nv) s4[lane + 128] = pf.v2;
    if (NV > 3 && lane + 192 < nv) s4[lane + 192] = pf.v3;
}

// Diagnostic phase stamps (s_memtime) of block 0: enabled only when pfpn_debug_stamps() installed a buffer.
// Layout: [wave][slot], slot 0 = count; never read by the kernel itself.
__device__ unsigned long long* g_stamps = nullptr;
#ifdef PFPN_ENABLE_STAMPS
#define PFPN_STAMP(id_)                                                                              \
    do {                                                                                             \
        if (stamps && blockIdx.x == 0 && lane == 0) stamps[wave * 64 + (id_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PFPN_STAMP(id_) do { } while (0)      /* build with -DPFPN_ENABLE_STAMPS for tools/stamp_probe.py */
#endif

// MAXT: launch bound.  640 threads (10 waves -> 3 waves/SIMD -> 168 VGPRs) covers the 9-wave tile of
// A = 36; at the generic bound of 1024 (128 VGPRs) the PPO instance spills its accumulators.
template <int N, int MODE, int MAXT>
__global__ __launch_bounds__(MAXT) void k_mix_packed(
    const float* __restrict__ logits, const float* __restrict__ x,
    const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ dlogp_in,      // MODE_BWD: [B]
    int B, int A, int R, int tanh_mode, int flags,       // flags: bit 0 float4 slice path, bit 1 one-pass tail
    float* __restrict__ logp_out,            // [B]   (FWD, PPO)
    float* __restrict__ pmix_out,            // [B,A] optional
    float* __restrict__ dlogits,             // [B,A,N] (BWD, PPO)
    float* __restrict__ dx_out,              // [B,A] optional (BWD)
    float* __restrict__ partial,             // [grid][2*A*N + 4] (BWD, PPO)
    PpoArgs ppo)
{
    constexpr int NP = (N + PACK_G - 1) / PACK_G;         // particles per lane (9 for N=35)
    constexpr int SLICE = PACK_WSEG * N;                  // floats per wave slice (16 segments)
    constexpr int NV = (SLICE / 4 + 63) / 64;             // float4 per lane per slice (3 for N=35)
    extern __shared__ float4 smem4[];
    const int planes = (flags >> 1) & 3;                  // tile-sized LDS planes: 1, 2 (double buffer), 3 (+ one-pass tail)
    const int AN = A * N;
    const int nseg = R * A;                               // segments per full tile
    const int nwaves = blockDim.x >> 6;
    const int TS = nwaves * SLICE;                        // floats per plane (>= R*AN)
    float* tile = reinterpret_cast<float*>(smem4);        // [planes][TS]
    float* segbuf = tile + planes * TS;                   // [2][nseg]  per-segment log p (double buffered)
    int* cnt = reinterpret_cast<int*>(segbuf + 2 * nseg); // [R] row arrival counters
    float* lossbuf = reinterpret_cast<float*>(cnt + R);   // [2*nwaves]
    const int dbuf = planes >= 2 ? 1 : 0;

    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int q = lane >> 2, g = lane & 3;
    const int sidx = wave * PACK_WSEG + q;                // segment slot in the tile
    const bool seg_lane = sidx < nseg;
    const int r = seg_lane ? sidx / A : 0;
    const int a = seg_lane ? sidx - r * A : 0;
    float* slice0 = tile + wave * SLICE;
    float* seg0 = slice0 + q * N + g;                     // == tile + sidx*N + g
    const bool last_ok = (g + PACK_G * (NP - 1)) < N;     // does my NP-th particle exist?
    // rows this wave touches
    const int s_lo = wave * PACK_WSEG, s_hi = min(nseg, s_lo + PACK_WSEG) - 1;     // inclusive
    const int r_lo = s_lo / A, r_hi = (s_hi >= s_lo) ? s_hi / A : r_lo - 1;

    const int s_lo_c = min(s_lo, nseg - 1), r_lo_c = min(r_lo, R - 1);   // clamped copies for the prefetch addresses
    for (int i = t; i < R; i += blockDim.x) cnt[i] = 0;
#ifdef PFPN_ENABLE_STAMPS
    unsigned long long* stamps = g_stamps;
#endif
    PFPN_STAMP(0);
    const int ntiles = (B + R - 1) / R;
    static_assert(NV <= 4, "slice prefetch holds at most 4 float4 per lane");
    // ---- one-tile-ahead register prefetch of EVERYTHING a tile reads from HBM: my logits slice, my
    //      segment's action value, the row scalars of the rows whose loss head my 16-lane group runs
    //      (PPO) or my row's incoming gradient (BWD).  The only vmcnt wait of the steady state is the
    //      hand-off at the END of a tile, a full tile after the loads were issued.
    Pref pf;
    pf.v0 = pf.v1 = pf.v2 = pf.v3 = make_float4(0.f, 0.f, 0.f, 0.f);
    float xn = 0.f, n_adv = 0.f, n_lpo = 0.f, n_val = 0.f, n_vold = 0.f, n_g = 0.f;
#define PFPN_FETCH_GROUP(tid_)                                                                          \
    do {                                                                                                \
        /* every load is unconditional (indices clamped into the tile): a conditional load would make   \
           the compiler wait for ALL outstanding loads -- the slice just requested -- at the join */    \
        const int row0_ = (tid_) * R;                                                                   \
        const int rows_ = min(R, B - row0_);                                                            \
        const int nfl_f = rows_ * AN - s_lo_c * N;      /* <= 0: my slice lies beyond a ragged last tile */ \
        fetch_slice<NV>(pf, logits, row0_, nfl_f, SLICE, AN, nfl_f > 0 ? s_lo_c * N : 0, lane);        \
        xn = x[(size_t)row0_ * A + min(sidx, rows_ * A - 1)];                                           \
        if (MODE == MODE_BWD) n_g = dlogp_in[row0_ + min(r, rows_ - 1)];                                \
        if (MODE == MODE_PPO) {                                                                         \
            /* 16-lane group k <-> k-th row this wave touches */                                        \
            const int b_ = row0_ + min(r_lo_c + (lane >> 4), rows_ - 1);                                \
            n_adv = ppo.adv[b_]; n_lpo = ppo.logp_old[b_]; n_val = ppo.value[b_];                       \
            n_vold = ppo.value_old[b_];                                                                 \
        }                                                                                               \
    } while (0)
    // logits slice of tile `tid_` (in the prefetch registers) -> LDS plane `pl_`
#define PFPN_STAGE(tid_, pl_)                                                                           \
    do {                                                                                                \
        const int row0_ = (tid_) * R;                                                                   \
        const int nfl_ = max(0, min(min(R, B - row0_) * AN - s_lo * N, SLICE));                         \
        put_slice<NV>(pf, slice0 + (pl_) * TS, nfl_ >> 2, lane);                                        \
    } while (0)
    // first tile's HBM fetch goes out before anything else: its latency hides the bank set-up below
    // (tile ids are clamped to the last tile instead of guarded: a redundant re-load at the very end is
    // harmless, a conditional load is not -- see above)
    const int tlast = ntiles - 1;
    PFPN_FETCH_GROUP(min((int)blockIdx.x, tlast));

    // particle bank of my (a, n) set -> registers.  scale = exp(logstd), cterm = 0.5log(2pi)+log(scale)
    // as the reference graph computes them, pre-scaled to base 2:
    //   z' = (x - loc) * isc2,  phi = exp2(-z'^2 - ct2),  isc2 = sqrt(0.5 log2 e)/scale, ct2 = cterm*log2 e
    float b_loc[NP], b_isc[NP], b_ct[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int n = min(g + PACK_G * i, N - 1);
        const float sc = expf(logstd[a * N + n]);
        b_loc[i] = loc[a * N + n];
        b_isc[i] = (1.0f / sc) * PFPN_SQRT_HALF_LOG2E;
        // +64: e*phi is formed as ONE exp2 of (softmax exponent + Gaussian exponent + 64), so products
        // down to 2^-190 stay normal; the 2^-64 is taken back after the segment sum.
        b_ct[i] = 64.0f - (PFPN_HALF_LOG_2PI + logf(sc)) * PFPN_LOG2E;
    }
    const float two_m64 = 5.421010862427522e-20f;      // 2^-64
    float acc_loc[NP], acc_ls[NP], acc_db[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) { acc_loc[i] = 0.f; acc_ls[i] = 0.f; acc_db[i] = 0.f; }
    float pol_sum = 0.f, val_sum = 0.f;
    float adv_mean = 0.f, adv_rstd = 1.f;
    if (MODE == MODE_PPO) { adv_mean = ppo.adv_stats[0]; adv_rstd = ppo.adv_stats[1]; }

    PFPN_STAMP(1);
    __syncthreads();                                       // cnt zeroed
    PFPN_STAMP(2);

    // stage the first tile, take over its scalars, start the second tile's loads
    float xv = 0.f, rs_adv = 0.f, rs_lpo = 0.f, rs_val = 0.f, rs_vold = 0.f, rs_g = 0.f;
    PFPN_STAGE(min((int)blockIdx.x, tlast), 0);
    xv = xn; rs_adv = n_adv; rs_lpo = n_lpo; rs_val = n_val; rs_vold = n_vold; rs_g = n_g;
    PFPN_FETCH_GROUP(min((int)(blockIdx.x + gridDim.x), tlast));
    int it = 0, cur = 0;
    for (int tileid = blockIdx.x; tileid < ntiles; tileid += gridDim.x, ++it) {
        const int row0 = tileid * R;
        const int rows = min(R, B - row0);
        const int nact = rows * A;                         // active segments in this tile
        const int nfl = max(0, min(nact * N - s_lo * N, SLICE));   // valid floats in my slice
        float* sb = segbuf + (it & 1) * nseg;
        float* slice = slice0 + cur * TS;
        float* seg = seg0 + cur * TS;
        const bool active = seg_lane && sidx < nact;
        if (it < 4) PFPN_STAMP(3 + it * 6);

        // ---- forward on my quarter of the segment (wave-private LDS slice: no barrier) -------
        float ephi[NP];
        float inv_sum = 0.f, p = 0.f;
        if (active) {
            float mx = seg[0];
#pragma unroll
            for (int i = 1; i < NP; ++i)
                if (i < NP - 1 || last_ok) mx = fmaxf(mx, seg[PACK_G * i]);
            mx = quad_max(mx);
            const float mxl = mx * PFPN_LOG2E;
            float sum = 0.f, num = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                ephi[i] = 0.f;
                if (i < NP - 1 || last_ok) {
                    // raw v_exp_f32 twice: softmax numerator, and numerator * Normal.prob * 2^64 in one go
                    const float le = fmaf(seg[PACK_G * i], PFPN_LOG2E, -mxl);
                    const float zq = (xv - b_loc[i]) * b_isc[i];
                    const float en = __builtin_amdgcn_exp2f(le);
                    ephi[i] = __builtin_amdgcn_exp2f(le + fmaf(-zq, zq, b_ct[i]));
                    if (MODE != MODE_FWD) seg[PACK_G * i] = en;
                    sum += en;
                    num += ephi[i];
                }
            }
            sum = quad_sum(sum);
            num = quad_sum(num);
            inv_sum = __builtin_amdgcn_rcpf(sum);
            p = (num * two_m64) * inv_sum;
            if (p < 1.401298464324817e-45f) p = 0.f;       // below the smallest fp32 denormal: the reference's p is 0
            if (g == 0) {
                float lp = __builtin_amdgcn_logf(p) * 0.6931471805599453f;      // v_log_f32 is log2
                if (tanh_mode) lp -= tanh_log_det(xv);
                if (pmix_out) pmix_out[(size_t)row0 * A + sidx] = p;
                sb[sidx] = lp;
            }
        }

        if (it < 4) PFPN_STAMP(4 + it * 6);
        // ---- row stage ------------------------------------------------------------------------
        float my_g = 0.f;                                   // d loss / d logp of my row
        if (MODE == MODE_BWD) {
            if (active) my_g = rs_g;
        } else {
            // publish: my segments of each touched row have arrived.  segbuf and cnt both live in LDS, and a
            // wave's LDS operations are executed in issue order by the one LDS unit of the CU, so the ds_write
            // of log p is visible to whoever observes the later ds_add: a compiler barrier is all that is needed.
            // (A workgroup-scope fence would also drain vmcnt, i.e. wait for the NEXT tile's prefetch loads.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int rr_l = r_lo + lane;
            if (rr_l <= r_hi && rr_l < rows) {
                const int n_r = min(min((rr_l + 1) * A, s_hi + 1), nact) - max(rr_l * A, s_lo);
                if (n_r > 0) atomicAdd(&cnt[rr_l], n_r);
                const int target = A * (it + 1);
                while (__hip_atomic_load(&cnt[rr_l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
                    __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            if (it < 4) PFPN_STAMP(5 + it * 6);
            // Row stage, up to 4 rows at once: 16-lane group k sums row base+k's A segment log-probs and runs
            // its loss head; every wave touching a row forms the identical sum (same lanes, same order).
            {
                const int base = r_lo;                                 // a wave touches <= 4 rows (A >= 4, see plan_packed)
                const int kq = lane >> 4, jl = lane & 15;
                const int rk = base + kq;
                const bool rk_ok = rk <= r_hi && rk < rows;
                float sacc = 0.f;
                if (rk_ok)
                    for (int i = jl; i < A; i += 16) sacc += sb[rk * A + i];
                sacc = quad_sum(sacc);
                sacc += dpp_shift<0x124, 0xf>(sacc, 0.f);             // row_ror:4
                sacc += dpp_shift<0x128, 0xf>(sacc, 0.f);             // row_ror:8
                // the rotation steps associate differently per lane: every lane takes its group's lane 0
                const float t0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sacc), 0));
                const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sacc), 16));
                const float t2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sacc), 32));
                const float t3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sacc), 48));
                const float s = kq == 0 ? t0 : kq == 1 ? t1 : kq == 2 ? t2 : t3;
                const int b = row0 + rk;
                const bool owner = rk_ok && jl == 0 && (rk * A) / PACK_WSEG == wave;   // wave holding the row's first segment
                float gval = 0.f;
                if (MODE == MODE_PPO) {
                    const float adv = rs_adv, lpo = rs_lpo, val_b = rs_val, vold_b = rs_vold;
                const float nadv = (adv - adv_mean) * adv_rstd;
                    const float ratio = expf(s - lpo);
                    const float sur = ratio * nadv;
                    const float cr = fmaxf(fminf(ratio, ppo.clip_hi), ppo.clip_lo);
                    const float csur = cr * nadv;
                    // grad of -mean(min(sur, csur)): tf.minimum routes to `sur` where sur <= csur,
                    // clip_by_value passes where lo <= ratio <= hi.
                    const bool pass = (sur <= csur) || (ratio >= ppo.clip_lo && ratio <= ppo.clip_hi);
                    gval = pass ? (-ppo.inv_B * nadv) * ratio : 0.f;
                    if (owner) {
                        pol_sum += fminf(sur, csur);
                        const float dvv = val_b - (adv + vold_b);
                        val_sum += dvv * dvv;
                        ppo.dvalue[b] = 2.f * dvv * ppo.value_coef * ppo.inv_B;
                        if (ppo.ratio_out) ppo.ratio_out[b] = ratio;
                    }
                }
                if (owner) logp_out[b] = s;
                const float g0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gval), 0));
                const float g1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gval), 16));
                const float g2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gval), 32));
                const float g3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gval), 48));
                const int rel = r - base;
                if (rel >= 0 && rel < 4) my_g = rel == 0 ? g0 : rel == 1 ? g1 : rel == 2 ? g2 : g3;
            }
        }
        if (it < 4) PFPN_STAMP(6 + it * 6);

        // ---- backward on my quarter (dlogits overwrite the LDS slot) ---------------------------
        if (MODE != MODE_FWD && active) {
            const float q_ = guard_finite(my_g * __builtin_amdgcn_rcpf(p));
            const float qs = q_ * inv_sum;
            const float qp = qs * p;
            const float qs64 = qs * two_m64;
            float dxa = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (i < NP - 1 || last_ok) {
                    const float tt = qs64 * ephi[i];                        // q * w_n * phi_n
                    const float dlg = fmaf(-seg[PACK_G * i], qp, tt);       // q * w_n * (phi_n - p)
                    seg[PACK_G * i] = dlg;
                    if (MODE == MODE_PPO) acc_db[i] += dlg;                 // fc_policy bias gradient
                    const float zq = (xv - b_loc[i]) * b_isc[i];            // z / c
                    acc_loc[i] = fmaf(tt, zq, acc_loc[i]);                  // scaled by isc*c in the tail
                    acc_ls[i] = fmaf(tt, fmaf(zq * zq, PFPN_2LN2, -1.f), acc_ls[i]);   // tt*(z^2 - 1)
                    if (dx_out) dxa = fmaf(tt * zq, b_isc[i] * PFPN_2LN2, dxa);        // isc*c = isc2*2ln2
                }
            }
            if (dx_out) {
                dxa = quad_sum(dxa);
                if (g == 0) {
                    float d = -dxa;
                    if (tanh_mode) d += my_g * 2.f * tanhf(xv);
                    dx_out[(size_t)row0 * A + sidx] = d;
                }
            }
        }
        if (it < 4) PFPN_STAMP(7 + it * 6);
        // ---- hand-off: next tile registers -> the other LDS plane (the one vmcnt wait of the loop: those
        //      loads are a tile old), its scalars -> live registers, the tile after that -> prefetch;
        //      then my slice of dlogits -> HBM (same wave wrote it: in-order LDS, no barrier).  The stores
        //      drain behind the next tile's compute.
#define PFPN_STORE_SLICE()                                                                               \
    do {                                                                                                \
        float* gdst_ = dlogits + (size_t)row0 * AN + (size_t)s_lo * N;                                  \
        const int nv_ = nfl >> 2;                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < NV; ++j_) {                                             \
            const int i_ = lane + 64 * j_;                                                              \
            if (i_ < nv_) reinterpret_cast<float4*>(gdst_)[i_] = reinterpret_cast<const float4*>(slice)[i_]; \
        }                                                                                               \
    } while (0)
        if (MODE != MODE_FWD && !dbuf) PFPN_STORE_SLICE(); // single plane: it must be read out first
        const int nxt = tileid + (int)gridDim.x;
        PFPN_STAGE(min(nxt, tlast), cur ^ dbuf);
        if (it < 4) PFPN_STAMP(8 + it * 6);
        xv = xn; rs_adv = n_adv; rs_lpo = n_lpo; rs_val = n_val; rs_vold = n_vold; rs_g = n_g;
        PFPN_FETCH_GROUP(min(nxt + (int)gridDim.x, tlast));
        if (MODE != MODE_FWD && dbuf) PFPN_STORE_SLICE();
        cur ^= dbuf;
    }
#undef PFPN_STORE_SLICE
#undef PFPN_STAGE
#undef PFPN_FETCH_GROUP

    if (MODE == MODE_FWD) return;
    PFPN_STAMP(40);
    // ---- block partials: dloc / dlogstd summed over the R segments that share an action dim,
    //      in row order; loss sums in wave order.  Slab layout [nq*AN + 4] per block, nq = 2 (+1 with
    //      the dlogits column sums of the PPO mode).
    const int nq = (MODE == MODE_PPO && ppo.want_db) ? 3 : 2;
    float* slab = partial + (size_t)blockIdx.x * (nq * AN + 4);
    float* seg = seg0;                                     // plane 0
    if (planes >= 3) {
        // three tile-sized planes: all gradient sums staged in one pass
        float* lsb = tile + TS;
        __syncthreads();
        if (seg_lane) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (i < NP - 1 || last_ok) {
                    seg[PACK_G * i] = acc_loc[i] * (b_isc[i] * PFPN_2LN2);       // dloc *= c/scale
                    lsb[sidx * N + g + PACK_G * i] = acc_ls[i];
                    if (nq == 3) lsb[TS + sidx * N + g + PACK_G * i] = acc_db[i];
                }
        }
        __syncthreads();
        for (int idx = t; idx < nq * AN; idx += blockDim.x) {
            const int which = idx >= 2 * AN ? 2 : idx >= AN ? 1 : 0;
            const int i = idx - which * AN;
            const float* src = which == 0 ? tile : lsb + (which - 1) * TS;
            float s = 0.f;
            int rr = 0;
            for (; rr + 4 <= R; rr += 4) {                  // four loads in flight, summed in row order
                const float v0 = src[rr * AN + i], v1 = src[(rr + 1) * AN + i];
                const float v2 = src[(rr + 2) * AN + i], v3 = src[(rr + 3) * AN + i];
                s = (((s + v0) + v1) + v2) + v3;
            }
            for (; rr < R; ++rr) s += src[rr * AN + i];
            slab[idx] = s;
        }
    } else {
    for (int which = 0; which < nq; ++which) {
        __syncthreads();
        if (seg_lane) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (i < NP - 1 || last_ok)
                    seg[PACK_G * i] = which == 0 ? acc_loc[i] * (b_isc[i] * PFPN_2LN2)   // dloc *= c/scale
                                    : which == 1 ? acc_ls[i] : acc_db[i];
        }
        __syncthreads();
        for (int i = t; i < AN; i += blockDim.x) {
            float s = 0.f;
            for (int rr = 0; rr < R; ++rr) s += tile[rr * AN + i];
            slab[which * AN + i] = s;
        }
    }
    }
    if (MODE == MODE_PPO) {
        pol_sum = wave_sum_dpp(pol_sum);                  // held by the 4 group-leading lanes
        val_sum = wave_sum_dpp(val_sum);
        if (lane == 0) { lossbuf[wave] = pol_sum; lossbuf[nwaves + wave] = val_sum; }
        __syncthreads();
        if (t == 0) {
            float ps = 0.f, vs = 0.f;
            for (int i = 0; i < nwaves; ++i) { ps += lossbuf[i]; vs += lossbuf[nwaves + i]; }
            slab[nq * AN + 0] = ps;
            slab[nq * AN + 1] = vs;
        }
    } else if (t == 0) {
        slab[nq * AN + 0] = 0.f;
        slab[nq * AN + 1] = 0.f;
    }
    PFPN_STAMP(41);
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

// ---------------------------------------------------------------------------------------
// wide path (N % 4 == 0, N >= 64; the 128 x 256 stress shape): one wavefront per segment, a lane
// owns M float4s (4 consecutive particles each): a segment is fetched as whole 1-KiB coalesced
// wave-loads straight into registers (no LDS).  Wave (a, c) keeps its action dim, so the particle
// bank and the dloc/dlogstd accumulators of its 4*M particles stay in registers while it walks rows
// b = c, c+C, ...; U rows are in flight per wave (U float4 loads per lane outstanding) to cover the
// HBM latency.  Reductions over a segment are wave shuffles.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <bool V> struct BoolTag { static constexpr bool value = V; };

template <int M, int MODE, bool TANH, bool FULLN>
__global__ __launch_bounds__(256) void k_mix_wide(
    const float* __restrict__ logits, const float* __restrict__ x,
    const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ dlogp_in,
    int B, int A, int N, int C,
    float* __restrict__ lpseg_out,           // [B,A] per-segment log p (FWD)
    float* __restrict__ pmix_out,            // [B,A] optional
    float* __restrict__ dlogits,             // BWD
    float* __restrict__ dx_out,              // BWD optional
    float* __restrict__ partial)             // [C][2*A*N]  (BWD)
{
    constexpr int U = 4;                                    // rows in flight per wave (reduced together)
    const int lane = threadIdx.x & 63;
    // wave-uniform indices through readfirstlane: row/segment address arithmetic goes to the scalar unit
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wave >= A * C) return;
    const int a = wave % A;                                 // neighbouring waves -> neighbouring dims
    const int c = wave / A;
    const int AN = A * N;
    const int nq = N >> 2;                                  // float4s per segment
    // bank in registers: loc, isc2 = sqrt(.5 log2 e)/scale, ct64 = 64 - (0.5log2pi + log scale)*log2 e.
    // e*phi = exp2(l*log2e - max*log2e - z'^2 + ct64) * 2^-64 in ONE v_exp_f32: the +64 keeps every
    // product that matters far above the flush-to-zero threshold of the raw instruction.
    bool ok[M];
    float4 b_loc[M], b_isc[M], b_ct[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int qi = lane + 64 * j;
        ok[j] = FULLN || qi < nq;                            // FULLN: N == 256*M, every lane owns particles
        const int o = a * N + 4 * (ok[j] ? qi : 0);
        const float4 lo = ld4(loc + o), ls = ld4(logstd + o);
        b_loc[j] = lo;
        float sc;
#define PFPN_BANK(c_)                                                                       \
    sc = expf(ls.c_);                                                                       \
    b_isc[j].c_ = (1.0f / sc) * PFPN_SQRT_HALF_LOG2E;                                        \
    b_ct[j].c_ = 64.f - (PFPN_HALF_LOG_2PI + logf(sc)) * PFPN_LOG2E;
        PFPN_BANK(x) PFPN_BANK(y) PFPN_BANK(z) PFPN_BANK(w)
#undef PFPN_BANK
    }
    float4 acc_loc[M], acc_ls[M];
#pragma unroll
    for (int j = 0; j < M; ++j) { acc_loc[j] = make_float4(0.f, 0.f, 0.f, 0.f); acc_ls[j] = acc_loc[j]; }
    const float two_m64 = 5.421010862427522e-20f;           // 2^-64

    // One group of U rows.  FULL = every row of the group exists: the body is then BRANCH-FREE straight-line
    // code (stores unconditional), so hipcc's waitcnt pass can count the stores younger than the prefetch
    // loads and emit s_waitcnt vmcnt(k) instead of vmcnt(0) -- otherwise every iteration would also wait for
    // its own stores to be acknowledged (~1 us exposed per iteration; CDNA4 vmcnt counts stores).
    auto group = [&](int b0, const float4 (&l)[U][M], const float (&xv)[U], const float (&gv)[U], auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        float mx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mx[u] = -INFINITY;
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (ok[j]) mx[u] = fmaxf(mx[u], fmaxf(fmaxf(l[u][j].x, l[u][j].y), fmaxf(l[u][j].z, l[u][j].w)));
        }
        wave_reduce4<true>(mx[0], mx[1], mx[2], mx[3], lane);
        float4 e[U][M], ef[U][M], zq[U][M];
        float sum[U], num[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float mxl = mx[u] * PFPN_LOG2E;
            sum[u] = 0.f; num[u] = 0.f;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                e[u][j] = ef[u][j] = zq[u][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok[j]) {
#define PFPN_EL(c_)                                                                              \
    {                                                                                            \
        const float le = fmaf(l[u][j].c_, PFPN_LOG2E, -mxl);                                     \
        zq[u][j].c_ = (xv[u] - b_loc[j].c_) * b_isc[j].c_;                                       \
        e[u][j].c_ = __builtin_amdgcn_exp2f(le);                                                 \
        ef[u][j].c_ = __builtin_amdgcn_exp2f(le + fmaf(-zq[u][j].c_, zq[u][j].c_, b_ct[j].c_));  \
    }
                    PFPN_EL(x) PFPN_EL(y) PFPN_EL(z) PFPN_EL(w)
#undef PFPN_EL
                    sum[u] += (e[u][j].x + e[u][j].y) + (e[u][j].z + e[u][j].w);
                    num[u] += (ef[u][j].x + ef[u][j].y) + (ef[u][j].z + ef[u][j].w);
                }
            }
        }
        wave_reduce4<false>(sum[0], sum[1], sum[2], sum[3], lane);
        wave_reduce4<false>(num[0], num[1], num[2], num[3], lane);
        float dxa[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u * C;
            const bool live = FULL || b < B;
            const float inv_sum = __builtin_amdgcn_rcpf(sum[u]);    // wave-uniform: one v_rcp, not an IEEE divide
            float p = (num[u] * two_m64) * inv_sum;
            if (p < 1.401298464324817e-45f) p = 0.f;         // below the smallest fp32 denormal: the reference's p is 0
            dxa[u] = 0.f;
            if (MODE == MODE_FWD) {
                float lp = __builtin_amdgcn_logf(p) * 0.6931471805599453f;     // v_log_f32 is log2
                if (TANH) lp -= tanh_log_det(xv[u]);
                // every lane stores the same value to the same address (one transaction, no branch)
                if (live) {
                    lpseg_out[(size_t)b * A + a] = lp;
                    if (pmix_out) pmix_out[(size_t)b * A + a] = p;
                }
            } else {
                const float q_ = live ? guard_finite(gv[u] * __builtin_amdgcn_rcpf(p)) : 0.f;
                const float qs = q_ * inv_sum;
                const float qp = qs * p;
                const float qs64 = qs * two_m64;
                float* drow = dlogits + (size_t)b * AN + (size_t)a * N;
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    if (ok[j]) {
                        float4 d;
#define PFPN_EL(c_)                                                                              \
    {                                                                                            \
        const float tt = qs64 * ef[u][j].c_;                                                     \
        const float z_ = zq[u][j].c_;                                                            \
        d.c_ = fmaf(-e[u][j].c_, qp, tt);                                                        \
        acc_loc[j].c_ = fmaf(tt, z_, acc_loc[j].c_);                                             \
        acc_ls[j].c_ = fmaf(tt, fmaf(z_ * z_, PFPN_2LN2, -1.f), acc_ls[j].c_);                   \
        if (dx_out) dxa[u] = fmaf(tt * z_, b_isc[j].c_ * PFPN_2LN2, dxa[u]);                     \
    }
                        PFPN_EL(x) PFPN_EL(y) PFPN_EL(z) PFPN_EL(w)
#undef PFPN_EL
                        if (live) *reinterpret_cast<float4*>(drow + 4 * (lane + 64 * j)) = d;
                    }
                }
            }
        }
        if (MODE == MODE_BWD && dx_out) {
            wave_reduce4<false>(dxa[0], dxa[1], dxa[2], dxa[3], lane);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = b0 + u * C;
                if (FULL || b < B) {
                    float dd = -dxa[u];
                    if (TANH) dd += gv[u] * 2.f * tanhf(xv[u]);
                    dx_out[(size_t)b * A + a] = dd;
                }
            }
        }
    };

    // Register double buffering: the loads of the NEXT group are issued before the current group is
    // processed, so a wave's own compute overlaps its HBM latency instead of adding to it.
    float4 ln[U][M];
    float xn[U], gn[U];
#define PFPN_WIDE_LOAD(b0_)                                                                        \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                \
        const int b_ = min((b0_) + u * C, B - 1);            /* clamped: loads unconditional */     \
        const float* row_ = logits + (size_t)b_ * AN + (size_t)a * N;                              \
        _Pragma("unroll") for (int j = 0; j < M; ++j) ln[u][j] = ld4(row_ + 4 * (ok[j] ? lane + 64 * j : 0)); \
        xn[u] = x[(size_t)b_ * A + a];                                                             \
        gn[u] = (MODE == MODE_BWD) ? dlogp_in[b_] : 0.f;                                           \
    }
    const int step = C * U;
    int b0 = c;
    PFPN_WIDE_LOAD(b0)
    for (; b0 + (U - 1) * C < B; b0 += step) {               // full groups: branch-free body
        float4 l[U][M];
        float xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < M; ++j) l[u][j] = ln[u][j];
            xv[u] = xn[u];
            gv[u] = gn[u];
        }
        PFPN_WIDE_LOAD(b0 + step)                            // clamped rows: harmless beyond the end
        group(b0, l, xv, gv, BoolTag<true>{});
    }
    if (b0 < B) group(b0, ln, xn, gn, BoolTag<false>{});     // ragged tail (1..U-1 live rows)
#undef PFPN_WIDE_LOAD
    if (MODE == MODE_BWD) {
        float* slab = partial + (size_t)c * 2 * AN + (size_t)a * N;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if (ok[j]) {
                float4 dl;
                dl.x = acc_loc[j].x * (b_isc[j].x * PFPN_2LN2); dl.y = acc_loc[j].y * (b_isc[j].y * PFPN_2LN2);
                dl.z = acc_loc[j].z * (b_isc[j].z * PFPN_2LN2); dl.w = acc_loc[j].w * (b_isc[j].w * PFPN_2LN2);
                *reinterpret_cast<float4*>(slab + 4 * (lane + 64 * j)) = dl;
                *reinterpret_cast<float4*>(slab + AN + 4 * (lane + 64 * j)) = acc_ls[j];
            }
        }
    }
}

// logp[b] = sum_a lpseg[b,a]: one wavefront per row, lanes read the row coalesced, fixed-order
// butterfly (deterministic).  (A serial per-thread loop here costs A dependent-load latencies.)
__global__ __launch_bounds__(256) void k_row_sum(const float* __restrict__ lpseg, int B, int A,
                                                 float* __restrict__ logp) {
    const int lane = threadIdx.x & 63;
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= B) return;
    float s = 0.f;
    for (int a = lane; a < A; a += 64) s += lpseg[(size_t)b * A + a];
    s = wave_sum(s);
    if (lane == 0) logp[b] = s;
}

// Fixed-order reduction of per-block slabs: out[i] = sum_k partial[k*stride + i], i < n.
// With `loss_out`, one extra block (the last) reduces the two loss columns behind the n data columns and
// emits loss_out = {loss, policy_loss, value_loss*coef}.  Grid: ceil(n/64) (+1) blocks of 1024.
__global__ __launch_bounds__(1024) void k_finalize(const float* __restrict__ partial, int nslab, int stride, int n,
                                                   float* __restrict__ out0, float* __restrict__ out1, int n0,
                                                   float* __restrict__ loss_out, float inv_B, float value_coef,
                                                   float* __restrict__ out2 = nullptr) {
    __shared__ float red[16][64];
    __shared__ float lsum[2];
    const int cl = threadIdx.x & 63;
    if (loss_out && blockIdx.x == gridDim.x - 1) {
        const float s = slab_reduce_1024<false>(partial, nslab, (size_t)stride, n + (cl & 1), cl < 2, red);
        if ((threadIdx.x >> 6) == 0 && cl < 2) lsum[cl] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float pl = -lsum[0] * inv_B, vl = lsum[1] * inv_B * value_coef;
            loss_out[0] = pl + vl;
            loss_out[1] = pl;
            loss_out[2] = vl;
        }
        return;
    }
    const int i = blockIdx.x * 64 + cl;
    const float s = slab_reduce_1024<false>(partial, nslab, (size_t)stride, i, i < n, red);
    if ((threadIdx.x >> 6) != 0 || i >= n) return;
    if (i < n0) out0[i] = s;
    else if (i < 2 * n0) out1[i - n0] = s;
    else out2[i - 2 * n0] = s;          // only reached with n = 3*n0
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
struct PackPlan { int R; int grid; int threads; size_t lds; int vec_ok; bool ok; };

static int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

static PackPlan plan_packed(int B, int A, int N, const void* logits, const void* dlogits) {
    PackPlan pl{};
    pl.ok = false;
    // A >= 4: the 16 consecutive segments of a wave then span at most 4 rows (one 16-lane group per row
    // in the row stage); narrower action spaces take the wave kernel
    if (N > 64 || N < 1 || A < 4 || B < 1) return pl;
    const int AN = A * N;
    // rows per tile: the smallest R0 with R0*A a multiple of 16 segments (full waves), repeated while the
    // block stays <= 12 waves (A=36: R=4, 9 waves, every lane busy)
    const int R0 = PACK_WSEG / gcd_i(A, PACK_WSEG);
    const int W0 = R0 * A / PACK_WSEG;
    if (W0 > 16) return pl;
    int k = 12 / W0;
    if (k < 1) k = 1;
    int R = R0 * k;
    // at <= 128 VGPRs a CU holds 16 waves: prefer ONE block that fills them (largest R with <= 16 waves)
    // over two half-size blocks that cannot be co-resident (measured on the 4096 x 36 x 35 step)
    static const int env_full = getenv("PFPN_MIX_FULL") ? atoi(getenv("PFPN_MIX_FULL")) : 0;
    if (env_full) {
        const int rmax = (16 * PACK_WSEG) / A;
        if (rmax >= 1 && (rmax * A + PACK_WSEG - 1) / PACK_WSEG <= 16 && rmax > R) R = rmax;
    }
    static const int env_rows = getenv("PFPN_MIX_ROWS") ? atoi(getenv("PFPN_MIX_ROWS")) : 0;   // tuning knob
    if (env_rows > 0 && (env_rows * A + PACK_WSEG - 1) / PACK_WSEG <= 16) R = env_rows;
    if (R > B) R = B;                                   // small batches: one partial tile
    const int waves = (R * A + PACK_WSEG - 1) / PACK_WSEG;
    pl.R = R;
    pl.threads = waves * 64;
    // LDS: tile-sized planes + small arrays.  1 plane is the minimum; 2 double-buffer the tile (the next
    // tile is staged while the current one is still being stored); 3 also give the backward tail room to
    // stage all three gradient sums in one pass.
    const size_t plane = (size_t)waves * PACK_WSEG * N * 4;
    const size_t small = (size_t)(2 * R * A + R + 2 * waves + 8) * 4;
    int planes = 1;
    if (2 * plane + small <= 64 * 1024) planes = 2;
    if (dlogits && 3 * plane + small <= 64 * 1024) planes = 3;
    pl.lds = planes * plane + small;
    pl.vec_ok = ((AN % 4 == 0) && ((uintptr_t)logits % 16 == 0) && (dlogits == nullptr || (uintptr_t)dlogits % 16 == 0)) ? 1 : 0;
    pl.vec_ok |= planes << 1;                           // kernel `flags`: bit 0 float4 path, bits 1-2 plane count
    const int ntiles = (B + R - 1) / R;
    // resident blocks per CU by wave slots (128 VGPRs: 4 waves/SIMD = 16 waves/CU) and LDS
    int per_cu = 16 / waves;
    const int by_lds = (int)((size_t)(150 * 1024) / (pl.lds ? pl.lds : 1));
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    static const int env_pcu = getenv("PFPN_MIX_PER_CU") ? atoi(getenv("PFPN_MIX_PER_CU")) : 0;   // tuning knob
    if (env_pcu > 0) per_cu = env_pcu;
    const int cap = 256 * per_cu;
    pl.grid = ntiles < cap ? ntiles : cap;
    // the packed kernel streams float4 slices only; unaligned shapes take the wave kernel
    pl.ok = pl.lds <= 64 * 1024 && waves <= 16 && (pl.vec_ok & 1);
    return pl;
}

template <int MODE>
static int launch_packed(int N, const PackPlan& pl, hipStream_t st, const float* logits, const float* x,
                         const float* loc, const float* logstd, const float* dlogp, int B, int A, int tanh_mode,
                         float* logp, float* pmix, float* dlogits, float* dx, float* partial, PpoArgs ppo) {
#define PFPN_LAUNCH_PACKED_T(NN, TT)                                                                        \
    hipLaunchKernelGGL((k_mix_packed<NN, MODE, TT>), dim3(pl.grid), dim3(pl.threads), pl.lds, st, logits, x, \
                       loc, logstd, dlogp, B, A, pl.R, tanh_mode, pl.vec_ok, logp, pmix, dlogits, dx,       \
                       partial, ppo)
#define PFPN_LAUNCH_PACKED(NN)                                                  \
    do {                                                                        \
        if (pl.threads <= 640) PFPN_LAUNCH_PACKED_T(NN, 640);                   \
        else PFPN_LAUNCH_PACKED_T(NN, PACK_BLOCK);                              \
    } while (0)
    switch (N) {
        case 35: PFPN_LAUNCH_PACKED(35); break;
        case 50: PFPN_LAUNCH_PACKED(50); break;
        case 16: PFPN_LAUNCH_PACKED(16); break;
        case 8: PFPN_LAUNCH_PACKED(8); break;
        default: return -1;
    }
#undef PFPN_LAUNCH_PACKED
#undef PFPN_LAUNCH_PACKED_T
    return 0;
}

static bool packed_has(int N) { return N == 35 || N == 50 || N == 16 || N == 8; }

// wide path: float4 per lane, up to 4 float4s per lane (N <= 1024)
static bool wide_ok(int A, int N, const void* logits, const void* dlogits) {
    return N >= 64 && N % 4 == 0 && N <= 1024 && ((uintptr_t)logits % 16 == 0) &&
           (dlogits == nullptr || (uintptr_t)dlogits % 16 == 0);
}
static int wide_c_env() { static const int v = getenv("PFPN_WIDE_C") ? atoi(getenv("PFPN_WIDE_C")) : 0; return v; }
// waves per action dim: ~16 waves per CU over the chip, at least 2 rows per wave, <= 64 slabs
static int wide_chunks(int B, int A, bool bwd = false) {
    // measured on the 8192 x 128 x 256 stress shape: ~16 waves/CU (fwd) and ~12 waves/CU (bwd, which also
    // streams 1 KiB stores per segment) keep about 8 MB of requests in flight chip-wide -- more in
    // flight only lengthens the HBM queues (tools/bw_probe.hip)
    int C = ((bwd ? 256 * 12 : 256 * 16) + A - 1) / A;
    if (C > 64) C = 64;
    if (wide_c_env() > 0) C = wide_c_env();
    if (C > (B + 3) / 4) C = (B + 3) / 4;      // at least U = 4 rows per wave
    if (C < 1) C = 1;
    return C;
}

}  // namespace pfpn

using namespace pfpn;

// =======================================================================================
// C ABI (declared in include/pfpn_hip.h)
// =======================================================================================
extern "C" {

// diagnostic: install (or clear with NULL) a device buffer of 16*64 uint64 for the phase stamps of block 0
int pfpn_debug_stamps(void* device_buffer) {
    unsigned long long* p = (unsigned long long*)device_buffer;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)) == hipSuccess ? PFPN_OK : PFPN_ERR_LAUNCH;
}

size_t pfpn_mixture_workspace_bytes(int B, int A, int N) {
    // slabs: packed path <= 1024 blocks x (2AN+4); wave/wide path C<=64 chunks x 2AN; + lpseg [B*A]
    const size_t AN = (size_t)A * N;
    const size_t slabs = (N <= 64 ? (size_t)1024 : (size_t)64) * (3 * AN + 4);
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
    if (wide_ok(A, N, logits, nullptr)) {
        const int C = wide_chunks(B, A);
        const int blocks = (A * C * 64 + 255) / 256;
        const int M = (N / 4 + 63) / 64;
#define PFPN_WIDE_FWD(MM, TT)                                                                                  \
    do {                                                                                                       \
        if (N == 256 * MM)                                                                                     \
            hipLaunchKernelGGL((k_mix_wide<MM, MODE_FWD, TT, true>), dim3(blocks), dim3(256), 0, st, logits, x, loc,   \
                               logstd, (const float*)nullptr, B, A, N, C, lpseg, pmix, (float*)nullptr,         \
                               (float*)nullptr, (float*)nullptr);                                               \
        else                                                                                                   \
            hipLaunchKernelGGL((k_mix_wide<MM, MODE_FWD, TT, false>), dim3(blocks), dim3(256), 0, st, logits, x, loc,  \
                               logstd, (const float*)nullptr, B, A, N, C, lpseg, pmix, (float*)nullptr,         \
                               (float*)nullptr, (float*)nullptr);                                               \
    } while (0)
        if (tanh_mode) { if (M == 1) PFPN_WIDE_FWD(1, true); else if (M == 2) PFPN_WIDE_FWD(2, true); else PFPN_WIDE_FWD(4, true); }
        else { if (M == 1) PFPN_WIDE_FWD(1, false); else if (M == 2) PFPN_WIDE_FWD(2, false); else PFPN_WIDE_FWD(4, false); }
#undef PFPN_WIDE_FWD
        PFPN_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_row_sum, dim3((B * 64 + 255) / 256), dim3(256), 0, st, lpseg, B, A, logp);
        PFPN_CHECK_LAUNCH();
        return PFPN_OK;
    }
    int C = (B < 64) ? B : 64;
    while ((size_t)A * C < 2048 && C < B) C = (C * 2 < B) ? C * 2 : B;
    const int waves = A * C;
    const int blocks = (waves * 64 + 255) / 256;
    hipLaunchKernelGGL((k_mix_wave<MODE_FWD>), dim3(blocks), dim3(256), 0, st, logits, x, loc, logstd,
                       (const float*)nullptr, B, A, N, C, tanh_mode, lpseg, pmix, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_row_sum, dim3((B * 64 + 255) / 256), dim3(256), 0, st, lpseg, B, A, logp);
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
    if (wide_ok(A, N, logits, dlogits)) {
        const int C = wide_chunks(B, A, true);
        if (workspace_bytes < (size_t)C * 2 * AN * sizeof(float)) return PFPN_ERR_WORKSPACE;
        const int blocks = (A * C * 64 + 255) / 256;
        const int M = (N / 4 + 63) / 64;
#define PFPN_WIDE_BWD(MM, TT)                                                                                  \
    do {                                                                                                       \
        if (N == 256 * MM)                                                                                     \
            hipLaunchKernelGGL((k_mix_wide<MM, MODE_BWD, TT, true>), dim3(blocks), dim3(256), 0, st, logits, x, loc,   \
                               logstd, dlogp, B, A, N, C, (float*)nullptr, (float*)nullptr, dlogits, dx, partial); \
        else                                                                                                   \
            hipLaunchKernelGGL((k_mix_wide<MM, MODE_BWD, TT, false>), dim3(blocks), dim3(256), 0, st, logits, x, loc,  \
                               logstd, dlogp, B, A, N, C, (float*)nullptr, (float*)nullptr, dlogits, dx, partial); \
    } while (0)
        if (tanh_mode) { if (M == 1) PFPN_WIDE_BWD(1, true); else if (M == 2) PFPN_WIDE_BWD(2, true); else PFPN_WIDE_BWD(4, true); }
        else { if (M == 1) PFPN_WIDE_BWD(1, false); else if (M == 2) PFPN_WIDE_BWD(2, false); else PFPN_WIDE_BWD(4, false); }
#undef PFPN_WIDE_BWD
        PFPN_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_finalize, dim3((2 * AN + 63) / 64), dim3(1024), 0, st, partial, C, 2 * AN, 2 * AN,
                           dloc, dlogstd, AN, (float*)nullptr, 0.f, 0.f);
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
                             float* dbias, float* loss3, void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!logits || !action || !loc || !logstd || !logp_old || !adv || !value || !value_old || !adv_stats ||
        !logp || !dlogits || !dloc || !dlogstd || !dvalue || !loss3 || B < 1 || A < 1 || N < 1)
        return PFPN_ERR_ARG;
    if (!workspace) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int AN = A * N;
    PackPlan pl = plan_packed(B, A, N, logits, dlogits);
    if (!(pl.ok && packed_has(N))) return PFPN_ERR_ARG;   // caller falls back to fwd + ppo_loss + bwd
    const int nq = dbias ? 3 : 2;
    if (workspace_bytes < (size_t)pl.grid * (nq * AN + 4) * sizeof(float)) return PFPN_ERR_WORKSPACE;
    PpoArgs pa{};
    pa.logp_old = logp_old; pa.adv = adv; pa.value = value; pa.value_old = value_old; pa.adv_stats = adv_stats;
    pa.inv_B = 1.0f / (float)B; pa.clip_lo = 1.0f - clip_eps; pa.clip_hi = 1.0f + clip_eps;
    pa.value_coef = value_coef; pa.dvalue = dvalue; pa.ratio_out = nullptr; pa.want_db = dbias ? 1 : 0;
    float* partial = (float*)workspace;
    launch_packed<MODE_PPO>(N, pl, st, logits, action, loc, logstd, nullptr, B, A, 0, logp, nullptr, dlogits,
                            nullptr, partial, pa);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_finalize, dim3((nq * AN + 63) / 64 + 1), dim3(1024), 0, st, partial, pl.grid, nq * AN + 4,
                       nq * AN, dloc, dlogstd, AN, loss3, pa.inv_B, value_coef, dbias);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

}  // extern "C"
