// Small-row dense layers for gfx950 (MI355X): ONE launch per layer and direction for batches of <= ~1024 rows.
//
// Replaces, for the short learner steps (the 512-row shard of a data-parallel step, the reference's own 256-row global
// step, a worker's 32-row minibatch, the SAC / TD3 replay batches) and the rollout's policy call, the sequence
//   library GEMM -> bias/ReLU6 pass -> (critic head pass)            forward   (reference networks/ops.py:82-118)
//   library dW GEMM, library dX GEMM -> Relu6Grad/BiasAddGrad pass   backward  (tf.gradients of the same lines)
// by a JOB TABLE kernel: a launch carries up to PFPN_DENSE_MAX_JOBS jobs, each a tiled fp32-MFMA product
// (v_mfma_f32_32x32x2_f32: exact fp32, the reference's arithmetic) with its epilogue fused --
//   NN  Y  = act(X W + b) (+ per-32-column partial dot products with a scalar head's weights: critic fc3 / twin Q)
//   NT  dX = (dY W^T) * [0 < H < 6]                      (MatMul_grad input side + Relu6Grad of the layer below)
//   TN  dW = X^T dY, db = column sums of dY               (MatMul_grad weight side + BiasAddGrad; the batch is the
//                                                          reduction, it never leaves the block: no partial slabs)
// or one of three tiny helpers (head partials -> value, the rank-1 backward of a scalar head, its weight gradient).
// A block picks its job from its index, so the dW and dX products of a layer -- different operand layouts, different
// tile shapes -- share one launch.  At these sizes a launch costs >= ~4.5 us from dispatch to completion whatever its
// rows, so the step time is the launch count: 16 -> 10 launches per DPPO learner step.
//
// Tiles.  256 threads = 4 waves per block, every wave owns ONE 32x32 accumulator (16 VGPRs; a single dependent MFMA
// chain issues back to back: 64 cycles issue = 64 cycles latency).  Three arrangements of the four waves:
//   cfg 0: 2x2 waves, tile 64x64, k-tile 32      cfg 1: 1x2 waves x 2 k-halves, tile 32x64, k-tile 64
//   cfg 2: 1x1 wave x 4 k-quarters, tile 32x32, k-tile 128
// so a 512-row layer still spreads over all 256 CUs (small tiles) while each block keeps four SIMDs busy (the K range
// of a tile is split over the block's waves and summed through LDS in wave order: bitwise reproducible).
// Staging: global -> registers (float4; scalar loads for operands whose rows are not 16-byte aligned, e.g. the
// 197-wide state) -> LDS, double buffered, the loads of k-tile t+2 in flight during the MFMAs of tile t; every load is
// branch-free (clamped address + select) so hipcc keeps counted vmcnt waits.  LDS images and fragment maps are those
// of gemm_kernels.hip: a K-contiguous operand as [rows][k-tile + 4] read by ds_read_b128 (conflict-free: 16 rows of a
// lane group land on 16 disjoint 16-B slots), a tile-contiguous operand as [k][tile] read by ds_read_b32; MFMA step q of
// chunk c multiplies k = 8c + 4h + q on lane half h for BOTH operands.
#include <cstdlib>
#include "pfpn_common.h"
#include "pfpn_hip.h"

namespace pfpn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RD_THREADS = 256;      // 4 waves per block (8-wave blocks with twice the in-block K split measured slower)

struct DenseJobDev {
    PfpnDenseJob j;
    int block0, tiles_m, tiles_n, avec, xcd_remap;
};
struct DenseTable {
    int n;
    // the launch also leaves the partial sums of the global gradient norm (one per block, fixed order: block index) and
    // rotates the Adam powers, like pfpn_reduce_jobs_norm -- the step's last dense launch then stands in for the separate
    // reduction launch
    float* sq_partial;
    float* beta_pow;
    int n_groups;
    float b1, b2;
    DenseJobDev job[PFPN_DENSE_MAX_JOBS];
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 sel4(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// one operand of a product.  KC: memory rows = tile rows, k along the row.  TC: memory rows = k, tile dim along the row.
template <bool KC, int T, int BKT, bool VEC>
struct RowOperand {
    static constexpr int LDK = BKT + 4;
    static constexpr int FLOATS = KC ? T * LDK : BKT * T;
    static constexpr int NL = T * BKT / (4 * RD_THREADS);     // float4 loads per thread and k-tile
    static constexpr int PER_ROW = KC ? BKT / 4 : T / 4;      // float4s per memory row of the tile
    const float* base;          // operand base (always readable: the clamp target)
    const float* p[NL];         // my float4s of the current k-tile
    bool tok[NL];               // tile index (row / column group) in range
    int tlim[NL];               // VEC = false, TC: valid elements of my float4 in the tile direction
    int kk[NL];                 // KC: my k offset inside the k-tile; TC: my k row inside the k-tile
    int sts[NL];
    long long step;             // pointer advance per k-tile
    int K;

    __device__ __forceinline__ void init(const float* P, int ld, int t0, int n_t, int Kred, int t) {
        base = P; K = Kred;
        step = KC ? (long long)BKT : (long long)BKT * ld;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int idx = t + RD_THREADS * j;
            const int r = idx / PER_ROW, c4 = (idx % PER_ROW) * 4;
            if (KC) {
                kk[j] = c4;
                tok[j] = (t0 + r) < n_t;
                tlim[j] = 4;
                p[j] = P + (size_t)(tok[j] ? t0 + r : 0) * ld + c4;
                sts[j] = r * LDK + c4;
            } else {
                kk[j] = r;
                tok[j] = (t0 + c4) < n_t;
                tlim[j] = n_t - (t0 + c4);
                p[j] = P + (size_t)r * ld + (tok[j] ? t0 + c4 : 0);
                sts[j] = r * T + c4;
            }
        }
    }
    // item j of the k-tile that starts at k0.  RAW load only (clamped address, always readable): nothing here uses the
    // loaded value, so the request stays in flight across the MFMAs of the current tile; out-of-range elements are
    // zeroed when the register goes to LDS (store1), one k-tile later.
    // MASKED = false (inside the full k-tiles): no predicate, no select -- one load, one pointer bump; rows / columns of
    // the tile that lie outside the matrix were clamped to row / column 0 at init and feed only outputs that are never
    // stored.  (An operand with unaligned rows qualifies only where its tile is interior: see dense_tile.)
    template <bool MASKED = true>
    __device__ __forceinline__ void load1(float4 (&r)[NL], int j, int k0) {
        if (VEC && !MASKED) {
            r[j] = ld4(p[j]);
        } else if (!MASKED) {               // unaligned rows, tile interior: four plain dword loads
            r[j] = make_float4(p[j][0], p[j][1], p[j][2], p[j][3]);
        } else if (VEC) {
            const bool ok = tok[j] && (k0 + kk[j] < K);
            r[j] = ld4(ok ? p[j] : base);
        } else if (KC) {
            const int kl = K - (k0 + kk[j]);                 // valid k elements from here
            float e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = *((tok[j] && q < kl) ? p[j] + q : base);
            r[j] = make_float4(e[0], e[1], e[2], e[3]);
        } else {
            const bool kok = (k0 + kk[j]) < K;
            float e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = *((kok && q < tlim[j]) ? p[j] + q : base);
            r[j] = make_float4(e[0], e[1], e[2], e[3]);
        }
        p[j] += step;
    }
    __device__ __forceinline__ void load(float4 (&r)[NL], int k0) {
#pragma unroll
        for (int j = 0; j < NL; ++j) load1(r, j, k0);
    }
    // register j of the k-tile that starts at k0 -> LDS, invalid elements as zeros (`r` is masked in place)
    template <bool MASKED = true>
    __device__ __forceinline__ void store1(float* __restrict__ S, float4 (&r)[NL], int j, int k0) const {
        if (!MASKED) {
        } else if (VEC) {
            r[j] = sel4(tok[j] && (k0 + kk[j] < K), r[j]);
        } else if (KC) {
            const int kl = K - (k0 + kk[j]);
            r[j] = make_float4((tok[j] && 0 < kl) ? r[j].x : 0.f, (tok[j] && 1 < kl) ? r[j].y : 0.f,
                               (tok[j] && 2 < kl) ? r[j].z : 0.f, (tok[j] && 3 < kl) ? r[j].w : 0.f);
        } else {
            const bool kok = (k0 + kk[j]) < K;
            r[j] = make_float4((kok && 0 < tlim[j]) ? r[j].x : 0.f, (kok && 1 < tlim[j]) ? r[j].y : 0.f,
                               (kok && 2 < tlim[j]) ? r[j].z : 0.f, (kok && 3 < tlim[j]) ? r[j].w : 0.f);
        }
        *reinterpret_cast<float4*>(S + sts[j]) = r[j];
    }
    __device__ __forceinline__ void store(float* __restrict__ S, float4 (&r)[NL], int k0) const {
#pragma unroll
        for (int j = 0; j < NL; ++j) store1(S, r, j, k0);
    }
    // fragment of chunk c of the wave's 32-k slice `wk`, 32-wide block `wt`: f[q], q = MFMA step inside the chunk
    __device__ __forceinline__ void frag(float (&f)[4], const float* __restrict__ S, int wt, int wk, int c, int lane) const {
        const int fh = lane >> 5, fc = lane & 31;
        if (KC) {
            const float4 v = *reinterpret_cast<const float4*>(S + (32 * wt + fc) * LDK + 32 * wk + 8 * c + 4 * fh);
            f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = S[(32 * wk + 8 * c + 4 * fh + q) * T + 32 * wt + fc];
        }
    }
};

template <int KIND, int WM, int WN, int WK, bool AVEC>
__device__ __forceinline__ float dense_tile(const DenseJobDev& d, int g, int tm, int tn, float* lds) {
    float sq = 0.f;                                  // squares of the values this thread stores (PFPN_DENSE_NORM)
    constexpr int TM = 32 * WM, TN = 32 * WN, BKT = 32 * WK;
    constexpr bool AKC = KIND != PFPN_DENSE_TN;
    constexpr bool BKC = KIND == PFPN_DENSE_NT;
    typedef RowOperand<AKC, TM, BKT, AVEC> OpA;
    typedef RowOperand<BKC, TN, BKT, true> OpB;
    constexpr int AF = OpA::FLOATS, BF = OpB::FLOATS;
    const PfpnDenseJob& j = d.j;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int m0 = tm * TM, n0 = tn * TN;
    const float* A = j.A + (size_t)g * j.sA;
    const float* B = j.B + (size_t)g * j.sB;
    const int K = j.K;

    OpA opA;
    OpB opB;
    opA.init(A, j.lda, m0, j.M, K, t);
    opB.init(B, j.ldb, n0, j.N, K, t);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);

    // Pipeline: two LDS stages, one register set.  Per k-tile t: every fragment of the wave's k-slice is read up front
    // (one LDS round trip per tile, not one per MFMA pair), then the 16 MFMAs issue (one dependent chain, 64 cycles each)
    // with ONE staging item in the shadow of each: the registers of tile t+1 go to the other LDS stage (its readers passed
    // the last barrier), then the loads of tile t+2 refill them (one iteration, ~0.5 us+, to land).  Items are
    // unconditional and the body is one basic block -- a branch per item makes hipcc drain vmcnt(0) at every join.  The
    // loop runs twice: over the k-tiles whose staged and loaded tiles lie wholly inside K with no predicate or select at
    // all (fp32 MFMA and VALU share the SIMD's issue: ~100 address / mask instructions per tile cost more than the
    // memory they guard), then over the last tiles with the masks (out-of-range elements become zeros on their way to LDS).
    constexpr int NA = OpA::NL, NB = OpB::NL;
    static_assert(2 * (NA + NB) <= 16, "one staging item per MFMA gap");
    constexpr int STAGE = AF + BF;
    float4 ra[NA], rb[NB];
    opA.load(ra, 0);
    opB.load(rb, 0);
    opA.store(lds, ra, 0);
    opB.store(lds + AF, rb, 0);
    if (KIND == PFPN_DENSE_TN) {
#pragma unroll
        for (int q = 0; q < NB; ++q) { csum.x += rb[q].x; csum.y += rb[q].y; csum.z += rb[q].z; csum.w += rb[q].w; }
    }
    opA.load(ra, BKT);
    opB.load(rb, BKT);
    __syncthreads();
    int buf = 0;
    int k0 = 0;
#ifdef RD_DIAG_NO_LOOP
    k0 = K;
#endif
#define RD_TILE(MASKED_)                                                                                     \
    do {                                                                                                     \
        const float* Ac = lds + buf * STAGE;                                                                 \
        const float* Bc = Ac + AF;                                                                           \
        float* An = lds + (buf ^ 1) * STAGE;                                                                 \
        float* Bn = An + AF;                                                                                 \
        float fa[16], fb[16];                                                                                \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                                   \
            opA.frag(*reinterpret_cast<float(*)[4]>(&fa[4 * c_]), Ac, wm, wk, c_, lane);                     \
            opB.frag(*reinterpret_cast<float(*)[4]>(&fb[4 * c_]), Bc, wn, wk, c_, lane);                     \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                                                  \
            RD_MFMA(i_);                                                                                     \
            RD_STAGE_ITEM(MASKED_, i_);                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
        __syncthreads();                                                                                     \
        buf ^= 1;                                                                                            \
    } while (0)
#ifndef RD_DIAG_NO_MFMA
#define RD_MFMA(i_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i_], fb[i_], acc, 0, 0, 0)
#else
#define RD_MFMA(i_) acc[i_] += fa[i_] * fb[i_]
#endif
#ifndef RD_DIAG_NO_STAGE
#define RD_STAGE_ITEM(MASKED_, i_)                                                                           \
    do {                                                                                                     \
        if ((i_) < NA) opA.template store1<MASKED_>(An, ra, (i_), k0 + BKT);                                 \
        else if ((i_) < NA + NB) {                                                                           \
            opB.template store1<MASKED_>(Bn, rb, (i_) - NA, k0 + BKT);                                       \
            if (KIND == PFPN_DENSE_TN) {                                                                     \
                const float4 u_ = rb[(i_) - NA];                                                             \
                csum.x += u_.x; csum.y += u_.y; csum.z += u_.z; csum.w += u_.w;                              \
            }                                                                                                \
        } else if ((i_) < 2 * NA + NB) opA.template load1<MASKED_>(ra, (i_) - NA - NB, k0 + 2 * BKT);        \
        else if ((i_) < 2 * (NA + NB)) opB.template load1<MASKED_>(rb, (i_) - 2 * NA - NB, k0 + 2 * BKT);    \
    } while (0)
#else
#define RD_STAGE_ITEM(MASKED_, i_) do { } while (0)
#endif
    // full tiles: iteration t stores tile t+1 and loads tile t+2, both inside K (and the block's rows / columns exist or
    // were clamped)
    // (an unaligned tile-contiguous A -- the 197-wide state as X^T -- may end inside a float4: its edge tile keeps the masks)
    const bool fast = AVEC || AKC || (m0 + TM <= j.M);
    if (fast)
        for (; k0 + 3 * BKT <= K; k0 += BKT) RD_TILE(false);
    for (; k0 < K; k0 += BKT) RD_TILE(true);
#undef RD_TILE
#undef RD_MFMA
#undef RD_STAGE_ITEM

    // ---- the K split over the block's waves: partial accumulators through LDS, summed in wave order ----------------
    if (WK > 1) {
        float* red = lds;
        if (wk > 0) {
            float* dst = red + (((wk - 1) * WM + wm) * WN + wn) * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r * 64] = acc[r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int s = 1; s < WK; ++s) {
                const float* src = red + (((s - 1) * WM + wm) * WN + wn) * 1024 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += src[r * 64];
            }
        }
    }

    // ---- epilogue (waves wk == 0).  32x32 C/D map: col = lane & 31, row = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) -----
    if (wk == 0) {
        const int fh = lane >> 5, fc = lane & 31;
        const int col = n0 + 32 * wn + fc;
        const bool cok = col < j.N;
        float* C = j.C + (size_t)g * j.sC;
        const float bv = ((j.flags & PFPN_DENSE_BIAS) && cok) ? j.bias[(size_t)g * j.sBias + col] : 0.f;
        const bool relu = j.flags & PFPN_DENSE_RELU6;
        const float* H = (j.flags & PFPN_DENSE_MASK) ? j.H + (size_t)g * j.sH : nullptr;
        const bool head = (KIND == PFPN_DENSE_NN) && (j.flags & PFPN_DENSE_HEADPART) && (j.hgroup < 0 || j.hgroup == g);
        const float hwv = (head && cok) ? j.hw[(j.hgroup < 0 ? (size_t)g * j.sHw : 0) + col] : 0.f;
        const int hparts = (j.N + 31) >> 5;
        float* hp = head ? j.aux + (j.hgroup < 0 ? (size_t)g * j.sAux : 0) : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * wm + 8 * (r >> 2) + 4 * fh + (r & 3);
            const bool ok = cok && row < j.M;
            float v = acc[r] + bv;
            if (relu) v = fminf(fmaxf(v, 0.f), 6.f);
            if (H) {
                const float h = ok ? H[(size_t)row * j.ldh + col] : 0.f;
                v = (h > 0.f && h < 6.f) ? v : 0.f;
            }
            if (ok) { C[(size_t)row * j.ldc + col] = v; sq += v * v; }
            if (head) {
                // sum over the 32 columns (lanes) of this lane half: four row_shr steps inside each row of 16 lanes, then
                // row_bcast:15 carries the sum of rows 0 / 2 into rows 1 / 3 -- lanes 31 and 63 hold the two halves' sums
                float pv = ok ? v * hwv : 0.f;
                pv += dpp_shift<0x111, 0xf>(pv, 0.f);
                pv += dpp_shift<0x112, 0xf>(pv, 0.f);
                pv += dpp_shift<0x114, 0xf>(pv, 0.f);
                pv += dpp_shift<0x118, 0xf>(pv, 0.f);
                pv += dpp_shift<0x142, 0xa>(pv, 0.f);
                if (fc == 31 && row < j.M) hp[(size_t)row * hparts + ((n0 >> 5) + wn)] = pv;
            }
        }
    }
    // ---- BiasAddGrad: column sums of the B operand (dY), threads with the same column group in fixed order ---------
    if (KIND == PFPN_DENSE_TN && (j.flags & PFPN_DENSE_COLSUM) && tm == 0) {
        constexpr int CG = TN / 4;              // column groups; thread t owns group t % CG on every load
        __syncthreads();
        float4* red4 = reinterpret_cast<float4*>(lds);
        red4[t] = csum;
        __syncthreads();
        if (t < CG) {
            float4 s = red4[t];
            for (int q = 1; q < RD_THREADS / CG; ++q) {
                const float4 u = red4[t + q * CG];
                s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
            }
            const int col = n0 + 4 * t;
            if (col < j.N) {
                *reinterpret_cast<float4*>(j.aux + (size_t)g * j.sAux + col) = s;
                sq += (s.x * s.x + s.y * s.y) + (s.z * s.z + s.w * s.w);
            }
        }
        __syncthreads();                             // (the caller may reuse the LDS words the column sums went through)
    }
    return sq;
}

// ---- helpers on [rows] vectors (no early exits: every thread reaches the block's square-sum hand-off) -----------------
// out[g][b] = hb[g*sBias] + sum_t part[g][b][t] (t ascending): finishes a scalar head whose per-32-column partial dot
// products the previous layer's epilogue left.
__device__ __forceinline__ void aux_head_final(const PfpnDenseJob& j, int local) {
    const int per = (j.M + RD_THREADS - 1) / RD_THREADS;
    const int g = local / per;
    const int b = (local - g * per) * RD_THREADS + threadIdx.x;
    if (b < j.M) {
        const float* part = j.A + (size_t)g * j.sA + (size_t)b * j.K;
        float s = j.bias[(size_t)g * j.sBias];
        for (int q = 0; q < j.K; ++q) s += part[q];
        j.C[(size_t)g * j.sC + b] = s;
    }
}
// C[g][b][c] = dv[g][b] * w[g][c] * [0 < H[g][b][c] < 6]: backward of a scalar head into the (ReLU6) layer below it
__device__ __forceinline__ void aux_rank1(const PfpnDenseJob& j, int local) {
    const int n4 = j.N >> 2;
    const size_t per_g = (size_t)j.M * n4;
    const int bpg = (int)((per_g + RD_THREADS - 1) / RD_THREADS);
    const int g = local / bpg;
    const size_t i = (size_t)(local - g * bpg) * RD_THREADS + threadIdx.x;
    if (i < per_g) {
        const int b = (int)(i / n4), c = (int)(i % n4) * 4;
        const float dv = j.A[(size_t)g * j.sA + b];
        const float4 w = ld4(j.B + (size_t)g * j.sB + c);
        const float4 h = ld4(j.H + (size_t)g * j.sH + (size_t)b * j.ldh + c);
        float4 o;
        o.x = (h.x > 0.f && h.x < 6.f) ? dv * w.x : 0.f;
        o.y = (h.y > 0.f && h.y < 6.f) ? dv * w.y : 0.f;
        o.z = (h.z > 0.f && h.z < 6.f) ? dv * w.z : 0.f;
        o.w = (h.w > 0.f && h.w < 6.f) ? dv * w.w : 0.f;
        *reinterpret_cast<float4*>(j.C + (size_t)g * j.sC + (size_t)b * j.ldc + c) = o;
    }
}
// dw[g][c] = sum_b dv[g][b] * h[g][b][c], db[g] = sum_b dv[g][b].  A block owns 32 columns; its 8 row phases each keep
// four interleaved partial sums (rows ph + 8 q + 32 i: four loads in flight per thread instead of one dependent chain),
// combined in a fixed order.  Returns the thread's share of the outputs' squares.
__device__ __forceinline__ float aux_head_wgrad(const PfpnDenseJob& j, int local, float* lds) {
    const int cblk = (j.N + 31) / 32;
    const int g = local / (cblk + 1);
    const int cb = local - g * (cblk + 1);
    const float* dv = j.A + (size_t)g * j.sA;
    const int t = threadIdx.x;
    float sq = 0.f;
    if (cb == cblk) {                       // the bias gradient: RD_THREADS strided partial sums, then in thread order
        float s = 0.f;
        for (int b = t; b < j.M; b += RD_THREADS) s += dv[b];
        lds[t] = s;
        __syncthreads();
        if (t == 0) {
            float tot = 0.f;
            for (int q = 0; q < RD_THREADS; ++q) tot += lds[q];
            j.aux[(size_t)g * j.sAux] = tot;
            sq = tot * tot;
        }
        __syncthreads();
        return sq;
    }
    constexpr int PH = RD_THREADS / 32;
    const int c = cb * 32 + (t & 31), ph = t >> 5;
    const float* h = j.H + (size_t)g * j.sH + (c < j.N ? c : 0);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = ph;
    for (; b + 3 * PH < j.M; b += 4 * PH) {
        const float h0 = h[(size_t)b * j.ldh], h1 = h[(size_t)(b + PH) * j.ldh];
        const float h2 = h[(size_t)(b + 2 * PH) * j.ldh], h3 = h[(size_t)(b + 3 * PH) * j.ldh];
        s0 += dv[b] * h0; s1 += dv[b + PH] * h1; s2 += dv[b + 2 * PH] * h2; s3 += dv[b + 3 * PH] * h3;
    }
    for (; b < j.M; b += PH) s0 += dv[b] * h[(size_t)b * j.ldh];
    lds[t] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ph == 0 && c < j.N) {
        float tot = lds[t];
        for (int q = 1; q < PH; ++q) tot += lds[t + 32 * q];
        j.C[(size_t)g * j.sC + c] = tot;
        sq = tot * tot;
    }
    __syncthreads();
    return sq;
}
// Column sums of K partial slabs in the order of pfpn_reduce_jobs (slab_reduce_1024: k-lane l sums slabs l, l+16, ..
// in k order, then lanes 0..15 in order), so a gradient has the same bits whichever launch finishes it.  16 columns x 16
// k-lanes per block; valid on k-lane 0.
__device__ __forceinline__ float slab_sum16(const float* __restrict__ slab, int K, size_t stride, int col, bool ok,
                                            float* lds) {
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    float acc = 0.f;
    if (ok)
        for (int k = kl; k < K; k += 16) acc += slab[(size_t)k * stride + col];
    lds[kl * 16 + cl] = acc;
    __syncthreads();
    float s = lds[cl];
    if (kl == 0)
        for (int q = 1; q < 16; ++q) s += lds[q * 16 + cl];
    __syncthreads();
    return s;
}
// out[i] = sum_{k<K} A[k*lda + i] (+ sum_{k<M} B[k*ldb + i] for a second slab set), i < N: the fixed-order finish of
// per-block partial slabs (the mixture kernel's particle-grid / head-bias gradients).
__device__ __forceinline__ float aux_slabsum(const PfpnDenseJob& j, int local, float* lds) {
    const int i = local * 16 + (threadIdx.x & 15);
    const bool ok = i < j.N;
    float s = slab_sum16(j.A, j.K, (size_t)j.lda, i, ok, lds);
    if (j.B) s = s + slab_sum16(j.B, j.M, (size_t)j.ldb, i, ok, lds);
    float sq = 0.f;
    if (ok && (threadIdx.x >> 4) == 0) {
        j.C[i] = s;
        sq = s * s;
    }
    return sq;
}
// squares of a FINISHED gradient segment A[0 .. N) (N % 4 == 0), block `local` of K takes its share
__device__ __forceinline__ float aux_sqsum(const PfpnDenseJob& j, int local) {
    const float4* g4 = reinterpret_cast<const float4*>(j.A);
    const size_t n4 = (size_t)j.N >> 2;
    const size_t share = (n4 + j.K - 1) / j.K;
    const size_t lo = (size_t)local * share, hi = lo + share < n4 ? lo + share : n4;
    float s = 0.f;
    size_t i = lo + threadIdx.x;
    for (; i + 3 * RD_THREADS < hi; i += 4 * RD_THREADS) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = g4[i + u * RD_THREADS];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
    }
    for (; i < hi; i += RD_THREADS) { const float4 v = g4[i]; s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    return s;
}
// the two loss columns of the fused PPO kernel's slabs -> loss3 = {loss, policy_loss, value_loss * coef} (ppo.py:39-54):
// A -> the columns, lda = slab stride, K = slabs, p0 = 1/B, p1 = value coefficient
__device__ __forceinline__ void aux_ppo_loss(const PfpnDenseJob& j, float* lds) {
    const int cl = threadIdx.x & 15;
    const float s = slab_sum16(j.A, j.K, (size_t)j.lda, cl & 1, cl < 2, lds);
    if ((threadIdx.x >> 4) == 0 && cl < 2) lds[256 + cl] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float pl = -lds[256] * j.p0, vl = lds[257] * j.p0 * j.p1;
        j.C[0] = pl + vl;
        j.C[1] = pl;
        j.C[2] = vl;
    }
    __syncthreads();
}
// float64 chunk moments of the state (pfpn_state_prologue) -> EMA of the normaliser (utils.py:60-68): A = moments
// [K chunks][2][N], M = batch rows, bias = step, B = mean_in, H = std_in, C = mean_out, aux = std_out.  16 columns per
// block, 16 chunk lanes per column (lane l sums chunks l, l+16, ..; then lanes 0..15 in order).
__device__ __forceinline__ void aux_norm_ema(const PfpnDenseJob& j, int local, float* lds) {
    double* dred = reinterpret_cast<double*>(lds);          // [2][16][16]
    const double* partial = reinterpret_cast<const double*>(j.A);
    const int S = j.N, C = j.K;
    const int ecl = threadIdx.x & 15, ekl = threadIdx.x >> 4;      // 256 threads: 16 columns x 16 chunk lanes
    const int c = local * 16 + ecl;
    const bool active = c < S;
    double a1 = 0.0, a2 = 0.0;
    if (active)
        for (int k = ekl; k < C; k += 16) {
            a1 += partial[((size_t)k * 2 + 0) * S + c];
            a2 += partial[((size_t)k * 2 + 1) * S + c];
        }
    dred[ekl * 16 + ecl] = a1;
    dred[256 + ekl * 16 + ecl] = a2;
    __syncthreads();
    if (active && ekl == 0) {
        for (int m = 1; m < 16; ++m) { a1 += dred[m * 16 + ecl]; a2 += dred[256 + m * 16 + ecl]; }
        const double Bd = (double)j.M;
        const double mu = a1 / Bd;
        double var = a2 / Bd - mu * mu;
        if (var < 0.0) var = 0.0;
        const float bmean = (float)mu, bvar = (float)var;
        const float step = j.bias[0];
        const float dk = fminf(0.9999f, (1.f + step) / (10.f + step));
        const float omd = 1.f - dk;
        j.C[c] = dk * j.B[c] + omd * bmean;
        j.aux[c] = fmaxf(1e-6f, dk * j.H[c] + omd * sqrtf(bvar));
    }
    __syncthreads();
}

__global__ __launch_bounds__(RD_THREADS, 2) void k_dense_jobs(DenseTable tab) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // the block's job, copied out of the kernel-argument segment with CONSTANT indices (a dynamic index makes hipcc
    // copy the whole table to scratch); everything of it is wave-uniform and stays in SGPRs
    DenseJobDev d = tab.job[0];
#pragma unroll
    for (int q = 1; q < PFPN_DENSE_MAX_JOBS; ++q)
        if (q < tab.n && (int)blockIdx.x >= tab.job[q].block0) d = tab.job[q];
    const int local = blockIdx.x - d.block0;
    const int kind = d.j.kind;
    float sq = 0.f;
    if (kind == PFPN_DENSE_HEAD_FINAL) aux_head_final(d.j, local);
    else if (kind == PFPN_DENSE_RANK1) aux_rank1(d.j, local);
    else if (kind == PFPN_DENSE_HEAD_WGRAD) sq = aux_head_wgrad(d.j, local, lds);
    else if (kind == PFPN_DENSE_SLABSUM) sq = aux_slabsum(d.j, local, lds);
    else if (kind == PFPN_DENSE_SQSUM) sq = aux_sqsum(d.j, local);
    else if (kind == PFPN_DENSE_PPO_LOSS) aux_ppo_loss(d.j, lds);
    else if (kind == PFPN_DENSE_NORM_EMA) aux_norm_ema(d.j, local, lds);
    else {
        // XCD-aware tile order.  Blocks are dealt to the 8 XCDs round-robin by index (observed, for speed only): blocks whose
        // index within the job is equal mod 8 share an L2.  Deal each of these 8 classes a CONTIGUOUS range of the job's
        // tiles (tm fastest, then tn, then the group): an XCD then works a few column panels of one group against that
        // group's rows -- a working set of ~2.5 MB in its 4 MiB L2 -- instead of every 8th tile of everything.
        const int per_g = d.tiles_m * d.tiles_n;
        int lin = local;
        if (d.xcd_remap) {
            const int T = per_g * d.j.batch, full = T >> 3, extra = T & 7;
            const int x = local & 7, r = local >> 3;
            lin = x * full + (x < extra ? x : extra) + r;
        }
        const int g = lin / per_g;
        const int rem = lin - g * per_g;
        const int tn = rem / d.tiles_m, tm = rem - tn * d.tiles_m;
        const int variant = (kind * 3 + d.j.cfg) * 2 + d.avec;
#define RD_CASE(KIND_, CFG_, WM_, WN_, WK_)                                                          \
    case (KIND_ * 3 + CFG_) * 2 + 1: sq = dense_tile<KIND_, WM_, WN_, WK_, true>(d, g, tm, tn, lds); break;
#define RD_CASE_U(KIND_, CFG_, WM_, WN_, WK_)                                                        \
    case (KIND_ * 3 + CFG_) * 2 + 0: sq = dense_tile<KIND_, WM_, WN_, WK_, false>(d, g, tm, tn, lds); break;
        switch (variant) {
            RD_CASE(PFPN_DENSE_NN, 0, 2, 2, 1) RD_CASE(PFPN_DENSE_NN, 1, 1, 2, 2) RD_CASE(PFPN_DENSE_NN, 2, 1, 1, 4)
            RD_CASE(PFPN_DENSE_NT, 0, 2, 2, 1) RD_CASE(PFPN_DENSE_NT, 1, 1, 2, 2) RD_CASE(PFPN_DENSE_NT, 2, 1, 1, 4)
            RD_CASE(PFPN_DENSE_TN, 0, 2, 2, 1) RD_CASE(PFPN_DENSE_TN, 1, 1, 2, 2) RD_CASE(PFPN_DENSE_TN, 2, 1, 1, 4)
            RD_CASE_U(PFPN_DENSE_NN, 0, 2, 2, 1) RD_CASE_U(PFPN_DENSE_NN, 1, 1, 2, 2) RD_CASE_U(PFPN_DENSE_NN, 2, 1, 1, 4)
            RD_CASE_U(PFPN_DENSE_TN, 0, 2, 2, 1) RD_CASE_U(PFPN_DENSE_TN, 1, 1, 2, 2) RD_CASE_U(PFPN_DENSE_TN, 2, 1, 1, 4)
            default: break;
        }
#undef RD_CASE
#undef RD_CASE_U
    }
    if (tab.sq_partial) {
        // this block's share of the global gradient norm: the squares of what it stored where the job says they are
        // gradients (PFPN_DENSE_NORM), threads in order -- fixed order, no atomics; EVERY block writes its slot
        __syncthreads();
        lds[threadIdx.x] = (d.j.flags & PFPN_DENSE_NORM) ? sq : 0.f;
        __syncthreads();
        if (threadIdx.x < 64) {
            float s4 = (lds[threadIdx.x] + lds[threadIdx.x + 64]) + (lds[threadIdx.x + 128] + lds[threadIdx.x + 192]);
            s4 = wave_sum(s4);
            if (threadIdx.x == 0) {
                tab.sq_partial[blockIdx.x] = s4;
                if (blockIdx.x == 0 && tab.beta_pow)
                    for (int k = 0; k < tab.n_groups; ++k) {          // powers: used <- next, next <- next * beta
                        float* bp = tab.beta_pow + 4 * k;
                        bp[2] = bp[0]; bp[3] = bp[1];
                        bp[0] *= tab.b1; bp[1] *= tab.b2;
                    }
            }
        }
    }
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static const int CFG_TM[3] = {64, 32, 32}, CFG_TN[3] = {64, 64, 32}, CFG_WK[3] = {1, 2, 4};

static size_t cfg_lds_bytes(int kind, int cfg) {
    const int TM = CFG_TM[cfg], TN = CFG_TN[cfg], BKT = 32 * CFG_WK[cfg];
    const size_t af = (kind != PFPN_DENSE_TN) ? (size_t)TM * (BKT + 4) : (size_t)BKT * TM;
    const size_t bf = (kind == PFPN_DENSE_NT) ? (size_t)TN * (BKT + 4) : (size_t)BKT * TN;
    return 2 * (af + bf) * sizeof(float);
}

// blocks of a product under a tile arrangement
static long long cfg_blocks(const PfpnDenseJob& j, int cfg) {
    return (long long)((j.M + CFG_TM[cfg] - 1) / CFG_TM[cfg]) * ((j.N + CFG_TN[cfg] - 1) / CFG_TN[cfg]) * j.batch;
}

}  // namespace pfpn

using namespace pfpn;

extern "C" {

static int dense_jobs_launch(const PfpnDenseJob* jobs, int n_jobs, float* sq_partial, size_t sq_capacity, int* n_partials,
                             float* beta_pow, int n_groups, float beta1, float beta2, void* stream) {
    PFPN_BEGIN();
    if (!jobs || n_jobs < 1 || n_jobs > PFPN_DENSE_MAX_JOBS) return PFPN_ERR_ARG;
    static const int force_cfg = getenv("PFPN_DENSE_CFG") ? atoi(getenv("PFPN_DENSE_CFG")) : -1;
    static const int xcd_remap = getenv("PFPN_DENSE_XCD") ? atoi(getenv("PFPN_DENSE_XCD")) : 1;
    DenseTable tab{};
    tab.n = n_jobs;
    tab.sq_partial = sq_partial; tab.beta_pow = beta_pow; tab.n_groups = n_groups; tab.b1 = beta1; tab.b2 = beta2;
    long long blocks = 0, gemm_work = 0;
    size_t lds = 8 * RD_THREADS * sizeof(float);
    // pass 1: validate, total work of the launch's products (the arrangement of each depends on what shares the chip)
    for (int i = 0; i < n_jobs; ++i) {
        const PfpnDenseJob& j = jobs[i];
        if (j.M < 1 || j.batch < 1) return PFPN_ERR_ARG;
        if (j.kind <= PFPN_DENSE_TN) {
            if (j.kind < 0 || !j.A || !j.B || !j.C || j.N < 1 || j.K < 1) return PFPN_ERR_ARG;
            gemm_work += (long long)j.M * j.N * j.batch * (long long)j.K;
        }
    }
    for (int i = 0; i < n_jobs; ++i) {
        DenseJobDev& d = tab.job[i];
        d.j = jobs[i];
        PfpnDenseJob& j = d.j;
        d.block0 = (int)blocks;
        d.xcd_remap = xcd_remap;
        if (j.kind <= PFPN_DENSE_TN) {
            const bool a_kc = j.kind != PFPN_DENSE_TN, b_kc = j.kind == PFPN_DENSE_NT;
            // B operand: float4 loads only
            if (!al16(j.B) || (j.ldb & 3) || (j.sB & 3) || (b_kc ? (j.K & 3) : (j.N & 3))) return PFPN_ERR_ARG;
            if (j.lda < (a_kc ? j.K : j.M) || j.ldb < (b_kc ? j.K : j.N) || j.ldc < j.N) return PFPN_ERR_ARG;
            d.avec = al16(j.A) && !(j.lda & 3) && !(j.sA & 3) && !((a_kc ? j.K : j.M) & 3);
            if (!d.avec && j.kind == PFPN_DENSE_NT) return PFPN_ERR_ARG;
            if ((j.flags & PFPN_DENSE_BIAS) && !j.bias) return PFPN_ERR_ARG;
            if ((j.flags & PFPN_DENSE_MASK) && (!j.H || j.ldh < j.N)) return PFPN_ERR_ARG;
            if (j.flags & PFPN_DENSE_COLSUM) {
                if (j.kind != PFPN_DENSE_TN || !j.aux || !al16(j.aux) || (j.sAux & 3)) return PFPN_ERR_ARG;
            }
            if (j.flags & PFPN_DENSE_HEADPART) {
                if (j.kind != PFPN_DENSE_NN || !j.aux || !j.hw || j.hgroup >= j.batch) return PFPN_ERR_ARG;
            }
            int cfg = j.cfg;
            if (cfg < 0 || cfg > 2) {
                // the largest tile arrangement that still gives the product its share of ~3/4 of the CUs (measured on the
                // learner's layers at 32 .. 1024 rows, tools/rowdense_gemm_probe.py): bigger tiles move fewer operand bytes
                // per MFMA, but at one block per CU an idle CU costs more than that
                cfg = 0;
                if (force_cfg >= 0 && force_cfg <= 2) cfg = force_cfg;
                else {
                    const double share = gemm_work > 0 ? (double)j.M * j.N * j.batch * (double)j.K / (double)gemm_work : 1.0;
                    const double target = 192.0 * share;
                    for (int c = 0; c <= 2; ++c) {
                        if (c > 0 && j.K / CFG_WK[c] < 32) break;
                        cfg = c;
                        if ((double)cfg_blocks(j, c) >= target) break;
                    }
                }
            }
            j.cfg = cfg;
            d.tiles_m = (j.M + CFG_TM[cfg] - 1) / CFG_TM[cfg];
            d.tiles_n = (j.N + CFG_TN[cfg] - 1) / CFG_TN[cfg];
            blocks += (long long)d.tiles_m * d.tiles_n * j.batch;
            const size_t need = cfg_lds_bytes(j.kind, cfg);
            if (need > lds) lds = need;
        } else if (j.kind == PFPN_DENSE_HEAD_FINAL) {
            if (!j.A || !j.C || !j.bias || j.K < 1) return PFPN_ERR_ARG;
            blocks += (long long)((j.M + RD_THREADS - 1) / RD_THREADS) * j.batch;
        } else if (j.kind == PFPN_DENSE_RANK1) {
            if (!j.A || !j.B || !j.C || !j.H || j.N < 4 || (j.N & 3) || (j.ldc & 3) || (j.ldh & 3) || (j.sB & 3) ||
                (j.sC & 3) || (j.sH & 3) || !al16(j.B) || !al16(j.C) || !al16(j.H))
                return PFPN_ERR_ARG;
            blocks += (((long long)j.M * (j.N >> 2) + RD_THREADS - 1) / RD_THREADS) * j.batch;
        } else if (j.kind == PFPN_DENSE_HEAD_WGRAD) {
            if (!j.A || !j.C || !j.H || !j.aux || j.N < 1) return PFPN_ERR_ARG;
            blocks += (long long)((j.N + 31) / 32 + 1) * j.batch;
        } else if (j.kind == PFPN_DENSE_SLABSUM) {
            if (!j.A || !j.C || j.N < 1 || j.K < 1 || j.lda < j.N || (j.B && (j.M < 1 || j.ldb < j.N))) return PFPN_ERR_ARG;
            blocks += (j.N + 15) / 16;
        } else if (j.kind == PFPN_DENSE_SQSUM) {
            if (!j.A || j.N < 4 || (j.N & 3) || !al16(j.A) || j.K < 1 || j.K > 4096) return PFPN_ERR_ARG;
            blocks += j.K;
        } else if (j.kind == PFPN_DENSE_PPO_LOSS) {
            if (!j.A || !j.C || j.K < 1 || j.lda < 2) return PFPN_ERR_ARG;
            blocks += 1;
        } else if (j.kind == PFPN_DENSE_NORM_EMA) {
            if (!j.A || !j.B || !j.H || !j.C || !j.aux || !j.bias || j.N < 1 || j.K < 1 || j.M < 1 || ((uintptr_t)j.A & 7))
                return PFPN_ERR_ARG;
            blocks += (j.N + 15) / 16;
        } else {
            return PFPN_ERR_ARG;
        }
        if (blocks > (1 << 20)) return PFPN_ERR_ARG;
    }
    if (sq_partial) {
        if (!n_partials || (size_t)blocks > sq_capacity) return PFPN_ERR_ARG;
        *n_partials = (int)blocks;
    }
    static size_t lds_set = 0;
    if (lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_jobs), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(72 * 1024)) != hipSuccess)
            return PFPN_ERR_LAUNCH;
        lds_set = 72 * 1024;
    }
    hipLaunchKernelGGL(k_dense_jobs, dim3((unsigned)blocks), dim3(RD_THREADS), lds, (hipStream_t)stream, tab);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_dense_jobs(const PfpnDenseJob* jobs, int n_jobs, void* stream) {
    return dense_jobs_launch(jobs, n_jobs, nullptr, 0, nullptr, nullptr, 0, 0.f, 0.f, stream);
}

int pfpn_dense_jobs_norm(const PfpnDenseJob* jobs, int n_jobs, float* sq_partial, size_t sq_capacity, int* n_partials,
                         float* beta_pow, int n_groups, float beta1, float beta2, void* stream) {
    if (!sq_partial || !n_partials || (beta_pow && (n_groups < 1 || n_groups > 4))) return PFPN_ERR_ARG;
    return dense_jobs_launch(jobs, n_jobs, sq_partial, sq_capacity, n_partials, beta_pow, n_groups, beta1, beta2, stream);
}

}  // extern "C"
