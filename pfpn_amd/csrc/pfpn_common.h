// Shared device helpers for the PFPN gfx950 kernels.  CDNA4 only: 64-lane wavefronts,
// 160 KiB LDS per CU, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define PFPN_OK 0
#define PFPN_ERR_ARG 1
#define PFPN_ERR_LAUNCH 2
#define PFPN_ERR_WORKSPACE 3

#define PFPN_WAVE 64
#define PFPN_HALF_LOG_2PI 0.9189385175704956f   // saved graph: actor/Normal/prob/add/x
#define PFPN_LOG2E 1.4426950408889634f
#define PFPN_SQRT_HALF_LOG2E 0.8493218002880191f   // sqrt(0.5 * log2 e)
#define PFPN_SQRT_2LN2 1.1774100225154747f         // sqrt(2 ln 2) = 1 / sqrt(0.5 log2 e)
#define PFPN_2LN2 1.3862943611198906f              // 2 ln 2

// Clear any stale sticky error left by other HIP users of this thread (e.g. PyTorch's benign
// hipErrorNotReady from event queries) so PFPN_CHECK_LAUNCH reports only our own launches.
#define PFPN_BEGIN() (void)hipGetLastError()

#define PFPN_CHECK_LAUNCH()                                                              \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            fprintf(stderr, "[pfpn] %s:%d launch failed: %s\n", __FILE__, __LINE__,      \
                    hipGetErrorString(e__));                                             \
            return PFPN_ERR_LAUNCH;                                                      \
        }                                                                                \
    } while (0)

namespace pfpn {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Full-wavefront reductions on the DPP network (no LDS crossbar): row_shr 1/2/4/8 inside each row of
// 16 lanes, then row_bcast:15 and row_bcast:31 carry the row totals upward; the grand total lands in
// lane 63 and is returned as a wave-uniform (scalar) value.  6 DPP steps + one v_readlane, versus
// 6 ds_bpermute round trips for the __shfl_xor tree.  Fixed order -> deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, ROW_MASK, 0xf,
                                                      false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_shift<0x111, 0xf>(v, 0.f);      // row_shr:1
    v += dpp_shift<0x112, 0xf>(v, 0.f);      // row_shr:2
    v += dpp_shift<0x114, 0xf>(v, 0.f);      // row_shr:4
    v += dpp_shift<0x118, 0xf>(v, 0.f);      // row_shr:8   -> lane 15 of every row holds the row sum
    v += dpp_shift<0x142, 0xa>(v, 0.f);      // row_bcast:15 into rows 1 and 3
    v += dpp_shift<0x143, 0xc>(v, 0.f);      // row_bcast:31 into rows 2 and 3 -> lane 63 = total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    const float ninf = -INFINITY;
    v = fmaxf(v, dpp_shift<0x111, 0xf>(v, ninf));
    v = fmaxf(v, dpp_shift<0x112, 0xf>(v, ninf));
    v = fmaxf(v, dpp_shift<0x114, 0xf>(v, ninf));
    v = fmaxf(v, dpp_shift<0x118, 0xf>(v, ninf));
    v = fmaxf(v, dpp_shift<0x142, 0xa>(v, ninf));
    v = fmaxf(v, dpp_shift<0x143, 0xc>(v, ninf));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Reductions over the 4 lanes of a quad (lanes 4k..4k+3): two quad_perm DPP steps, result in all 4 lanes.
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_shift<0xB1, 0xf>(v, 0.f);       // quad_perm [1,0,3,2]  (xor 1)
    v += dpp_shift<0x4E, 0xf>(v, 0.f);       // quad_perm [2,3,0,1]  (xor 2)
    return v;
}
// (the maximum as two hand-written v_max_f32_dpp: through fmaxf() hipcc emits a move, a DPP move and two canonicalising
// v_max_f32 x, x around each step -- nine instructions for these two.  s_nop 4: see row_step_* below.)
__device__ __forceinline__ float quad_max(float v) {
    float a, b;
    asm("s_nop 4\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(v));
    asm("s_nop 4\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(b) : "v"(a));
    return b;
}

// Four independent wave reductions for the price of ~one: a transposing butterfly on gfx950's lane-swap instructions.
// v_permlane16_swap(a, b) -> {[a.row0 b.row0 a.row2 b.row2], [a.row1 b.row1 a.row3 b.row3]} (rows of 16 lanes) and
// v_permlane32_swap(a, b) -> {[a.lo32 | b.lo32], [a.hi32 | b.hi32]} (tools/permlane_probe.hip prints both): combining the
// two results of a swap halves the number of live values while every lane keeps working.  r0..r3 -> 2 values -> ONE value
// whose row j (lanes 16j..16j+15) carries r_j summed over the four rows' lanes of the same column; four DPP steps
// (row_ror 8, 4, quad_perm xor 2, xor 1) finish inside the row.  10 VALU instructions, no LDS round trip (the previous
// form: 6 selects, 5 DPP steps and two ds_bpermute for the xor-16 / xor-32 steps).  Fixed order -> deterministic.
// (max as v_med3_f32(a, b, +inf): fmaxf() makes hipcc canonicalise both operands first -- a v_max_f32 x, x each -- whenever
// they come out of a lane swap or a DPP move; the values here are finite products, never signalling NaNs)
template <bool IS_MAX>
__device__ __forceinline__ float red_comb(float a, float b) { return IS_MAX ? __builtin_amdgcn_fmed3f(a, b, INFINITY) : a + b; }

// one in-row butterfly step k = op(k, k rotated / permuted inside its row of 16): the sum folds into v_add_f32_dpp by
// itself; for the maximum the DPP form is written out (hipcc emits fill + v_mov_dpp + 2 canonicalisations + v_max).
// s_nop 4: a DPP read needs two wait states after the VALU write of its source and five after a VALU write of EXEC; hipcc's
// hazard recogniser cannot see the DPP inside the asm, so the longer of the two is always paid (five idle cycles).
#define PFPN_ROW_STEP(name, ctrl_text, CTRL)                                                                        \
    template <bool IS_MAX>                                                                                          \
    __device__ __forceinline__ float name(float k) {                                                                \
        if (IS_MAX) {                                                                                               \
            float r;                                                                                                \
            asm("s_nop 4\n\tv_max_f32_dpp %0, %1, %1 " ctrl_text " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(k)); \
            return r;                                                                                               \
        }                                                                                                           \
        return k + dpp_shift<CTRL, 0xf>(k, 0.f);                                                                    \
    }
PFPN_ROW_STEP(row_step_ror8, "row_ror:8", 0x128)
PFPN_ROW_STEP(row_step_ror4, "row_ror:4", 0x124)
PFPN_ROW_STEP(row_step_xor2, "quad_perm:[2,3,0,1]", 0x4E)
PFPN_ROW_STEP(row_step_xor1, "quad_perm:[1,0,3,2]", 0xB1)
#undef PFPN_ROW_STEP

// -> every lane holds the total of r[wave_reduce4_row(lane)]: results left DISTRIBUTED, for per-row follow-up math that
// can then run once across the four rows instead of four times on wave-uniform values.
__device__ __forceinline__ int wave_reduce4_row(int lane) { return lane >> 4; }

template <bool IS_MAX>
__device__ __forceinline__ float wave_reduce4_dist(float r0, float r1, float r2, float r3, int /*lane*/) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(r0), __float_as_uint(r1), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(r2), __float_as_uint(r3), false, false);
    const float s01 = red_comb<IS_MAX>(__uint_as_float(a[0]), __uint_as_float(a[1]));   // rows [r0 r1 r0 r1]
    const float s23 = red_comb<IS_MAX>(__uint_as_float(b[0]), __uint_as_float(b[1]));   // rows [r2 r3 r2 r3]
    const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
    float k = red_comb<IS_MAX>(__uint_as_float(c[0]), __uint_as_float(c[1]));           // rows [r0 r1 r2 r3]
    k = row_step_ror8<IS_MAX>(k);
    k = row_step_ror4<IS_MAX>(k);
    k = row_step_xor2<IS_MAX>(k);
    k = row_step_xor1<IS_MAX>(k);
    return k;
}

// the same, results wave-uniform (v_readlane of one lane per row)
template <bool IS_MAX>
__device__ __forceinline__ void wave_reduce4(float& r0, float& r1, float& r2, float& r3, int lane) {
    const float k = wave_reduce4_dist<IS_MAX>(r0, r1, r2, r3, lane);
    r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k), 0));
    r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k), 16));
    r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k), 32));
    r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k), 48));
}

// Block-wide sum in a fixed order (deterministic): wave shuffle, then wave 0 adds the
// per-wave partials left to right.  `red` needs >= blockDim.x/64 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// The NaN/Inf guard of MixtureGaussianDistribution.log_prob's custom gradient
// (reference networks/utils.py:109-117): zero an incoming gradient that is NaN or +-Inf.
__device__ __forceinline__ float guard_finite(float q) {
    return (isnan(q) || isinf(q)) ? 0.f : q;
}

// log(1 - tanh(x)^2) as the reference writes it: 2*(log 2 - x - softplus(-2x))
// (reference networks/utils.py:132-133).
__device__ __forceinline__ float tanh_log_det(float x) {
    const float t = -2.f * x;
    const float sp = fmaxf(t, 0.f) + log1pf(expf(-fabsf(t)));
    return 2.f * (0.6931471805599453f - x - sp);
}

// Deterministic column reduction of K slabs: out[c] = sum_k slab[k*stride + c] (or max).
// 1024-thread block = 64 columns x 16 k-lanes; every k-lane walks k = lane, lane+16, ... with its
// loads issued 4 at a time (independent of the running sum), then the 16 partials are combined
// through LDS in lane order.  Replaces a serial K-long dependent-load loop (K * ~0.5 us).
template <bool IS_MAX>
__device__ __forceinline__ float slab_reduce_1024(const float* __restrict__ slab, int K, size_t stride, int col,
                                                  bool col_ok, float (*red)[64]) {
    const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;       // 16 k-lanes
    float acc = IS_MAX ? -INFINITY : 0.f;
    if (col_ok) {
        int k = kl;
        // 16 loads in flight per thread (K = 256 slabs: one round trip instead of four); additions stay in k order
        for (; k + 240 < K; k += 256) {
            float a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = slab[(size_t)(k + 16 * j) * stride + col];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = IS_MAX ? fmaxf(acc, a[j]) : acc + a[j];
        }
        for (; k + 48 < K; k += 64) {
            const float a0 = slab[(size_t)k * stride + col];
            const float a1 = slab[(size_t)(k + 16) * stride + col];
            const float a2 = slab[(size_t)(k + 32) * stride + col];
            const float a3 = slab[(size_t)(k + 48) * stride + col];
            if (IS_MAX) acc = fmaxf(fmaxf(fmaxf(fmaxf(acc, a0), a1), a2), a3);
            else acc = (((acc + a0) + a1) + a2) + a3;
        }
        for (; k < K; k += 16) {
            const float a0 = slab[(size_t)k * stride + col];
            acc = IS_MAX ? fmaxf(acc, a0) : acc + a0;
        }
    }
    red[kl][cl] = acc;
    __syncthreads();
    float s = red[0][cl];
    if (kl == 0) {
        for (int j = 1; j < 16; ++j) s = IS_MAX ? fmaxf(s, red[j][cl]) : s + red[j][cl];
    }
    __syncthreads();
    return s;          // valid on k-lane 0
}

// Philox4x32-10 (Salmon et al., SC'11: the counter-based generator TF and PyTorch use) and the SAC learner step's four
// noise values of element i drawn from ONE block {counter = (i, offset), key = seed}: rows 0, 1 of u [4][n] become Gumbel
// noise -log(-log u), rows 2, 3 a Box-Muller pair of independent standard normals (utils.py:156-161 for s and s').
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void sac_noise_element(float* __restrict__ u, size_t n, size_t i, unsigned long long seed,
                                                  unsigned long long offset) {
    unsigned r[4];
    philox4x32_10((unsigned)i, (unsigned)(i >> 32), (unsigned)offset, (unsigned)(offset >> 32), (unsigned)seed,
                  (unsigned)(seed >> 32), r);
    // 24-bit uniforms in [0, 1), as torch's uniform_() makes them for float32
    const float u0 = (r[0] >> 8) * 5.9604644775390625e-08f, u1 = (r[1] >> 8) * 5.9604644775390625e-08f;
    const float u2 = (r[2] >> 8) * 5.9604644775390625e-08f, u3 = (r[3] >> 8) * 5.9604644775390625e-08f;
    u[i] = -logf(-logf(fmaxf(u0, 1.17549435e-38f)));
    u[n + i] = -logf(-logf(fmaxf(u1, 1.17549435e-38f)));
    const float rr = sqrtf(-2.f * logf(fmaxf(u2, 1.17549435e-38f)));
    float sn, cs;
    sincosf(6.28318530717958647692f * u3, &sn, &cs);
    u[2 * n + i] = rr * cs;
    u[3 * n + i] = rr * sn;
}

// tf.clip_by_value of the normalised advantage (actor_critic.py:158-170); bounds -inf / +inf = no clipping.  Written
// with comparisons so a NaN advantage stays NaN, as Minimum/Maximum leave it in TensorFlow.
__device__ __forceinline__ float clip_adv(float a, float lo, float hi) {
    a = a < lo ? lo : a;
    return a > hi ? hi : a;
}

}  // namespace pfpn
