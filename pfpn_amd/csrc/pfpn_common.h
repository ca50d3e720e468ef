// Shared device helpers for the PFPN gfx950 kernels.  CDNA4 only: 64-lane wavefronts,
// 160 KiB LDS per CU, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PFPN_OK 0
#define PFPN_ERR_ARG 1
#define PFPN_ERR_LAUNCH 2
#define PFPN_ERR_WORKSPACE 3

#define PFPN_WAVE 64
#define PFPN_HALF_LOG_2PI 0.9189385175704956f   // saved graph: actor/Normal/prob/add/x

#define PFPN_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return PFPN_ERR_LAUNCH;        \
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

}  // namespace pfpn
