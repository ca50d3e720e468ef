/*
 * libpfpn_hip.so -- C ABI of the MI355X-native (gfx950) PFPN particle-policy hot path.
 *
 * The reference (xupei0610/PFPN) is pure TF-1.14 Python: it has NO FFI, plugin registry or C
 * ABI for this path.  Each entry point below therefore replaces a TF op
 * SUBGRAPH that the reference's Python builds; the citation names the reference lines whose
 * arithmetic the entry point reproduces.  A maintainer binds these with ctypes from the
 * reference-shaped Python classes.
 *
 * Conventions
 *  - every pointer is DEVICE memory (fp32 unless typed otherwise), row-major, owned by the
 *    caller; nothing is allocated or freed inside; calls are stream-ordered on `stream`
 *    (a hipStream_t passed as void*; NULL = default stream) and never synchronise, so they can
 *    be captured into a hipGraph;
 *  - shapes: B batch rows, A action dims, N particles per dim, S state dim.
 *    logits [B, A*N] (particle (a,n) at column a*N+n), loc/logstd/max_active/sum_active [A,N],
 *    action/x [B,A];
 *  - `workspace` is caller-provided scratch of at least pfpn_*_workspace_bytes();
 *  - return value: PFPN_OK (0), PFPN_ERR_ARG (1: null pointer / bad shape), PFPN_ERR_LAUNCH (2:
 *    HIP launch error), PFPN_ERR_WORKSPACE (3: workspace missing or too small).
 *  - reductions over the batch use fixed-order partial slabs: results are bitwise reproducible.
 */
#ifndef PFPN_HIP_H_
#define PFPN_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFPN_OK 0
#define PFPN_ERR_ARG 1
#define PFPN_ERR_LAUNCH 2
#define PFPN_ERR_WORKSPACE 3

/* library / ABI version, and a cheap "is the device code loadable" probe (no compute). */
int pfpn_abi_version(void);

/* ---- mixture log-prob (K1/K2) ------------------------------------------------------------ */
/* scratch needed by the mixture entry points for this shape */
size_t pfpn_mixture_workspace_bytes(int B, int A, int N);

/* MixtureGaussianDistribution.log_prob -- reference networks/utils.py:108-134.
 * logp[b] = sum_a log( sum_n softmax(logits[b,a,:])_n * Normal(x[b,a]; loc[a,n], exp(logstd[a,n])) )
 * tanh_mode: x is the pre-tanh value and 2(log2 - x - softplus(-2x)) is subtracted (utils.py:132-133).
 * pmix (optional, may be NULL): the per-dim mixture density p[B,A]. */
int pfpn_logprob_fwd(const float* logits, const float* x, const float* loc, const float* logstd,
                     int B, int A, int N, int tanh_mode, float* logp, float* pmix,
                     void* workspace, size_t workspace_bytes, void* stream);

/* TF autodiff of the above incl. the NaN/Inf guard of utils.py:109-117.
 * dlogp [B] in; dlogits [B,A*N], dloc [A,N], dlogstd [A,N] out (overwritten, not accumulated);
 * dx [B,A] optional (needed by SAC where x depends on the parameters). */
int pfpn_logprob_bwd(const float* logits, const float* x, const float* loc, const float* logstd,
                     const float* dlogp, int B, int A, int N, int tanh_mode,
                     float* dlogits, float* dloc, float* dlogstd, float* dx,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Advantage moments for normalize_advantage -- reference actor_critic.py:151-156:
 * stats = {mean, 1/(sqrt(population var)+1e-8)}; normalize=0 writes {0,1}. */
int pfpn_adv_stats(const float* adv, int B, int normalize, float* stats, void* stream);

/* Fused learner kernel: log_prob(action_hist) + clipped-PPO loss head + full backward in one pass
 * over the logits (reference ppo.py:31-54, actor_critic.py:132-184, utils.py:108-134).
 * Outputs: logp [B], dlogits [B,A*N], dloc/dlogstd [A,N], dvalue [B] (d loss / d value),
 * loss3 = {loss, policy_loss, value_loss*coef}.  Returns PFPN_ERR_ARG for shapes the packed path
 * does not cover (N > 64 or no template instance): compose fwd + pfpn_ppo_loss + bwd instead. */
int pfpn_ppo_mixture_fwd_bwd(const float* logits, const float* action, const float* loc, const float* logstd,
                             const float* logp_old, const float* adv, const float* value,
                             const float* value_old, const float* adv_stats,
                             int B, int A, int N, float clip_eps, float value_coef,
                             float* logp, float* dlogits, float* dloc, float* dlogstd, float* dvalue,
                             float* loss3, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PFPN_HIP_H_ */
