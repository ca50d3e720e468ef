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
 *
 * Concurrency (what may be in flight at the same time, on different streams of one process)
 *  - NO entry point keeps state of a CALL in the library: every intermediate (partial slabs, in-launch hand-off tickets,
 *    staging) lives in memory the caller hands in -- the outputs and `workspace`.  (The library's only process-wide memory
 *    is a per-device cache of device properties -- CU count, one occupancy figure, the > 64 KB LDS opt-in -- keyed by
 *    hipGetDevice() and filled on first use with values that do not depend on the call.)  Two calls are independent, and may run
 *    concurrently on two streams, exactly when they share no output and no workspace; a workspace is exclusive to one
 *    call at a time (calls that reuse it must be ordered: same stream, or events).  Read-only inputs may be shared.
 *  - A workspace needs no initialisation and keeps no meaning between calls: the mixture entry points zero the hand-off
 *    tickets they use (the last words of a pfpn_mixture_workspace_bytes() workspace) stream-ordered in front of their
 *    launch, so an uninitialised, reused or poisoned workspace gives the same bits as a fresh one.  A workspace smaller
 *    than pfpn_*_workspace_bytes() asks for is refused with PFPN_ERR_WORKSPACE (or, where a launch-per-piece form needs
 *    less, served by that form).
 *  - The exceptions are documented at their prototypes: pfpn_debug_stamps() and pfpn_sac_noise_fill()'s draw counter
 *    are explicit caller-owned device state; pfpn_ppo_wide_fused() needs the device to itself while it runs (grid-wide
 *    co-residency); environment knobs (PFPN_WIDE_C, PFPN_WIDE_STREAM) are read once, PFPN_WIDE_F2L_SAFE (tests) at every call.
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
/* diagnostic only: device buffer (16*64 uint64) receiving s_memtime phase stamps of block 0 of the tile
 * mixture kernel (k_mix_tile); NULL (default) disables stamping. */
int pfpn_debug_stamps(void* device_buffer);

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

/* N >= 64 shapes (where pfpn_ppo_mixture_fwd_bwd does not apply): the forward above AND the clipped-PPO head of
 * pfpn_ppo_loss (reference ppo.py:31-54, actor_critic.py:132-184) in ONE launch -- the block that finishes a row chunk's
 * log-probs runs the head on those rows; loss3 = {loss, policy_loss, value_loss*coef} is complete when the launch ends.
 * PFPN_ERR_ARG where that form does not apply (N < 64 shapes, A % 4 != 0, ...): call the two entry points instead. */
int pfpn_logprob_fwd_ppo(const float* logits, const float* action, const float* loc, const float* logstd,
                         const float* logp_old, const float* adv, const float* value, const float* value_old,
                         const float* adv_stats, int B, int A, int N, float clip_eps, float value_coef,
                         float* logp, float* dlogp, float* dvalue, float* loss3,
                         void* workspace, size_t workspace_bytes, void* stream);

/* pfpn_logprob_fwd_ppo + pfpn_logprob_bwd(_bias) of a wide particle grid (64 <= N <= 256, N % 4 == 0, A % 4 == 0, A <= 128, a
 * batch of at least 8 row chunks of 4 rows) as ONE launch that reads the logits once and writes dlogits once (reference networks/utils.py:108-134 +
 * its autodiff, ppo.py:31-54, actor_critic.py:132-184): the waves of a row chunk exchange their per-dimension log p through
 * memory once per four rows (k_mix_wide_f2l; 379 us against 598 us for 8192 x 128 x 256).  A row chunk = (B + C - 1) / C CONSECUTIVE
 * rows, which a wave addresses through one 32-bit buffer window: PFPN_ERR_ARG when a chunk's rows span 2 GiB or more
 * (> 16384 rows per chunk at A * N = 32768).  Same outputs as the two entry points
 * (+ dlogp [B] = d loss / d logp); sums in a fixed order: bitwise reproducible run to run, equal to the two-launch composition to
 * summation order.  dbias optional (NULL: no bias plane).  *chunks_out (optional) receives the chunk count.
 * CONCURRENCY: the waves of a row chunk wait for each other, so a chunk's blocks (8 * A / 4 consecutive blocks of the dispatch
 * order) must be resident together: the entry point sizes the grid to the device's occupancy, and NOTHING else may occupy the
 * device while it runs (not a second call on another stream, not another kernel, not another process's work: on a SHARED
 * GPU compose the two entry points instead -- the Python package: PFPN_WIDE_FUSED=0).  Every in-kernel wait is bounded
 * (~0.2 s): on expiry the gradients come out NaN, word 65 of the workspace's ticket area is set (that word is zeroed in
 * front of the next launch) AND 1 is OR-ed into *fault_flag -- a device word the CALLER owns, zeroes once and reads wherever it
 * synchronises anyway (loss read-back, checkpoint): nothing in the library ever clears it, so one expired launch stays
 * visible however many launches follow (NULL: no sticky record).  A loud failure, not a hang (tests/test_gpu_mixture.py
 * forces it).  PFPN_ERR_ARG for shapes outside the form (compose the two entry points). */
int pfpn_ppo_wide_fused(const float* logits, const float* action, const float* loc, const float* logstd,
                        const float* logp_old, const float* adv, const float* value, const float* value_old,
                        const float* adv_stats, int B, int A, int N, float clip_eps, float value_coef,
                        float* logp, float* dlogits, float* dloc, float* dlogstd, float* dvalue, float* dlogp,
                        float* dbias, float* loss3, int* chunks_out, uint32_t* fault_flag,
                        void* workspace, size_t workspace_bytes, void* stream);

/* TF autodiff of the above incl. the NaN/Inf guard of utils.py:109-117.
 * dlogp [B] in; dlogits [B,A*N], dloc [A,N], dlogstd [A,N] out (overwritten, not accumulated);
 * dx [B,A] optional (needed by SAC where x depends on the parameters). */
int pfpn_logprob_bwd(const float* logits, const float* x, const float* loc, const float* logstd,
                     const float* dlogp, int B, int A, int N, int tanh_mode,
                     float* dlogits, float* dloc, float* dlogstd, float* dx, int finalize,
                     void* workspace, size_t workspace_bytes, void* stream);
/* pfpn_logprob_bwd(finalize = 1) that also leaves dbias [A*N] = the column sums of dlogits (fc_policy's bias gradient,
 * a2c.py:554 BiasAdd) -- for particle grids the wide kernel serves (N >= 64, A <= 4096; else PFPN_ERR_ARG and the caller
 * sums the columns in a pass of its own): a third plane of the kernel's slabs, finished in its hand-off.  Workspace:
 * pfpn_mixture_workspace_bytes. */
int pfpn_logprob_bwd_bias(const float* logits, const float* x, const float* loc, const float* logstd,
                          const float* dlogp, int B, int A, int N, int tanh_mode,
                          float* dlogits, float* dloc, float* dlogstd, float* dx, float* dbias,
                          void* workspace, size_t workspace_bytes, void* stream);
/* finalize = 0: the per-block partial sums of dloc / dlogstd stay in the workspace (dloc, dlogstd may be NULL) as
 * *nslab slabs of *stride floats [dloc A*N | dlogstd A*N | ..]; a PFPN_JOB_COLSUM job sums them in the same order. */
int pfpn_logprob_bwd_slabs(const float* logits, const float* dlogits, int B, int A, int N, int* nslab, int* stride);

/* Advantage moments for normalize_advantage -- reference actor_critic.py:151-156:
 * stats = {mean, 1/(sqrt(population var)+1e-8)}; normalize=0 writes {0,1}.
 * Everywhere below, `adv_stats` / `stats` is a device array of FOUR floats {mean, rstd, clip_lo, clip_hi}: the loss heads
 * use clip((adv - mean) * rstd, clip_lo, clip_hi) (clip_advantage, actor_critic.py:158-170); this entry point and
 * pfpn_state_prologue write only the first two, the caller keeps the bounds (-inf, +inf: no clipping). */
int pfpn_adv_stats(const float* adv, int B, int normalize, float* stats, void* stream);

/* Fused learner kernel: log_prob(action_hist) + clipped-PPO loss head + full backward in one pass
 * over the logits (reference ppo.py:31-54, actor_critic.py:132-184, utils.py:108-134).
 * Outputs: logp [B], dlogits [B,A*N], dloc/dlogstd [A,N], dvalue [B] (d loss / d value),
 * dbias [A*N] (optional, may be NULL: column sums of dlogits = the fc_policy bias gradient, a2c.py:541-554),
 * loss3 = {loss, policy_loss, value_loss*coef} (NULL: the per-block partial sums are left unreduced in the workspace --
 * dloc/dlogstd/dbias/loss3 are then NOT written; used by bench.py to time the main kernel alone).
 * Returns PFPN_ERR_ARG for shapes the tile kernel
 * does not cover (N > 64, A > 44 or no template instance): compose fwd + pfpn_ppo_loss + bwd instead. */
int pfpn_ppo_mixture_fwd_bwd(const float* logits, const float* action, const float* loc, const float* logstd,
                             const float* logp_old, const float* adv, const float* value,
                             const float* value_old, const float* adv_stats,
                             int B, int A, int N, float clip_eps, float value_coef,
                             float* logp, float* dlogits, float* dloc, float* dlogstd, float* dvalue,
                             float* dbias, float* loss3, void* workspace, size_t workspace_bytes, void* stream);

/* Unfused PPO head on [B] vectors (same arithmetic as inside pfpn_ppo_mixture_fwd_bwd) --
 * reference ppo.py:31-54: dlogp [B], dvalue [B], loss3 = {loss, policy_loss, value_loss*coef}. */
int pfpn_ppo_loss(const float* logp, const float* logp_old, const float* adv, const float* value,
                  const float* value_old, const float* adv_stats, int B, float clip_eps, float value_coef,
                  float* dlogp, float* dvalue, float* loss3, void* stream);

/* ---- rollout: sample / statistics / deterministic action / entropy (K3) ------------------- */
size_t pfpn_rollout_workspace_bytes(int B, int A, int N);

/* MixtureGaussianDistribution.sample(1), plain branch -- reference networks/utils.py:187-194, fused
 * with the log-prob of the drawn action (ppo.py:27-29 build_action_sampler_with_log_prob) and the
 * running activation statistics (a2c.py:356-365) when max_active/sum_active are non-NULL.
 * u_cat [B,A] float64 in [0,1): one uniform per categorical draw (TF-1.14 CPU Multinomial rule:
 * float64 running sum of exp(double(l)-max), first index whose sum exceeds u*total);
 * eps_sel [B,A]: the standard-normal draw of the selected particle.
 * k [B,A] int32, action [B,A] = loc[a,k] + exp(logstd[a,k])*eps, logp [B] (optional). */
int pfpn_sample(const float* logits, const float* loc, const float* logstd, const double* u_cat,
                const float* eps_sel, int B, int A, int N, int32_t* k, float* action, float* logp,
                float* max_active, float* sum_active, void* workspace, size_t workspace_bytes, void* stream);

/* max_active <- max(max_active, max_b softmax), sum_active += sum_b softmax -- a2c.py:356-360. */
int pfpn_active_stats(const float* logits, int B, int A, int N, float* max_active, float* sum_active,
                      void* workspace, size_t workspace_bytes, void* stream);

/* MixtureGaussianDistribution.mean -- utils.py:202-236: loc[a, argmax_n logits] (tanh(.) in tanh mode). */
int pfpn_mean_action(const float* logits, const float* loc, int B, int A, int N, int tanh_mode,
                     int32_t* k, float* action, void* stream);

/* MixtureGaussianDistribution.entropy -- utils.py:146-151: H [B,A]; bwd: dlogits (+)= dH * dH/dlogits. */
int pfpn_entropy_fwd(const float* logits, int B, int A, int N, float* H, void* stream);
int pfpn_entropy_bwd(const float* logits, const float* dH, int B, int A, int N, int accumulate,
                     float* dlogits, void* stream);

/* Gumbel noise from uniforms in [0,1), in place: g = -log(-log(max(u, FLT_MIN))) -- the sampler of TFP 0.7's
 * RelaxedOneHotCategorical behind utils.py:96, :156-160 (the uniforms themselves come from the caller's generator). */
int pfpn_gumbel_from_uniform(float* u, size_t n, void* stream);
/* The four noise tensors of one SAC learner step (utils.py:156-161 for s and s') from uniforms u [4][n] in [0,1), in place:
 * rows 0, 1 -> Gumbel noise as above; rows 2, 3 -> a Box-Muller pair of independent N(0,1) draws. */
int pfpn_sac_noise_from_uniform(float* u, size_t n, void* stream);
/* The same four tensors drawn on the device in one launch: element i of buf [4][n] takes its four uniforms from the
 * Philox4x32-10 block {counter = (i, offset), key = seed} (the role of tf.random.uniform / tf.random.normal behind
 * utils.py:156-161; the stream itself cannot match TensorFlow's and is the caller's to choose: a new `offset` per step). */
int pfpn_sac_noise_fill(float* buf, size_t n, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream);
/* (offset_dev != NULL: the offset is read from device memory when the launch RUNS, so a captured launch draws fresh noise
 * on every replay) */

/* SAC reparameterised sample (K4) -- utils.py:156-186.  gumbel [B,A*N] = -log(-log u), eps [B,A*N]:
 * k = argmax(logits+gumbel), s_out = p_k = loc+exp(logstd)*eps, a_out = tanh(p_k).
 * bwd: custom gradients `mask`/`mask2` (utils.py:165-183) with upstream da, ds [B,A]. */
int pfpn_rsample_fwd(const float* logits, const float* loc, const float* logstd, const float* gumbel,
                     const float* eps, int B, int A, int N, int32_t* k, float* a_out, float* s_out,
                     void* stream);
int pfpn_rsample_bwd(const float* logits, const float* loc, const float* logstd, const float* gumbel,
                     const float* eps, const float* da, const float* ds, int B, int A, int N, int accumulate,
                     float* dlogits, float* dloc, float* dlogstd, int finalize, void* workspace, size_t workspace_bytes,
                     void* stream);
/* finalize = 0: the parameter-gradient partial sums stay in the workspace: *nslab slabs of *stride floats
 * [dloc A*N | dlogstd A*N] from *float_offset floats on; the caller adds their slab-order sum (PFPN_JOB_COLSUM's second
 * slab set) to what pfpn_logprob_bwd contributed. */
int pfpn_rsample_bwd_slabs(int B, int A, int N, int* nslab, int* stride, long long* float_offset);

/* ---- resample (K5) ------------------------------------------------------------------------ */
/* ParticleFilteringA2CNetwork.build_resample_ops -- a2c.py:385-474.  topk = 0: the `resample == -1` branch
 * (source = categorical draw over log avg activation, :400-404); topk = k > 0: the `resample == k` branch (source =
 * candidate number min(k-1, floor(u_resample*k)) among the k most active particles, ties to the lower index, :405-408;
 * k is clamped to N).  fixed_sigma > 0: the `fixed_sigma` setting (:430-440, :463-465): the jitter uses that std, in tanh
 * mode around tanh(source loc), and logstd is left alone.  In place on
 * loc/logstd [A,N], bias [A*N], weight [H, A*N] (fc_policy, a particle is a column).
 * u_resample [A,N] float64 in [0,1): uniforms of tf.random.categorical(log avg, N) / of the uniform candidate choice;
 * u_noise [A,N] in [-1,1): tf.random.uniform for dead slot (a,j).  threshold <= 0 -> 0.05/N.
 * dead_out / src_out [A,N] int32: dead mask and drawn source index per slot.  The caller zeroes
 * max_active/sum_active/train_flag afterwards (a2c.py:373-378). */
int pfpn_resample(float* loc, float* logstd, float* bias, float* weight, int H,
                  const float* max_active, const float* sum_active, const double* u_resample,
                  const float* u_noise, int A, int N, float threshold, int tanh_mode, int topk, float fixed_sigma,
                  int32_t* dead_out, int32_t* src_out, void* stream);

/* ---- host utility ------------------------------------------------------------------------- */
/* CRC-32C (Castagnoli) of a HOST buffer, chained through `crc` (0 for the first call): the per-tensor and
 * per-block checksum of TensorFlow's tensor_bundle checkpoint files, which the reference writes through
 * tf.train.Saver (models/distributed_model.py:436-496).  Used by pfpn_amd/checkpoint.py. */
uint32_t pfpn_crc32c(const void* data, size_t n, uint32_t crc);

/* ---- learner utilities (K6, K8, K9, dense epilogues) ------------------------------------- */
size_t pfpn_learner_workspace_bytes(int rows, int H);

/* generalized_advantage_estimate + discount -- a2c.py:30-40, utils.py:5-15.  E ragged sequences:
 * sequence e has T_e = offsets[e+1]-offsets[e] rewards at reward[offsets[e]..] and T_e+1 values at
 * value[offsets[e]+e ..] (last one = bootstrap).  fp32 deltas, float64 sequential scan, fp32 out. */
int pfpn_gae(const float* reward, const float* value, const int64_t* offsets, int E, float gamma,
             double gae_gamma, float* adv, void* stream);

/* The learner step's prologue in one launch: x = clip((s-mean)/std) as pfpn_state_norm, plus (want_moments) the
 * float64 column moments of s for pfpn_state_norm_ema, plus (adv_stats != NULL) pfpn_adv_stats' {mean, 1/(std+1e-8)}
 * of the advantages.  pfpn_state_norm_ema turns the moments into the EMA update of pfpn_state_norm_update.
 * s2 != NULL: a second [B,S] batch (SAC's next states, sac.py:141-153) is normalised with the same moments into x2 in
 * the same launch (it does not enter the moments). */
int pfpn_state_prologue(const float* s, const float* mean, const float* std, int B, int S, float clip, float* x,
                        int want_moments, const float* adv, int normalize_adv, float* adv_stats,
                        const float* s2, float* x2, void* workspace, size_t workspace_bytes, void* stream);
int pfpn_state_norm_ema(const void* workspace, int B, int S, const float* step, const float* mean_in,
                        const float* std_in, float* mean_out, float* std_out, void* stream);

/* V-trace targets -- reference networks/actor_critic/vtrace.py:101-180.  E episodes of T steps, all [E*T]
 * vectors flat in episode-major order (index e*T + t: vtrace.py:16-37); seq_len [E] (optional, int32) marks
 * zero-padded episodes.  fp32, the reverse tf.scan's operation order; rho_clip / pg_rho_clip <= 0 = None.
 * Outputs (flat, same order): vs = value target, pg_adv = policy-gradient advantage, mask (optional) =
 * valid_data_mask as 0/1 floats. */
int pfpn_vtrace(const float* logp_target, const float* logp_behavior, const float* value, const float* reward,
                const float* not_terminal, const int32_t* seq_len, int E, int T, float gamma, float rho_clip,
                float pg_rho_clip, float* vs, float* pg_adv, float* mask, void* stream);
/* Policy-gradient loss head of the A2C / A3C / IMPALA networks -- reference a2c.py:53-65, actor_critic.py:132-184:
 * policy_loss = -mean_valid(logp*adv_n), value_loss = coef*mean_valid((v - value_target)^2); mask optional
 * (valid_data_mask, 0/1).  Outputs dlogp [B], dvalue [B], loss4 = {loss, policy_loss, value_loss*coef, n_valid}. */
int pfpn_a2c_loss(const float* logp, const float* adv, const float* value, const float* value_target,
                  const float* adv_stats, const float* mask, int B, float value_coef, float* dlogp, float* dvalue,
                  float* loss4, void* stream);

/* x = clip((s-mean)/std, +-clip) -- actor_critic.py:223-244 (clip <= 0: no clipping). */
int pfpn_state_norm(const float* s, const float* mean, const float* std, int B, int S, float clip, float* x,
                    void* stream);
/* online_normalizer(moving_average=True) -- utils.py:60-68, population moments over the batch.
 * step: DEVICE scalar holding float(global_step) (a device value so the call is graph-capturable). */
int pfpn_state_norm_update(const float* s, int B, int S, const float* step, const float* mean_in, const float* std_in,
                           float* mean_out, float* std_out, void* workspace, size_t workspace_bytes,
                           void* stream);

/* fc_layer epilogue -- ops.py:105-117: y <- act(y + bias); y [rows,H], bias [G,H], rows of group g are
 * [g*group_rows, (g+1)*group_rows) (group_rows = 0: one group). */
int pfpn_bias_act(float* y, const float* bias, int rows, int H, int group_rows, int relu6, void* stream);
/* Relu6Grad + BiasAddGrad: dy <- dy*[0<y<6] in place (relu6), db [G,H] = column sums of dy per group. */
int pfpn_act_bwd_bias_grad(const float* y, float* dy, int rows, int H, int group_rows, int relu6, float* db,
                           void* workspace, size_t workspace_bytes, void* stream);
/* BiasAdd + Relu6 (ops.py:105-117) that also leaves the Relu6Grad mask as bits -- bit (i & 63) of the 64-bit word
 * mask[(i >> 6) * 4 + k] = (0 < y + bias < 6) for float4 index i (row-major over [rows, H/4]) and component k -- and the
 * Relu6Grad + per-chunk BiasAddGrad partials of pfpn_act_bwd_partial reading those bits instead of the activations (1 bit
 * instead of 4 bytes per element).  pfpn_act_bwd_partial_mask needs H % 256 == 0. */
size_t pfpn_relu6_mask_bytes(int rows, int H);
int pfpn_bias_act_mask(float* y, const float* bias, int rows, int H, int group_rows, void* mask, size_t mask_bytes,
                       void* stream);
int pfpn_act_bwd_partial_mask(float* dy, const void* mask, int rows, int H, int group_rows, float* partial,
                              size_t partial_floats, int* chunks_per_group, void* stream);
/* Last trunk layer's epilogue with the critic head folded in (ops.py:105-117 + actor_critic.py:206-210):
 * y <- act(y + bias) on all rows, and v[b] = y[value_group rows][b,:] . value_w + value_b[0].  group_rows % 4 == 0.
 * value_group < 0: EVERY group g has its own head (twin Q networks, sac.py:123-139): value_w [G,H],
 * group g's bias at value_b[g*value_b_stride], v [G*B]. */
int pfpn_bias_act_value(float* y, const float* bias, int rows, int H, int group_rows, int relu6, int value_group,
                        const float* value_w, const float* value_b, int value_b_stride, float* v, void* stream);
/* Two epilogue passes of one width H in ONE launch -- a SAC step evaluates the live twin critics (2B rows, mask bits for the
 * backward) and the target twin critics (B rows) on unrelated inputs, and a dependent launch costs ~5 us whatever it does.
 * pfpn_bias_act_mask_pair: per pass what pfpn_bias_act_mask (mask != NULL) or pfpn_bias_act(relu6) (mask NULL) does;
 * pfpn_bias_act_value_pair: per pass pfpn_bias_act_value(relu6, value_group = -1).  Same arithmetic, same bits. */
typedef struct PfpnEpilogueJob {
    float* y;
    const float* bias;
    int rows, group_rows;
    void* mask;               /* optional (pfpn_bias_act_mask_pair) */
    size_t mask_bytes;
    const float* value_w;     /* pfpn_bias_act_value_pair: [G, H], [G * value_b_stride], v [G, group_rows] */
    const float* value_b;
    int value_b_stride;
    float* v;
} PfpnEpilogueJob;
int pfpn_bias_act_mask_pair(const PfpnEpilogueJob* a, const PfpnEpilogueJob* b, int H, void* stream);
int pfpn_bias_act_value_pair(const PfpnEpilogueJob* a, const PfpnEpilogueJob* b, int H, void* stream);
/* Relu6Grad + per-chunk BiasAddGrad partials only: like pfpn_act_bwd_bias_grad but the column sums stay as
 * partial[G][chunks_per_group][H] (chunks_per_group is returned) for pfpn_reduce_jobs.  With value_w != NULL the critic
 * head's backward is folded in: on group value_group, dy is FORMED as dvalue[b]*value_w[c] (not read), masked and
 * written, and value_partial[chunks_per_group][H] receives the partials of dW_fc3 = sum_b y[b,:]*dvalue[b].
 * value_group < 0: every group has a head -- value_w [G,H], dvalue [G*group_rows], value_partial [G][chunks][H]. */
int pfpn_act_bwd_partial(const float* y, float* dy, int rows, int H, int group_rows, int relu6, int value_group,
                         const float* value_w, const float* dvalue, float* value_partial, float* partial,
                         size_t partial_floats, int* chunks_per_group, void* stream);

/* Deferred fixed-order reductions, one launch for up to PFPN_MAX_REDUCE_JOBS jobs (the gradients the reference gets
 * from tf.gradients' BiasAddGrad / Sum nodes, sync_model.py:110-121, and the UPDATE_OPS EMA, utils.py:60-68):
 *   PFPN_JOB_COLSUM   out[i] = sum_{k<nslab} src[k*stride + i], i < n           (float slabs); in0 != NULL: plus a second
 *                     slab set, + sum_{k<p0} in0[k*p1 + i] (p0 slabs of stride p1, both exact small integers)
 *   PFPN_JOB_VECSUM   out[0] = sum_{i<n} src[i]                                  (nslab = 1)
 *   PFPN_JOB_PPO_LOSS src -> the two loss columns of pfpn_ppo_mixture_fwd_bwd's slabs; out = loss3; p0 = 1/B,
 *                     p1 = value_coef; n = 2
 *   PFPN_JOB_NORM_EMA src = float64 moments of pfpn_state_prologue (nslab = pfpn_state_moment_chunks(B) chunks),
 *                     n = S, stride = B; in0 = step, in1 = mean_in, in2 = std_in; out = mean_out, out2 = std_out */
#define PFPN_MAX_REDUCE_JOBS 32
#define PFPN_JOB_COLSUM 0
#define PFPN_JOB_VECSUM 1
#define PFPN_JOB_PPO_LOSS 2
#define PFPN_JOB_NORM_EMA 3
#define PFPN_JOB_SQSUM 4
typedef struct PfpnReduceJob {
    int kind, nslab, stride, n;
    const void* src;
    float* out;
    float* out2;
    const float* in0;
    const float* in1;
    const float* in2;
    float p0, p1;
    int norm;               /* COLSUM / VECSUM: the output is a gradient, count its squares (pfpn_reduce_jobs_norm) */
} PfpnReduceJob;
int pfpn_reduce_jobs(const PfpnReduceJob* jobs, int n_jobs, void* stream);
/* The same launch also yielding the partial sums of the global gradient norm (tf.clip_by_global_norm, base_worker.py:97-102):
 * outputs of jobs with norm = 1 are squared where they are produced; PFPN_JOB_SQSUM jobs (src = a finished gradient
 * segment, n floats, 16-byte aligned, n % 4 == 0, nslab = blocks to spread it over) cover every other gradient.  Block b
 * writes sq_partial[b]; *n_partials = number of blocks.  beta_pow != NULL: block 0 rotates the Adam powers as
 * pfpn_clip_adam's first launch does.  pfpn_clip_adam_prenorm is pfpn_clip_adam's second launch on those partials. */
int pfpn_reduce_jobs_norm(const PfpnReduceJob* jobs, int n_jobs, float* sq_partial, size_t sq_capacity, int* n_partials,
                          float* beta_pow, int n_groups, float beta1, float beta2, void* stream);
/* pfpn_clip_scale_inplace's second launch on the partial sums of pfpn_reduce_jobs_norm (powers rotated there). */
int pfpn_clip_scale_inplace_prenorm(float* grads, size_t n, float clip_norm, float* norm_scale, const float* sq_partial,
                                    int n_partials, void* stream);
/* pfpn_clip_adam_prenorm of learner step k with pfpn_state_prologue of step k+1 (want_moments = 1, no second batch) as extra
 * blocks of the SAME launch -- for multi-step hipGraph captures, where the next step's batch is known: the two kernels touch
 * disjoint data (the prologue reads the normaliser statistics that the reduction launch before this one updated), so a step
 * loses a launch without a second stream.  Same arithmetic, bit for bit, as the two separate launches.
 * (reference: base_worker.py:97-120 for the optimiser part, actor_critic.py:223-244 / utils.py:46-68 for the prologue.) */
typedef struct {
    const float* s;          /* [B,S] the next step's states */
    const float* mean;       /* [S] state-normaliser statistics (as the next step's prologue would read them) */
    const float* std;
    int B, S;
    float clip;
    float* x;                /* [B,S] normalised states out */
    const float* adv;        /* [B] advantages (with adv_stats) */
    int normalize_adv;
    float* adv_stats;        /* [4] or NULL */
    void* workspace;         /* pfpn_learner_workspace_bytes(B, S): the float64 column moments */
    size_t workspace_bytes;
    /* SAC (sac.py:141-153): a second batch normalised with the same moments (the next states) */
    const float* s2;         /* [B,S] or NULL */
    float* x2;               /* [B,S] or NULL */
    /* SAC: the NEXT step's four noise tensors drawn by extra blocks of the same launch (as pfpn_sac_noise_fill with
     * offset_dev = noise_epoch); NULL: none.  `s` may then be NULL (noise only: the last step of a captured replay). */
    float* noise;            /* [4][noise_n] or NULL */
    size_t noise_n;
    uint64_t noise_seed;
    const uint64_t* noise_epoch;
} PfpnStatePrologue;
/* soft_target != NULL: the SAC soft target update of the parameters just written, as in pfpn_clip_adam_prenorm. */
int pfpn_clip_adam_prenorm_prologue(float* params, const float* grads, float* m, float* v, size_t n, int n_groups,
                                    const long long* group_off, const long long* group_len, const float* group_lr,
                                    float beta1, float beta2, float eps, const float* beta_pow, float clip_norm,
                                    float* norm_scale, float* step_counter, const float* sq_partial, int n_partials,
                                    const PfpnStatePrologue* next, float* soft_target, long long soft_off,
                                    long long soft_len, float soft_tau, void* stream);
int pfpn_clip_adam_prenorm(float* params, const float* grads, float* m, float* v, size_t n, int n_groups,
                           const long long* group_off, const long long* group_len, const float* group_lr, float beta1,
                           float beta2, float eps, const float* beta_pow, float clip_norm, float* norm_scale,
                           float* step_counter, const float* sq_partial, int n_partials, float* soft_target,
                           long long soft_off, long long soft_len, float soft_tau, void* stream);
int pfpn_state_moment_chunks(int B);
/* Slab geometry of pfpn_ppo_mixture_fwd_bwd called with loss3 = NULL (same arguments): nslab slabs of `stride`
 * floats laid out [dloc A*N | dlogstd A*N | (dbias A*N) | policy-loss, value-loss, pad, pad].  PFPN_ERR_ARG when the
 * tile kernel does not cover the shape. */
int pfpn_ppo_mixture_slabs(int B, int A, int N, int want_dbias, const float* logits, const float* dlogits,
                           int* nslab, int* stride);

/* critic fc3 (H -> 1, squeeze) -- actor_critic.py:206-210: v[b] = h[b,:].w + bias[0]; backward gives
 * dh = dv (x) w masked by the ReLU6 of h, dw [H], db [1]. */
int pfpn_value_head_fwd(const float* h, const float* w, const float* bias, int B, int H, float* v, void* stream);
int pfpn_value_head_bwd(const float* h, const float* w, const float* dv, int B, int H, int relu6, float* dh,
                        float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* tf.clip_by_global_norm -- base_worker.py:97-102: norm_scale = {norm, clip*min(1/norm, 1/clip)};
 * apply != 0 scales grads in place (needed before a data-parallel all-reduce). */
int pfpn_clip_by_global_norm(float* grads, size_t n, float clip_norm, int apply, float* norm_scale,
                             void* workspace, size_t workspace_bytes, void* stream);
/* TF ApplyAdam on a flat buffer (epsilon outside the bias-corrected sqrt; NOT torch.optim.Adam).
 * beta_pow [2] = {beta1^t, beta2^t} device scalars, advanced when advance_beta_pow != 0.
 * norm_scale (optional) multiplies the gradient by norm_scale[1]; grad_mul by a host scalar.
 * step_counter (optional device scalar): += 1 together with the beta powers (float(global_step)). */
int pfpn_tf_adam(float* params, const float* grads, float* m, float* v, size_t n, float lr, float beta1,
                 float beta2, float eps, float* beta_pow, int advance_beta_pow, const float* norm_scale,
                 float grad_mul, float* step_counter, void* stream);
/* One-replica optimiser step in two launches: tf.clip_by_global_norm over the whole gradient buffer folded into
 * TF ApplyAdam for up to 4 parameter groups (contiguous, 4-aligned ranges with their own learning rate: the critic /
 * actor optimisers of base_worker.py:33-45).  beta_pow is [n_groups][4] = {b1^t, b2^t as TF's beta*_power variables
 * hold them, and the pair this apply used}; group_* are HOST arrays.  Also writes norm_scale = {norm, scale} and
 * bumps step_counter (device scalar, optional).  Same arithmetic as pfpn_clip_by_global_norm + pfpn_tf_adam. */
int pfpn_clip_adam(float* params, const float* grads, float* m, float* v, size_t n, int n_groups,
                   const long long* group_off, const long long* group_len, const float* group_lr, float beta1,
                   float beta2, float eps, float* beta_pow, float clip_norm, float* norm_scale, float* step_counter,
                   void* workspace, size_t workspace_bytes, void* stream);
/* The same step split around a data-parallel all-reduce (sync_model.py:64-95: every replica clips its OWN gradients,
 * the mean of the clipped gradients is applied): pfpn_clip_scale_inplace scales grads by the clip factor (2 launches;
 * with beta_pow it also rotates the Adam powers exactly as pfpn_clip_adam's first launch does), then -- after the
 * caller's all-reduce (sum) -- pfpn_adam_groups applies TF-Adam with grad_mul = 1/W, using the "used" pair of
 * beta_pow, and bumps step_counter (1 launch); scale_buf[0..scale_n) *= scale_mul in the same launch (the replicated
 * statistics in the bucket tail: sums -> replica means, sync_model.py:37-46; scale_n = 0: none).
 * soft_target != NULL (pfpn_adam_groups, pfpn_clip_adam_prenorm): the SAC soft target update (sac.py:67-73) of the
 * parameters [soft_off, soft_off + soft_len) happens in the same launch, on the values Adam just wrote:
 * soft_target[i] <- (1 - soft_tau) soft_target[i] + soft_tau params[soft_off + i] (pfpn_soft_update's arithmetic). */
int pfpn_clip_scale_inplace(float* grads, size_t n, float clip_norm, float* norm_scale, float* beta_pow, int n_groups,
                            float beta1, float beta2, void* workspace, size_t workspace_bytes, void* stream);
/* Calibration kernel (no reference counterpart): dst[i] = src[i] as a float4 grid-stride stream, `unroll` (1, 2, 4, 8) loads
 * in flight per thread before their stores, `blocks` blocks of 256 threads, nontemporal != 0: nt loads and stores.  The
 * box's read+write rate for the mixture kernels' access pattern (bench.py: roofline_stress.device_copy).  unroll = 0: the
 * read-only stream -- src is summed into dst [blocks * 256] (one float per thread), the forward kernels' pattern. */
int pfpn_stream_copy(float* dst, const float* src, size_t n, int unroll, int nontemporal, int blocks, void* stream);
/* Sharded exchange of the same step (W replicas; sync_model.py:64-95 semantics unchanged): the mean of the per-replica
 * clipped gradients is sum_w s_w g_w / W with ONE scalar s_w per replica, so the gradient bytes need not wait for the norm.
 * pfpn_clip_scale_only leaves {norm, s_w} in norm_scale WITHOUT touching the gradients (from the partial sums of
 * pfpn_reduce_jobs_norm when sq_partial != NULL -- the powers were rotated there -- else from grads, rotating beta_pow
 * like pfpn_clip_scale_inplace).  After an all-to-all (replica w's slice j of the bucket [grads | statistics] to replica j)
 * and an all-gather of the W scalars, pfpn_shard_adam on replica j: for its bucket-space slice [shard_lo, shard_lo +
 * shard_len): gradient = (sum_w scales[w] * recv[w][i]) / W in replica order, TF-Adam ("used" powers, group learning
 * rates) on params_x / m / v at those indices; indices in [n_params, n_bucket) (the replicated statistics) become the
 * plain replica mean, written to params_x at the same index; bumps step_counter.  recv [world][shard_len].  An all-gather
 * of params_x then hands every replica the owners' bits (replicas stay bit-identical). */
int pfpn_clip_scale_only(const float* grads, size_t n, float clip_norm, float* norm_scale, const float* sq_partial,
                         int n_partials, float* beta_pow, int n_groups, float beta1, float beta2, void* workspace,
                         size_t workspace_bytes, void* stream);
int pfpn_shard_adam(float* params_x, float* m, float* v, const float* recv, const float* scales, int world,
                    long long shard_lo, long long shard_len, long long n_params, long long n_bucket, int n_groups,
                    const long long* group_off, const long long* group_len, const float* group_lr, float beta1,
                    float beta2, float eps, const float* beta_pow, float* step_counter, void* stream);
int pfpn_adam_groups(float* params, const float* grads, float* m, float* v, size_t n, int n_groups,
                     const long long* group_off, const long long* group_len, const float* group_lr, float beta1,
                     float beta2, float eps, const float* beta_pow, float grad_mul, float* step_counter,
                     float* scale_buf, size_t scale_n, float scale_mul, float* soft_target, long long soft_off,
                     long long soft_len, float soft_tau, void* stream);
/* SAC loss head -- sac.py:123-139 (vf_target, q_target) and :160-173 (Q and policy losses).
 * q_pi, q_run, q_tgt are [2,B] (Q1 row then Q2 row): Q(s, a_pi), Q(s, action_hist), Q'(s', a').
 * Outputs dq_pi [2,B] (d policy_loss/dQ through tf.minimum), dq_run [2,B] (d value_loss/dQ),
 * dlogp [B] (= alpha/B), dlog_alpha [1], loss3 = {loss, policy_loss, value_loss*coef}.
 * q_ld: row stride of q_pi, q_run, dq_pi, dq_run (0 = B; 2B when Q(s, a_pi) and Q(s, action_hist) are the two halves of
 * one [2, 2B] evaluation); q_tgt is [2,B] contiguous. */
int pfpn_sac_loss(const float* q_pi, const float* q_run, const float* q_tgt, const float* logp, const float* logp_t,
                  const float* reward, const float* not_terminal, const float* log_alpha, int B, float gamma,
                  float value_coef, float target_entropy, float* dq_pi, float* dq_run, float* dlogp,
                  float* dlog_alpha, float* loss3, int q_ld, uint64_t* noise_epoch, void* stream);
/* (noise_epoch, optional: a device counter this launch advances by one -- the offset the step's NEXT noise draw reads,
 * pfpn_sac_noise_fill(offset_dev) / PfpnStatePrologue.noise_epoch; its readers run in other launches) */
/* pfpn_sac_loss riding in the first launch of the critics' backward pass -- for the step that evaluates the twin live critics
 * ONCE over the 2B rows [(s, a_pi); (s, action_hist)] (q_both / dq_both [2, 2B]: [g][0:B] = Q_g(s, a_pi), [g][B:2B] =
 * Q_g(s, action_hist)).  = pfpn_act_bwd_partial(rows = 4B, group_rows = 2B, relu6, value_group = -1) on dvalue = the dQ of
 * pfpn_sac_loss(q_ld = 2B), which every block forms for its own rows (a row's dQ depends on that row alone) and the first
 * column tile writes to dq_both; one block adds loss3, dlog_alpha, dlogp [B] and the draw counter.  dQ: the bits of
 * pfpn_sac_loss; the three losses: equal to rounding (256 threads sum them here, 1024 there). */
typedef struct PfpnSacLoss {
    const float* q_both;
    const float* q_tgt;          /* [2, B] */
    const float* logp;
    const float* logp_t;
    const float* reward;
    const float* not_terminal;
    const float* log_alpha;
    int B;
    float gamma, value_coef, target_entropy;
    float* dq_both;
    float* dlogp;
    float* dlog_alpha;
    float* loss3;
    uint64_t* noise_epoch;       /* optional */
} PfpnSacLoss;
int pfpn_act_bwd_partial_sac(const float* y, float* dy, int H, const float* value_w, float* value_partial, float* partial,
                             size_t partial_floats, int* chunks_per_group, const PfpnSacLoss* loss, void* stream);
/* The three critic inputs of a SAC step in one launch -- sac.py:108-121 (tf.concat([state, action], -1) for Q(s, a_pi),
 * Q(s, action_hist) and the target Q'(s', a')): x, x_next [B,S] normalised states, a_* [B,A]; outputs [B,S+A]. */
int pfpn_sac_q_inputs(const float* x, const float* x_next, const float* a_pi, const float* a_hist, const float* a_next,
                      int B, int S, int A, float* xa_pi, float* xa_hist, float* xa_next, void* stream);
/* The SAC actor side of a learner step as two launches (sac.py:108-175 with utils.py:108-134, :156-186) -- what
 * pfpn_rsample_fwd + pfpn_logprob_fwd(tanh_mode) + pfpn_sac_q_inputs, and pfpn_logprob_bwd + pfpn_rsample_bwd + the
 * column sums of dlogits (fc_policy's bias gradient) compute in seven, for N <= 64 particles and A <= 256 action dims
 * (pfpn_sac_actor_supported: 1 / 0; other shapes keep the piece-wise entry points).
 * fwd: rows [0, R) of logits / gumbel / eps [R, A*N] (R = 2B: the actor on [s; s']); k, a_out = tanh(p_k), s_out = p_k
 * [R, A], logp [R].  xq_both != NULL: the critic inputs too -- x [R, S] normalised states of the same rows, a_hist [B, A];
 * row r < B -> xq_both[r] = [x_r | a_r], xq_both[B + r] = [x_r | a_hist_r]; row r >= B -> xq_next[r - B] = [x_r | a_r].
 * bwd: rows [0, B); x [B, A] = s_out, dlogp [B], da [B, A] upstream of a_out (the upstream of s_out is the log-prob's own
 * dx); dlogits [B, A*N] is written; the workspace receives *nslab slabs of *stride = 3 A N floats {dloc | dlogstd |
 * d fc_policy bias}, each summed over one block's rows -- the caller adds them in slab order (PFPN_JOB_COLSUM). */
int pfpn_sac_actor_supported(int A, int N);
int pfpn_sac_actor_fwd(const float* logits, const float* loc, const float* logstd, const float* gumbel,
                       const float* eps, int R, int A, int N, int32_t* k, float* a_out, float* s_out, float* logp,
                       const float* x, const float* a_hist, int B, int S, float* xq_both, float* xq_next,
                       void* stream);
int pfpn_sac_actor_bwd_slabs(int B, int A, int N, int* nslab, int* stride);
size_t pfpn_sac_actor_workspace_bytes(int B, int A, int N);
int pfpn_sac_actor_bwd(const float* logits, const float* loc, const float* logstd, const float* gumbel,
                       const float* eps, const float* x, const float* dlogp, const float* da, int B, int A, int N,
                       float* dlogits, void* workspace, size_t workspace_bytes, void* stream);
/* SAC soft target update -- sac.py:67-73. */
int pfpn_soft_update(float* target, const float* source, size_t n, float tau, void* stream);

/* Minibatch slicing of a training round -- reference models/distributed_model.py:320-352 (`buf[cand]` for every experience
 * buffer, on the host).  Here the rollout is staged on the device once and every learner step gathers its rows in ONE
 * launch: dst_k[b, :] = src_k[rows[b], :] for k < ncol <= 8 columns of widths[k] floats.
 * srcs / dsts / widths are HOST arrays (of device pointers / ints) -- the only host pointers of this ABI; rows [B] is
 * device int64, clamped to [0, n_src). */
int pfpn_gather_columns(const void* const* srcs, void* const* dsts, const int* widths, int ncol, const int64_t* rows,
                        long long n_src, int B, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PFPN_HIP_H_ */
