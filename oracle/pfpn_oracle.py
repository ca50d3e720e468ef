"""CPU restatement (NumPy, fp32 op-for-op) of the PFPN particle-policy hot path.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the shipped path (pfpn_amd/) never does and fails loudly when the
HIP library is missing.

PARITY PINNING.  The reference (xupei0610/PFPN) ships no tests and no golden input/output
vectors, and its arithmetic lives in un-vendored third-party packages (tensorflow-gpu==1.14,
tensorflow_probability==0.7, requirements.txt:1-2) that cannot be installed here.  It does ship its
own TRAINING GRAPH (ckpt_DeepMimicWalk-v0/.../model.ckpt-78000.meta, 6,052 nodes).  The oracle is
pinned to that graph:
  * tests/golden/graph_*.npz hold outputs obtained by interpreting the GraphDef node by node
    (tests/golden/graphdef_eval.py + make_graph_golden.py, build container only): log-prob, mixture
    probability, value, the three losses, all 14 gradients of the `tf.gradients` sub-graph, the
    global norm and clipped gradients, 14 x ApplyAdam (var, m, v, beta powers), the state-normaliser
    update, the activation statistics, the whole `cond/...` resample sub-graph, the rollout sampler,
    for one and two workers, including rows that hit the NaN/Inf guard.
    tests/test_oracle_graph_golden.py: this module reproduces every one of them (<= 7.6e-6 of the
    tensor scale, tests/golden/graph_golden_report.txt; integer outputs bit-exact).
  * the particle-grid initialisers and every scalar constant decoded from the shipped MetaGraphDef
    files (tests/golden/meta_constants.json, meta_samples*_init.npy; tests/test_oracle_golden.py).
What the graph cannot pin, because it is not in it: the three random ops' streams (fed as explicit
inputs; the categorical draw rule is this module's stated definition, `multinomial_index`), the
per-op arithmetic of TF's kernels below fp32 summation-order level, and the SAC / V-trace variants
(no SAC or IMPALA graph ships: those functions follow the Python source and are checked against
independent float64 autograd, tests/test_oracle_math.py).  Each function cites the lines it restates.

RNG: TF's Philox streams cannot be reproduced without TF, so every random draw is
an explicit input (u_cat, eps, gumbel, u_resample, u_noise).  Index-critical
transcendentals are evaluated in float64 and (where the reference is fp32) rounded
to fp32, so CPU and GPU agree bit-for-bit on indices (DESIGN.md "Index parity").

Paths cited are relative to /root/reference.
"""
import numpy as np

F = np.float32
HALF_LOG_2PI = F(0.9189385175704956)   # saved graph: global_net/actor/Normal/prob/add/x
LOG2 = F(np.log(2.0))


# --------------------------------------------------------------------------- #
# particle grid                                                               #
# --------------------------------------------------------------------------- #
def init_particles(A, N, tanh_mode=False, init_sigma=None):
    """networks/actor_critic/a2c.py:476-534 (build_policy, identical particle count).

    Bounds are forced to [-1, 1] (a2c.py:479-480).  Returns (loc[A,N], logstd[A,N]) fp32,
    exactly what the `samples` / `samples_std` constant initialisers hold.
    """
    u = np.ones((A, N), dtype=np.float64)
    l = -u
    ar = np.arange(N, dtype=np.float64)[None, :]
    if tanh_mode:
        loc = l + (u - l) / N * (ar + 0.5)                       # a2c.py:486
    else:
        loc = l + (u - l) / (N - 1) * ar                         # a2c.py:488
    if init_sigma:
        std = np.broadcast_to(np.asarray(init_sigma, dtype=np.float64).reshape(-1, 1)
                              if np.ndim(init_sigma) else np.float64(init_sigma), loc.shape).copy()
        if tanh_mode:                                            # a2c.py:492-498
            loc_ = loc
            loc = np.arctanh(loc)
            std = np.maximum(loc - np.arctanh(np.maximum(1e-6 - 1, loc_ - std)),
                             np.arctanh(np.minimum(1 - 1e-6, loc_ + std)) - loc)
    else:
        std = (u - l) / (N - 1)                                  # a2c.py:500
        if tanh_mode:                                            # a2c.py:501-512
            assert N > 3
            loc = np.arctanh(loc)
            d0 = loc - loc[:, np.maximum(0, np.arange(N) - 1)]
            d1 = loc[:, np.minimum(N - 1, np.arange(N) + 1)] - loc
            std = np.maximum(d0, d1)
    logstd = np.log(std)                                         # a2c.py:523
    return loc.astype(F), logstd.astype(F)


# --------------------------------------------------------------------------- #
# mixture distribution (networks/utils.py:85-236)                             #
# --------------------------------------------------------------------------- #
def softmax(logits):
    """TF `Softmax` on the last axis, fp32: exp(l - max) / sum  (Appendix A)."""
    logits = np.asarray(logits, dtype=F)
    v = logits - logits.max(axis=-1, keepdims=True)
    e = np.exp(v, dtype=F)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F)).astype(F)


def _normal_prob(x, loc, logstd):
    """tf.distributions.Normal(loc, scale).prob(x[..., None]), op order of Appendix A:
    z=(x-loc)/scale; exp(-0.5*z^2 - (0.9189385 + log(scale))), scale = exp(logstd)."""
    scale = np.exp(logstd.astype(F), dtype=F)
    z = ((x[..., None].astype(F) - loc[None]) / scale[None]).astype(F)
    arg = (F(-0.5) * (z * z) - (HALF_LOG_2PI + np.log(scale, dtype=F))[None]).astype(F)
    with np.errstate(under="ignore"):
        phi = np.exp(arg, dtype=F)
    return phi, z, scale


def _tanh_correction(x):
    """2*(log2 - x - softplus(-2x)) == log(1 - tanh(x)^2)  (utils.py:132-133)."""
    x = x.astype(F)
    sp = np.logaddexp(F(0), F(-2) * x).astype(F)         # tf.nn.softplus
    return (F(2) * (LOG2 - x - sp)).astype(F)


def mixture_log_prob(logits, x, loc, logstd, tanh_mode=False):
    """MixtureGaussianDistribution.log_prob, networks/utils.py:108-134.

    logits [B,A,N], x [B,A] (the pre-tanh value s_ when tanh_mode), loc/logstd [A,N].
    Returns (logp[B], p[B,A]).  Computed in probability space like the reference (no
    log-sum-exp): p may underflow to 0 -> logp = -inf.
    """
    w = softmax(logits)
    phi, _, _ = _normal_prob(np.asarray(x, F), loc, logstd)
    p = (w * phi).sum(axis=-1, dtype=F)
    with np.errstate(divide="ignore"):
        lp = np.log(p, dtype=F)
    if tanh_mode:
        lp = lp - _tanh_correction(np.asarray(x, F))
    return lp.sum(axis=-1, dtype=F), p


def mixture_log_prob_bwd(logits, x, loc, logstd, dlogp, tanh_mode=False):
    """Autodiff of mixture_log_prob as TF builds it (Appendix A "Backward of log-prob",
    NaN/Inf guard of utils.py:109-117 applied to d/dp right after Log_grad).

    Returns dlogits[B,A,N], dloc[A,N], dlogstd[A,N], dx[B,A].
    """
    logits = np.asarray(logits, F)
    x = np.asarray(x, F)
    w = softmax(logits)
    phi, z, scale = _normal_prob(x, loc, logstd)
    p = (w * phi).sum(axis=-1, dtype=F)
    g = np.asarray(dlogp, F)[:, None]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        q = (g / p).astype(F)
    q = np.where(np.isnan(q) | np.isinf(q), F(0), q).astype(F)       # foo() guard
    qw = (q[..., None] * w).astype(F)
    dlogits = (qw * (phi - p[..., None])).astype(F)                  # softmax grad
    t = (qw * phi).astype(F)                                         # dL/dphi * phi
    dloc = (t * z / scale[None]).sum(axis=0, dtype=F)
    dlogstd = (t * (z * z - F(1))).sum(axis=0, dtype=F)
    dx = -(t * z / scale[None]).sum(axis=-1, dtype=F)
    if tanh_mode:
        dx = dx + g * F(2) * np.tanh(x, dtype=F)
    return dlogits, dloc.astype(F), dlogstd.astype(F), dx.astype(F)


def categorical_entropy(logits):
    """MixtureGaussianDistribution.entropy, networks/utils.py:146-151 -> H[B,A]."""
    logits = np.asarray(logits, F)
    v = logits - logits.max(axis=-1, keepdims=True)
    s0 = np.exp(v, dtype=F)
    s1 = s0.sum(axis=-1, keepdims=True, dtype=F)
    p = s0 / s1
    return (p * (np.log(s1, dtype=F) - v)).sum(axis=-1, dtype=F)


def categorical_entropy_bwd(logits, dH):
    """d/dlogits of categorical_entropy: -w * (log w + H) * dH."""
    logits = np.asarray(logits, F)
    v = logits - logits.max(axis=-1, keepdims=True)
    s0 = np.exp(v, dtype=F)
    s1 = s0.sum(axis=-1, keepdims=True, dtype=F)
    w = s0 / s1
    logw = v - np.log(s1, dtype=F)
    H = -(w * logw).sum(axis=-1, keepdims=True, dtype=F)
    return (-(w * (logw + H)) * np.asarray(dH, F)[..., None]).astype(F)


def multinomial_index(logits2d, u):
    """TF-1.14 CPU `Multinomial` draw rule (definition): per row
    cdf_j = sum_{i<=j} exp(double(l_i) - max l) (fp64 running sum, left to right),
    pick the first j with cdf_j > u * cdf_{N-1}  (std::upper_bound).

    logits2d [R,N] fp32, u [R] or [R,S] in [0,1) fp64 -> int32 index [R] / [R,S]."""
    l = np.asarray(logits2d, F).astype(np.float64)
    m = l.max(axis=-1, keepdims=True)
    e = np.exp(l - m)
    cdf = np.empty_like(e)
    run = np.zeros(e.shape[:-1], np.float64)
    for j in range(e.shape[-1]):                 # explicit left-to-right order
        run = run + e[..., j]
        cdf[..., j] = run
    u = np.asarray(u, np.float64)
    if u.ndim == l.ndim - 1:
        target = (u * run)[..., None]
        k = (cdf <= target).sum(axis=-1)
    else:                                        # several draws per row
        target = u * run[..., None]
        k = (cdf[..., None, :] <= target[..., None]).sum(axis=-1)
    return np.minimum(k, l.shape[-1] - 1).astype(np.int32)


def sample_plain(logits, loc, logstd, u_cat, eps_sel):
    """MixtureGaussianDistribution.sample(1), plain branch, networks/utils.py:187-194.

    k ~ Categorical(logits) via multinomial_index(u_cat[B,A]); a = loc[a,k] + scale[a,k]*eps.
    The reference draws eps for all N particles and keeps one; `eps_sel[B,A]` is the
    kept draw (same distribution).  Returns (k[B,A] int32, action[B,A] fp32)."""
    B, A, N = logits.shape
    k = multinomial_index(np.asarray(logits, F).reshape(B * A, N),
                          np.asarray(u_cat, np.float64).reshape(B * A)).reshape(B, A)
    scale = np.exp(logstd.astype(F), dtype=F)
    ai = np.arange(A)[None, :]
    a = (loc[ai, k] + scale[ai, k] * np.asarray(eps_sel, F)).astype(F)
    return k, a


def active_stats(logits, max_active, sum_active):
    """a2c.py:356-360: max_active <- max(max_active, max_B w); sum_active += sum_B w."""
    w = softmax(logits)
    new_max = np.maximum(max_active, w.max(axis=0)).astype(F)
    new_sum = (sum_active + w.sum(axis=0, dtype=F)).astype(F)
    return new_max, new_sum


def mean_action(logits, loc, tanh_mode=False):
    """MixtureGaussianDistribution.mean, networks/utils.py:202-236: loc[a, argmax_N];
    tanh(loc) in tanh mode (argmax of softmax == argmax of logits, first index on ties)."""
    k = np.asarray(logits, F).argmax(axis=-1)
    ai = np.arange(loc.shape[0])[None, :]
    a = loc[ai, k].astype(F)
    if tanh_mode:
        a = np.tanh(a, dtype=F)
    return k.astype(np.int32), a


def rsample(logits, loc, logstd, gumbel, eps):
    """MixtureGaussianDistribution.sample(1), rsample branch (SAC), utils.py:156-186.

    gumbel [B,A,N] = -log(-log u) (TFP 0.7 ExpRelaxedOneHotCategorical, T=1), eps [B,A,N].
    Returns (a=tanh(p_k) [B,A], s_=p_k [B,A], k [B,A] int32)."""
    y = (np.asarray(logits, F) + np.asarray(gumbel, F)).astype(F)
    k = y.argmax(axis=-1)
    scale = np.exp(logstd.astype(F), dtype=F)
    p = (loc[None] + scale[None] * np.asarray(eps, F)).astype(F)
    pk = np.take_along_axis(p, k[..., None], axis=-1)[..., 0]
    return np.tanh(pk, dtype=F), pk.astype(F), k.astype(np.int32)


def rsample_bwd(logits, loc, logstd, gumbel, eps, da, ds):
    """Custom gradients `mask` / `mask2` of utils.py:165-183 followed by the
    softmax((logits+g)/1) Jacobian and p = loc + scale*eps.

    da, ds [B,A]: upstream grads w.r.t. a (= tanh p_k) and s_ (= p_k).
    Returns dlogits [B,A,N], dloc [A,N], dlogstd [A,N]."""
    y = (np.asarray(logits, F) + np.asarray(gumbel, F)).astype(F)
    w = softmax(y)
    k = y.argmax(axis=-1)
    scale = np.exp(logstd.astype(F), dtype=F)
    eps = np.asarray(eps, F)
    p = (loc[None] + scale[None] * eps).astype(F)
    t = np.tanh(p, dtype=F)
    tk = np.take_along_axis(t, k[..., None], axis=-1)
    gap = (t - tk).astype(F)
    da = np.asarray(da, F)[..., None]
    ds = np.asarray(ds, F)[..., None]
    gw = (gap * da + gap / np.maximum(F(1e-6), F(1) - tk * tk) * ds).astype(F)   # dL/dw
    dlogits = (w * (gw - (gw * w).sum(axis=-1, keepdims=True, dtype=F))).astype(F)
    m = np.zeros_like(p)
    np.put_along_axis(m, k[..., None], F(1), axis=-1)
    # a = sum m*tanh(p): grad to tanh_p is m*da, through tanh: (1-t^2); s_: m*ds
    gp = (m * (da * (F(1) - t * t) + ds)).astype(F)
    dloc = gp.sum(axis=0, dtype=F)
    dlogstd = (gp * scale[None] * eps).sum(axis=0, dtype=F)
    return dlogits, dloc.astype(F), dlogstd.astype(F)


# --------------------------------------------------------------------------- #
# dead-particle resampling (a2c.py:385-474)                                   #
# --------------------------------------------------------------------------- #
def _seq_sum_rows(x):
    """Left-to-right fp32 row sum (the order the HIP kernel uses, so indices agree)."""
    s = np.zeros(x.shape[0], F)
    for j in range(x.shape[1]):
        s = (s + x[:, j]).astype(F)
    return s


def resample_particles(loc, logstd, bias, weight, max_active, sum_active,
                       u_resample, u_noise, threshold=None, tanh_mode=False, topk=0, fixed_sigma=None):
    """ParticleFilteringA2CNetwork.build_resample_ops (a2c.py:385-474).  topk = 0: the `resample == -1` branch
    (:400-404, the one the shipped settings use, deepmimic_dppo_pfpn.py:5); topk = k > 0: the `resample == k` branch
    (:405-408): source = candidate number min(k-1, floor(u_resample*k)) among the k most active particles of the
    action dimension (tf.math.top_k; ties to the lower index; the reference's sorted=False order is unspecified and the
    choice is uniform, so the distribution does not depend on it).

    loc, logstd, max_active, sum_active [A,N]; bias [A*N]; weight [H, A*N] (a particle is a
    column).  u_resample [A,N] in [0,1): uniforms for `tf.random.categorical(log avg, N)`;
    u_noise [A,N] in [-1,1): uniform for dead particle (a,j) (`tf.random.uniform(-1,1)`,
    indexed by the dead particle's own slot so the kernel needs no compaction).
    Returns new (loc, logstd, bias, weight) and `info` (dead mask, source index per slot).
    Log of the average activation is fl32(log_f64(avg)) -- see module docstring.
    """
    A, N = loc.shape
    thr = F(threshold) if threshold else F(.05 / N)              # a2c.py:391
    s = _seq_sum_rows(sum_active.astype(F))
    with np.errstate(divide="ignore", invalid="ignore"):
        avg = (sum_active.astype(F) / s[:, None]).astype(F)      # a2c.py:394
        log_avg = np.log(avg.astype(np.float64)).astype(F)       # a2c.py:402 (tf.math.log)
    dead = max_active.astype(F) < thr                            # a2c.py:395
    if topk and topk > 0:
        k = min(N, int(topk))                                     # a2c.py:405
        order = np.argsort(-avg, axis=1, kind="stable")[:, :k]    # a2c.py:407 (ties: lower index first)
        choice = np.clip((np.asarray(u_resample, np.float64) * k).astype(np.int64), 0, k - 1)   # a2c.py:408
        src = np.take_along_axis(order, choice, axis=1)           # src[a,j] = cand[a, choice(a,j)]
    else:
        cand = multinomial_index(log_avg, np.asarray(u_resample, np.float64))   # [A,N] draws
        # choice = invalid[:,1] (a2c.py:403): dead particle (a,j) takes draw number j
        src = cand                                                # src[a,j] = cand[a, j]
    loc = loc.astype(F).copy(); logstd = logstd.astype(F).copy()
    bias = bias.astype(F).copy(); weight = weight.astype(F).copy()
    old_loc, old_logstd, old_bias, old_weight = loc.copy(), logstd.copy(), bias.copy(), weight.copy()
    scale = np.exp(old_logstd, dtype=F)
    for a in range(A):
        js = np.nonzero(dead[a])[0]
        if js.size == 0:
            continue
        srcs = src[a, js]
        # count[u] = #clones per unique source; delta = 1 if the source is itself dead (a2c.py:453-457)
        cnt = np.bincount(srcs, minlength=N).astype(F)
        delta = dead[a].astype(F)
        with np.errstate(divide="ignore"):      # entries with cnt == 0 are never used
            new_bias_src = (old_bias[a * N:(a + 1) * N]
                            - np.log((cnt + F(1) - delta).astype(F), dtype=F)).astype(F)   # per source col
        # first scatter: bias[tc] = bias_new ; second: bias[c] = bias_new (a2c.py:465-466)
        for sidx in np.unique(srcs):
            bias[a * N + sidx] = new_bias_src[sidx]
        for j, sidx in zip(js, srcs):
            bias[a * N + j] = new_bias_src[sidx]
            sc = F(fixed_sigma) if fixed_sigma else scale[a, sidx]                    # a2c.py:438-439
            base = old_loc[a, sidx]
            if fixed_sigma and tanh_mode:
                base = np.tanh(base, dtype=F)                                         # a2c.py:430-432
            noise = (sc * F(u_noise[a, j])).astype(F)                                 # a2c.py:441
            noise = (noise + (F(-1e-4) if noise < 0 else F(1e-4))).astype(F)          # a2c.py:442-444
            nl = (base + noise).astype(F)                                             # a2c.py:445
            if tanh_mode:   # a2c.py:448-450 (quirk: atanh of a pre-tanh value, and/or precedence)
                nl = np.arctanh(np.clip(nl, F(1e-6 - 1), F(1 - 1e-6)), dtype=F)
            loc[a, j] = nl
            if not fixed_sigma:                                                       # a2c.py:463-465
                logstd[a, j] = np.clip(old_logstd[a, sidx], F(-20), F(2))             # a2c.py:451
            weight[:, a * N + j] = old_weight[:, a * N + sidx]                        # a2c.py:467-469
    return loc, logstd, bias, weight, {"dead": dead, "src": src.astype(np.int32)}


# --------------------------------------------------------------------------- #
# GAE (a2c.py:30-40, utils.py:5-15)                                           #
# --------------------------------------------------------------------------- #
def gae(reward, value, gamma, gae_gamma):
    """generalized_advantage_estimate: td = r + gamma*v' - v on fp32 arrays; reverse scan
    adv_t = td_t + gae_gamma*adv_{t+1} with a float64 accumulator (NumPy<2 scalar promotion
    of `np.float32 + python_float`), emitted as float64 list, cast to fp32 by flat_train
    (distributed_model.py:324-326).  reward [T], value [T+1] -> adv [T] fp32."""
    r = np.asarray(reward, dtype=F)
    v = np.asarray(value[:-1], dtype=F)
    v_ = np.asarray(value[1:], dtype=F)
    td = (r + (F(gamma) * v_).astype(F)).astype(F)
    td = (td - v).astype(F)
    out = np.empty(len(td), np.float64)
    acc = 0.0
    f = float(gae_gamma)
    for t in reversed(range(len(td))):
        acc = float(td[t]) + f * acc
        out[t] = acc
    return out.astype(F)


# --------------------------------------------------------------------------- #
# state normaliser (actor_critic.py:223-244, utils.py:46-68)                  #
# --------------------------------------------------------------------------- #
def state_normalize(s, mean, std, clip=5.0):
    x = ((np.asarray(s, F) - mean.astype(F)) / std.astype(F)).astype(F)
    if clip:
        x = np.maximum(np.minimum(x, F(clip)), F(-clip))
    return x


def state_norm_update(s, mean, std, step):
    """online_normalizer(moving_average=True).update_op, utils.py:60-68.
    d = min(0.9999, (1+step)/(10+step)); population variance over the batch."""
    s = np.asarray(s, F)
    stp = F(step)
    d = np.minimum(F(0.9999), (F(1) + stp) / (F(10) + stp)).astype(F)
    m = s.mean(axis=0, dtype=F)
    var = ((s - m) ** 2).mean(axis=0, dtype=F)
    new_mean = (d * mean + (F(1) - d) * m).astype(F)
    new_std = np.maximum(F(1e-6), d * std + (F(1) - d) * np.sqrt(var, dtype=F)).astype(F)
    return new_mean, new_std


# --------------------------------------------------------------------------- #
# dense layers (ops.py:82-118) and the actor-critic (actor_critic.py:206-218) #
# --------------------------------------------------------------------------- #
def relu6(x):
    return np.minimum(np.maximum(x, F(0)), F(6))


def fc(x, w, b, act):
    y = (x @ w + b).astype(F)
    return relu6(y) if act else y


PARAM_ORDER = ["actor/fc1/weight", "actor/fc1/bias", "actor/fc2/weight", "actor/fc2/bias",
               "actor/samples", "actor/samples_std", "actor/fc_policy/weight", "actor/fc_policy/bias",
               "critic/fc1/weight", "critic/fc1/bias", "critic/fc2/weight", "critic/fc2/bias",
               "critic/fc3/weight", "critic/fc3/bias"]            # saved-graph variable order


def init_params(rng, S, A, N, hidden=(1024, 512), tanh_mode=False, sigma=0.01):
    """14 trainables, truncated-normal(0, 0.01) weights, zero biases (deepmimic_base.py:7-8)."""
    def tn(shape):
        x = rng.standard_normal(shape)
        bad = np.abs(x) > 2
        while bad.any():
            x[bad] = rng.standard_normal(bad.sum())
            bad = np.abs(x) > 2
        return (x * sigma).astype(F)
    h1, h2 = hidden
    loc, logstd = init_particles(A, N, tanh_mode)
    return {
        "actor/fc1/weight": tn((S, h1)), "actor/fc1/bias": np.zeros(h1, F),
        "actor/fc2/weight": tn((h1, h2)), "actor/fc2/bias": np.zeros(h2, F),
        "actor/samples": loc, "actor/samples_std": logstd,
        "actor/fc_policy/weight": tn((h2, A * N)), "actor/fc_policy/bias": np.zeros(A * N, F),
        "critic/fc1/weight": tn((S, h1)), "critic/fc1/bias": np.zeros(h1, F),
        "critic/fc2/weight": tn((h1, h2)), "critic/fc2/bias": np.zeros(h2, F),
        "critic/fc3/weight": tn((h2, 1)), "critic/fc3/bias": np.zeros(1, F),
    }


def actor_forward(P, x, A, N):
    h1 = fc(x, P["actor/fc1/weight"], P["actor/fc1/bias"], True)
    h2 = fc(h1, P["actor/fc2/weight"], P["actor/fc2/bias"], True)
    logits = fc(h2, P["actor/fc_policy/weight"], P["actor/fc_policy/bias"], False)
    return logits.reshape(-1, A, N), (h1, h2)


def critic_forward(P, x):
    c1 = fc(x, P["critic/fc1/weight"], P["critic/fc1/bias"], True)
    c2 = fc(c1, P["critic/fc2/weight"], P["critic/fc2/bias"], True)
    v = fc(c2, P["critic/fc3/weight"], P["critic/fc3/bias"], False)[:, 0]
    return v, (c1, c2)


def relu6_mask(y_out):
    """Relu6Grad's branch: the units whose gradient passes (ops.py:105-117)."""
    return (y_out > 0) & (y_out < 6)


def _fc_bwd(x_in, w, y_out, dy, act, mask=None):
    """`mask` (flip-census tests only): Relu6Grad's branch decisions taken from another implementation instead of from
    y_out -- the arithmetic given the SAME decisions is what those tests compare."""
    if act:
        dy = np.where(relu6_mask(y_out) if mask is None else mask, dy, F(0)).astype(F)   # Relu6Grad
    dw = (x_in.T @ dy).astype(F)
    db = dy.sum(axis=0, dtype=F)
    dx = (dy @ w.T).astype(F)
    return dx, dw, db


# --------------------------------------------------------------------------- #
# PPO loss (ppo.py:14-74, actor_critic.py:132-184)                            #
# --------------------------------------------------------------------------- #
def clip_advantage_bounds(clip_advantage):
    """actor_critic.py:160-168: a scalar c means [-c, c], a pair its sorted ends; None / False: no clipping."""
    if not clip_advantage:
        return None
    if hasattr(clip_advantage, "__len__"):
        return F(min(clip_advantage)), F(max(clip_advantage))
    return F(-clip_advantage), F(clip_advantage)


def ppo_loss(logp, logp_old, adv, v, v_old, eps_clip=0.2, value_loss_coef=0.5, normalize_adv=True,
             clip_advantage=None, grad_on=None):
    """Returns (loss, policy_loss, value_loss*coef, dlogp[B], dv[B]).  `grad_on` [B] bool (flip-census tests only): which
    rows' surrogate passes a gradient, taken from another implementation instead of from this one's ratio."""
    logp = np.asarray(logp, F); adv = np.asarray(adv, F); v = np.asarray(v, F)
    B = F(len(adv))
    if normalize_adv:                                             # actor_critic.py:151-156
        m = adv.mean(dtype=F)
        var = ((adv - m) ** 2).mean(dtype=F)
        nadv = ((adv - m) / (np.sqrt(var, dtype=F) + F(1e-8))).astype(F)
    else:
        nadv = adv
    bounds = clip_advantage_bounds(clip_advantage)               # actor_critic.py:158-170 (value target is NOT clipped)
    if bounds is not None:
        nadv = np.maximum(np.minimum(nadv, bounds[1]), bounds[0])
    with np.errstate(over="ignore", invalid="ignore"):
        ratio = np.exp(logp - np.asarray(logp_old, F), dtype=F)   # ppo.py:47-48
        sur = (ratio * nadv).astype(F)
        lo, hi = F(1.0 - eps_clip), F(1.0 + eps_clip)
        cr = np.maximum(np.minimum(ratio, hi), lo)                # tf.clip_by_value
        csur = (cr * nadv).astype(F)
        policy_loss = -np.minimum(sur, csur).mean(dtype=F)        # ppo.py:54
        target = (adv + np.asarray(v_old, F)).astype(F)           # ppo.py:31-34
        value_loss = ((v - target) ** 2).mean(dtype=F) * F(value_loss_coef)
        # backward.  tf.minimum grad goes to `sur` where sur <= csur, else to csur; the
        # clip_by_value grad passes only where lo <= ratio <= hi (inclusive, TF semantics).
        to_sur = sur <= csur
        inside = (ratio >= lo) & (ratio <= hi)
        dmin = F(-1) / B
        on = (to_sur | inside) if grad_on is None else np.asarray(grad_on, bool)
        dratio = np.where(on, dmin * nadv, F(0)).astype(F)
        dlogp = (dratio * ratio).astype(F)
    dv = (F(2) * (v - target) * F(value_loss_coef) / B).astype(F)
    return F(policy_loss + value_loss), F(policy_loss), F(value_loss), dlogp, dv


# --------------------------------------------------------------------------- #
# optimiser (base_worker.py:97-102, TF ApplyAdam)                             #
# --------------------------------------------------------------------------- #
def clip_by_global_norm(grads, clip_norm=1.0):
    """tf.clip_by_global_norm: norm = sqrt(2*sum_i L2Loss(g_i)); scale = c*min(1/norm, 1/c)."""
    half = [F(0.5) * np.sum(np.square(g.astype(F)), dtype=F) for g in grads]
    norm = np.sqrt(F(2) * np.sum(np.asarray(half, F), dtype=F), dtype=F)
    c = F(clip_norm)
    with np.errstate(divide="ignore"):
        scale = c * np.minimum(F(1) / norm, F(1) / c)
    if not np.isfinite(norm):
        scale = F(np.nan)
    return [(g * scale).astype(F) for g in grads], norm


def tf_adam(p, g, m, v, b1p, b2p, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """TF `ApplyAdam` CPU kernel form: alpha = lr*sqrt(1-b2^t)/(1-b1^t);
    m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); p -= m*alpha/(sqrt(v)+eps).
    b1p/b2p are the fp32 `beta*_power` variables BEFORE this step's update (they start at
    beta and are multiplied by beta after the applies).  Not torch.optim.Adam."""
    lr, b1, b2, eps = F(lr), F(b1), F(b2), F(eps)
    alpha = (lr * np.sqrt(F(1) - F(b2p), dtype=F) / (F(1) - F(b1p))).astype(F)
    m = (m + (g - m) * (F(1) - b1)).astype(F)
    v = (v + (g * g - v) * (F(1) - b2)).astype(F)
    p = (p - (m * alpha) / (np.sqrt(v, dtype=F) + eps)).astype(F)
    return p, m, v


# --------------------------------------------------------------------------- #
# one DPPO learner step                                          #
# --------------------------------------------------------------------------- #
def dppo_grads(P, norm_mean, norm_std, batch, A, N, eps_clip=0.2, value_loss_coef=0.5, force=None):
    """Forward + backward of loss = policy_loss + 0.5*value_loss on one (rank-local) batch.
    batch = dict(state[B,S], action[B,A], value[B], log_prob[B], advantage[B]).
    Returns (losses, grads dict, aux).  `force` (flip-census tests only): branch decisions of the loss's discontinuities
    taken from another implementation -- {"actor/fc1": bool [B,H], "actor/fc2", "critic/fc1", "critic/fc2": Relu6Grad masks,
    "ppo_on": bool [B]}; aux carries this evaluation's own decisions under "branches"."""
    force = force or {}
    x = state_normalize(batch["state"], norm_mean, norm_std, 5.0)
    logits, (h1, h2) = actor_forward(P, x, A, N)
    v, (c1, c2) = critic_forward(P, x)
    logp, _ = mixture_log_prob(logits, batch["action"], P["actor/samples"], P["actor/samples_std"])
    loss, pl, vl, dlogp, dv = ppo_loss(logp, batch["log_prob"], batch["advantage"], v, batch["value"],
                                       eps_clip, value_loss_coef, grad_on=force.get("ppo_on"))
    dlogits, dloc, dlogstd, _ = mixture_log_prob_bwd(logits, batch["action"], P["actor/samples"],
                                                     P["actor/samples_std"], dlogp)
    G = {"actor/samples": dloc, "actor/samples_std": dlogstd}
    dl = dlogits.reshape(len(x), A * N)
    dh2, G["actor/fc_policy/weight"], G["actor/fc_policy/bias"] = _fc_bwd(h2, P["actor/fc_policy/weight"], None, dl, False)
    dh1, G["actor/fc2/weight"], G["actor/fc2/bias"] = _fc_bwd(h1, P["actor/fc2/weight"], h2, dh2, True, force.get("actor/fc2"))
    _, G["actor/fc1/weight"], G["actor/fc1/bias"] = _fc_bwd(x, P["actor/fc1/weight"], h1, dh1, True, force.get("actor/fc1"))
    dc2, G["critic/fc3/weight"], G["critic/fc3/bias"] = _fc_bwd(c2, P["critic/fc3/weight"], None, dv[:, None], False)
    dc1, G["critic/fc2/weight"], G["critic/fc2/bias"] = _fc_bwd(c1, P["critic/fc2/weight"], c2, dc2, True, force.get("critic/fc2"))
    _, G["critic/fc1/weight"], G["critic/fc1/bias"] = _fc_bwd(x, P["critic/fc1/weight"], c1, dc1, True, force.get("critic/fc1"))
    branches = {"actor/fc1": relu6_mask(h1), "actor/fc2": relu6_mask(h2), "critic/fc1": relu6_mask(c1),
                "critic/fc2": relu6_mask(c2), "ppo_on": dlogp != 0}
    return (loss, pl, vl), G, {"logp": logp, "value": v, "logits": logits, "x": x, "branches": branches, "dlogp": dlogp}


class DPPOState:
    """Everything a learner replica owns (variables of the saved graph, Appendix A)."""

    def __init__(self, rng, S=197, A=36, N=35, hidden=(1024, 512), resample_interval=368):
        self.S, self.A, self.N = S, A, N
        self.P = init_params(rng, S, A, N, hidden)
        self.m = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.b1p, self.b2p = F(0.9), F(0.999)
        self.norm_mean, self.norm_std = np.zeros(S, F), np.ones(S, F)
        self.max_active, self.sum_active = np.zeros((A, N), F), np.zeros((A, N), F)
        self.global_step = 0
        self.train_flag = 0
        self.resample_interval = resample_interval


def dppo_learner_step(st, shards, lr=1e-4, clip_norm=1.0, u_resample=None, u_noise=None):
    """One global learner step over W rank-local shards (W=1: a single batch).

    Mirrors sync_model.py:62-101 + SURVEY App. B "State-norm EMA": each worker (a) updates
    its LOCAL state-normaliser copy on its shard, (b) computes grads with the PS statistics of
    the previous apply, (c) clips by global norm PER WORKER, then the PS takes the mean of the
    clipped grads and of the 4 pushed stat tensors, applies TF-Adam, ticks the resampler.
    Returns list of per-shard losses."""
    W = len(shards)
    losses, gsum, stat_sum = [], None, None
    for b in shards:
        lm, ls = state_norm_update(b["state"], st.norm_mean, st.norm_std, st.global_step)
        (loss, pl, vl), G, _ = dppo_grads(st.P, st.norm_mean, st.norm_std, b, st.A, st.N)
        names = PARAM_ORDER
        clipped, _ = clip_by_global_norm([G[n] for n in names], clip_norm)
        losses.append((loss, pl, vl))
        stats = [lm, ls, st.max_active, st.sum_active]
        if gsum is None:
            gsum = [c.astype(F) for c in clipped]; stat_sum = [s.astype(F) for s in stats]
        else:
            gsum = [(a + c).astype(F) for a, c in zip(gsum, clipped)]
            stat_sum = [(a + s).astype(F) for a, s in zip(stat_sum, stats)]
    gmean = [(g / F(W)).astype(F) for g in gsum]
    st.norm_mean, st.norm_std, st.max_active, st.sum_active = [(s / F(W)).astype(F) for s in stat_sum]
    for n, g in zip(PARAM_ORDER, gmean):
        st.P[n], st.m[n], st.v[n] = tf_adam(st.P[n], g, st.m[n], st.v[n], st.b1p, st.b2p, lr)
    st.b1p, st.b2p = F(st.b1p * F(0.9)), F(st.b2p * F(0.999))
    st.global_step += 1
    st.train_flag += 1
    if st.train_flag >= st.resample_interval:                      # a2c.py:370-383
        if u_resample is not None:
            (st.P["actor/samples"], st.P["actor/samples_std"], st.P["actor/fc_policy/bias"],
             st.P["actor/fc_policy/weight"], _) = resample_particles(
                st.P["actor/samples"], st.P["actor/samples_std"], st.P["actor/fc_policy/bias"],
                st.P["actor/fc_policy/weight"], st.max_active, st.sum_active, u_resample, u_noise)
        st.max_active[:] = 0; st.sum_active[:] = 0; st.train_flag = 0
    return losses


# --------------------------------------------------------------------------- #
# minibatch schedule of a training round (models/distributed_model.py:320-352) #
# --------------------------------------------------------------------------- #
def flat_train_schedule(n, batch_size, opt_epochs, rng, on_policy=True, sample_counter=0):
    """Row ids of every learner step of one `flat_train` round, in order.  On-policy (:323-344): ONE id array,
    shuffled in place once per epoch (an epoch permutes the previous epoch's order), cut into consecutive slices of
    `batch_size` (last one ragged; no batch size -> one step on all rows).  Off-policy (:345-351): per step
    `choice(n, batch_size)` (with replacement), `opt_epochs` steps or -- when that is None/0 -- `sample_counter`."""
    out = []
    if on_policy:
        ids = np.arange(n)
        epoch = 0
        while epoch < opt_epochs:
            rng.shuffle(ids)
            if batch_size:
                for s in range(0, n, batch_size):
                    out.append(ids[s:s + batch_size].copy())
            else:
                out.append(ids.copy())
            epoch += 1
    elif n >= batch_size:
        for _ in range(opt_epochs if opt_epochs else sample_counter):
            out.append(rng.choice(n, batch_size))
    return out


# --------------------------------------------------------------------------- #
# SAC-PFPN (networks/actor_critic/sac.py:12-175)                  #
# --------------------------------------------------------------------------- #
SAC_ACTOR = ["actor/fc1/weight", "actor/fc1/bias", "actor/fc2/weight", "actor/fc2/bias",
             "actor/samples", "actor/samples_std", "actor/fc_policy/weight", "actor/fc_policy/bias",
             "alpha/log_alpha"]
SAC_CRITIC = ["critic/%s/fc%d/%s" % (q, i, k) for q in ("q1", "q2") for i in (1, 2, 3) for k in ("weight", "bias")]


def init_sac_params(rng, S, A, N, hidden=(1024, 512), sigma=0.01):
    P = init_params(rng, S, A, N, hidden, tanh_mode=True, sigma=sigma)
    P = {k: v for k, v in P.items() if k.startswith("actor")}
    h1, h2 = hidden

    def tn(shape):
        x = rng.standard_normal(shape)
        bad = np.abs(x) > 2
        while bad.any():
            x[bad] = rng.standard_normal(bad.sum())
            bad = np.abs(x) > 2
        return (x * sigma).astype(F)
    for q in ("q1", "q2"):
        P["critic/%s/fc1/weight" % q] = tn((S + A, h1)); P["critic/%s/fc1/bias" % q] = np.zeros(h1, F)
        P["critic/%s/fc2/weight" % q] = tn((h1, h2)); P["critic/%s/fc2/bias" % q] = np.zeros(h2, F)
        P["critic/%s/fc3/weight" % q] = tn((h2, 1)); P["critic/%s/fc3/bias" % q] = np.zeros(1, F)
    P["alpha/log_alpha"] = np.zeros((), F)
    return P


def q_forward(P, q, xa):
    c1 = fc(xa, P["critic/%s/fc1/weight" % q], P["critic/%s/fc1/bias" % q], True)
    c2 = fc(c1, P["critic/%s/fc2/weight" % q], P["critic/%s/fc2/bias" % q], True)
    v = fc(c2, P["critic/%s/fc3/weight" % q], P["critic/%s/fc3/bias" % q], False)[:, 0]
    return v, (c1, c2)


def _q_backward(P, q, xa, acts, dq, G=None, masks=(None, None)):
    """returns d xa; fills G with parameter grads when G is a dict.  `masks`: forced Relu6Grad decisions of fc1 / fc2."""
    c1, c2 = acts
    dc2, dw3, db3 = _fc_bwd(c2, P["critic/%s/fc3/weight" % q], None, dq[:, None], False)
    dc1, dw2, db2 = _fc_bwd(c1, P["critic/%s/fc2/weight" % q], c2, dc2, True, masks[1])
    dxa, dw1, db1 = _fc_bwd(xa, P["critic/%s/fc1/weight" % q], c1, dc1, True, masks[0])
    if G is not None:
        for i, (dw, db) in zip((1, 2, 3), ((dw1, db1), (dw2, db2), (dw3, db3))):
            G["critic/%s/fc%d/weight" % (q, i)] = dw
            G["critic/%s/fc%d/bias" % (q, i)] = db
    return dxa


def sac_grads(P, PT, norm_mean, norm_std, batch, noise, A, N, gamma=0.95, value_loss_coef=0.5, force=None):
    """One SAC-PFPN forward/backward.  P: eval params (actor + q1/q2 + log_alpha), PT: target q1/q2.
    batch: state, action, reward, not_terminal, state_.  noise: gumbel/eps for the eval draw and
    gumbel_/eps_ for the target draw, each [B,A,N].
    Returns (loss tuple (loss, policy_loss, value_loss*coef), grads dict, aux).  `force` (flip-census tests only): branch
    decisions taken from another implementation -- Relu6Grad masks "actor/fc1", "actor/fc2", "q1/pi/fc1" .. "q2/run/fc2"
    (the live critics on (s, a_pi) and on (s, action_hist)) and "sel1" [B] (which twin the minimum routes its gradient
    to); aux["branches"] = this evaluation's own."""
    force = force or {}
    B = len(batch["state"])
    loc, logstd = P["actor/samples"], P["actor/samples_std"]
    alpha = np.exp(P["alpha/log_alpha"], dtype=F)                          # sac.py:38 (stop-grad)
    x = state_normalize(batch["state"], norm_mean, norm_std, 5.0)
    logits, (h1, h2) = actor_forward(P, x, A, N)
    a, s_, k = rsample(logits, loc, logstd, noise["gumbel"], noise["eps"])
    logp, _ = mixture_log_prob(logits, s_, loc, logstd, tanh_mode=True)    # sac.py:128-130
    xa = np.concatenate([x, a], -1).astype(F)
    q1p, acts1p = q_forward(P, "q1", xa)
    q2p, acts2p = q_forward(P, "q2", xa)
    xh = np.concatenate([x, np.asarray(batch["action"], F)], -1).astype(F)
    q1r, acts1r = q_forward(P, "q1", xh)
    q2r, acts2r = q_forward(P, "q2", xh)
    # target net: shared actor + normaliser, own Q1', Q2' (sac.py:141-153)
    x2 = state_normalize(batch["state_"], norm_mean, norm_std, 5.0)
    logits2, _ = actor_forward(P, x2, A, N)
    a2, s2, _ = rsample(logits2, loc, logstd, noise["gumbel_"], noise["eps_"])
    logp2, _ = mixture_log_prob(logits2, s2, loc, logstd, tanh_mode=True)
    xa2 = np.concatenate([x2, a2], -1).astype(F)
    vf2 = (np.minimum(q_forward(PT, "q1", xa2)[0], q_forward(PT, "q2", xa2)[0]) - alpha * logp2).astype(F)
    q_target = (np.asarray(batch["reward"], F) + F(gamma) * np.asarray(batch["not_terminal"], F) * vf2).astype(F)
    value_loss = ((q_target - q1r) ** 2 + (q_target - q2r) ** 2).mean(dtype=F) * F(value_loss_coef)
    qmin = np.minimum(q1p, q2p)
    target_entropy = F(-A)
    l = (alpha * logp - qmin - P["alpha/log_alpha"] * (logp + target_entropy)).astype(F)
    policy_loss = l.mean(dtype=F)
    G = {}
    invB = F(1) / F(B)
    # critic grads: d value_loss / d q_running
    fm = lambda q, path: (force.get("%s/%s/fc1" % (q, path)), force.get("%s/%s/fc2" % (q, path)))
    _q_backward(P, "q1", xh, acts1r, (F(-2) * (q_target - q1r) * F(value_loss_coef) * invB).astype(F), G, fm("q1", "run"))
    _q_backward(P, "q2", xh, acts2r, (F(-2) * (q_target - q2r) * F(value_loss_coef) * invB).astype(F), G, fm("q2", "run"))
    # actor grads: policy_loss w.r.t. actor variables and log_alpha only (sac.py:166-173)
    G["alpha/log_alpha"] = F(-(logp + target_entropy).mean(dtype=F))
    own_sel1 = q1p <= q2p                                                 # tf.minimum grad routing
    sel1 = own_sel1 if force.get("sel1") is None else np.asarray(force["sel1"], bool)
    dxa = _q_backward(P, "q1", xa, acts1p, np.where(sel1, -invB, F(0)).astype(F), None, fm("q1", "pi")) \
        + _q_backward(P, "q2", xa, acts2p, np.where(sel1, F(0), -invB).astype(F), None, fm("q2", "pi"))
    da = dxa[:, -A:].astype(F)
    dlogp = np.full(B, alpha * invB, F)
    dl1, dloc1, dls1, dxs = mixture_log_prob_bwd(logits, s_, loc, logstd, dlogp, tanh_mode=True)
    dl2, dloc2, dls2 = rsample_bwd(logits, loc, logstd, noise["gumbel"], noise["eps"], da, dxs)
    dlogits = (dl1 + dl2).reshape(B, A * N).astype(F)
    G["actor/samples"] = (dloc1 + dloc2).astype(F)
    G["actor/samples_std"] = (dls1 + dls2).astype(F)
    dh2, G["actor/fc_policy/weight"], G["actor/fc_policy/bias"] = _fc_bwd(h2, P["actor/fc_policy/weight"], None, dlogits, False)
    dh1, G["actor/fc2/weight"], G["actor/fc2/bias"] = _fc_bwd(h1, P["actor/fc2/weight"], h2, dh2, True, force.get("actor/fc2"))
    _, G["actor/fc1/weight"], G["actor/fc1/bias"] = _fc_bwd(x, P["actor/fc1/weight"], h1, dh1, True, force.get("actor/fc1"))
    branches = {"actor/fc1": relu6_mask(h1), "actor/fc2": relu6_mask(h2), "sel1": own_sel1}
    for q, ap, ar in (("q1", acts1p, acts1r), ("q2", acts2p, acts2r)):
        for path, acts in (("pi", ap), ("run", ar)):
            branches["%s/%s/fc1" % (q, path)] = relu6_mask(acts[0])
            branches["%s/%s/fc2" % (q, path)] = relu6_mask(acts[1])
    aux = {"logp": logp, "a": a, "s_": s_, "k": k, "q_target": q_target, "q1r": q1r, "q1p": q1p, "q2p": q2p, "alpha": alpha,
           "branches": branches}
    return (F(policy_loss + value_loss), F(policy_loss), F(value_loss)), G, aux


class SACState:
    def __init__(self, rng, S=197, A=36, N=35, hidden=(1024, 512), resample_interval=12000):
        self.S, self.A, self.N = S, A, N
        self.P = init_sac_params(rng, S, A, N, hidden)
        self.PT = {k: v.copy() for k, v in self.P.items() if k.startswith("critic")}   # hard sync at init
        self.m = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.bp_c = [F(0.9), F(0.999)]
        self.bp_a = [F(0.9), F(0.999)]
        self.norm_mean, self.norm_std = np.zeros(S, F), np.ones(S, F)
        self.max_active, self.sum_active = np.zeros((A, N), F), np.zeros((A, N), F)
        self.global_step = 0
        self.train_flag = 0
        self.resample_interval = resample_interval


def sac_learner_step(st, batch, noise, lr_c=1e-4, lr_a=1e-4, clip_norm=1.0, tau=0.005, gamma=0.95):
    """Single-worker SAC update (workers == 1, deepmimic_sac_base.py:4): grads -> ONE joint global-norm
    clip over critic+actor gradients (base_worker.py:97-102) -> critic Adam then actor Adam
    (base_worker.py:33-45) -> soft target update (sac.py:67-73) -> resample tick."""
    lm, ls = state_norm_update(batch["state"], st.norm_mean, st.norm_std, st.global_step)
    losses, G, aux = sac_grads(st.P, st.PT, st.norm_mean, st.norm_std, batch, noise, st.A, st.N, gamma)
    names = SAC_CRITIC + SAC_ACTOR
    clipped, _ = clip_by_global_norm([np.asarray(G[n], F) for n in names], clip_norm)
    Gc = dict(zip(names, clipped))
    for n in SAC_CRITIC:
        st.P[n], st.m[n], st.v[n] = tf_adam(st.P[n], Gc[n], st.m[n], st.v[n], st.bp_c[0], st.bp_c[1], lr_c)
    st.bp_c = [F(st.bp_c[0] * F(0.9)), F(st.bp_c[1] * F(0.999))]
    for n in SAC_ACTOR:
        st.P[n], st.m[n], st.v[n] = tf_adam(st.P[n], Gc[n], st.m[n], st.v[n], st.bp_a[0], st.bp_a[1], lr_a)
    st.bp_a = [F(st.bp_a[0] * F(0.9)), F(st.bp_a[1] * F(0.999))]
    st.norm_mean, st.norm_std = lm, ls
    for n in st.PT:
        st.PT[n] = ((F(1) - F(tau)) * st.PT[n] + F(tau) * st.P[n]).astype(F)
    st.global_step += 1
    st.train_flag += 1
    if st.train_flag >= st.resample_interval:
        st.max_active[:] = 0; st.sum_active[:] = 0; st.train_flag = 0
    return losses, aux


# --------------------------------------------------------------------------------------- #
# V-trace / A2C (A3C, IMPALA) learner -- SURVEY 8(f)-2                                    #
# --------------------------------------------------------------------------------------- #
# --------------------------------------------------------------------------- #
# TD3 / DDPG-PFPN (networks/actor_critic/ddpg.py:11-243)                      #
# --------------------------------------------------------------------------- #
TD3_ACTOR = ["actor/fc1/weight", "actor/fc1/bias", "actor/fc2/weight", "actor/fc2/bias",
             "actor/samples", "actor/samples_std", "actor/fc_policy/weight", "actor/fc_policy/bias"]
TD3_CRITIC = ["critic/%s/fc%d/%s" % (q, i, k) for q in ("q", "twin_q") for i in (1, 2, 3) for k in ("weight", "bias")]


def init_td3_params(rng, S, A, N, hidden=(1024, 512), sigma=0.01, act_noise=0.1):
    """eval net of ParticleFilteringDDPGNetwork: the PFPN actor on the tanh grid with a FIXED particle width
    (`act_noise` becomes `init_sigma` with `fixed_sigma=True`, ddpg.py:48-50) + the critics `q` and `twin_q` on
    concat(normalised state, action) (ddpg.py:174-197)."""
    P = init_params(rng, S, A, N, hidden, tanh_mode=True, sigma=sigma)
    P = {k: v for k, v in P.items() if k.startswith("actor")}
    P["actor/samples"], P["actor/samples_std"] = init_particles(A, N, tanh_mode=True, init_sigma=act_noise)
    sacP = init_sac_params(rng, S, A, N, hidden, sigma)
    for k, v in sacP.items():
        if k.startswith("critic/q1"):
            P[k.replace("critic/q1", "critic/q")] = v
        elif k.startswith("critic/q2"):
            P[k.replace("critic/q2", "critic/twin_q")] = v
    return P


def deterministic_action_bwd(logits, loc, da):
    """Custom gradient `mask` of MixtureGaussianDistribution.mean(), tanh mode (utils.py:205-221):
    a = tanh(loc)[argmax softmax(logits)];  d a / d w_n = tanh(loc_n) - tanh(loc_k);  d a / d tanh(loc) = onehot(k),
    summed over the batch; then the softmax Jacobian and tanh'.  -> dlogits [B,A,N], dloc [A,N]."""
    logits = np.asarray(logits, F)
    w = softmax(logits)
    k = logits.argmax(axis=-1)
    t = np.tanh(loc, dtype=F)[None]                                             # [1,A,N]
    tk = np.take_along_axis(np.broadcast_to(t, logits.shape), k[..., None], axis=-1)
    gw = ((t - tk) * np.asarray(da, F)[..., None]).astype(F)
    dlogits = (w * (gw - (gw * w).sum(axis=-1, keepdims=True, dtype=F))).astype(F)
    m = np.zeros_like(logits)
    np.put_along_axis(m, k[..., None], F(1), axis=-1)
    dt = (m * np.asarray(da, F)[..., None]).sum(axis=0, dtype=F)                # [A,N]
    dloc = (dt * (F(1) - t[0] * t[0])).astype(F)
    return dlogits, dloc


def td3_grads(P, PT, norm, norm_t, batch, noise_t, A, N, sync, gamma=0.95, value_loss_coef=0.5,
              target_noise=0.2, lb=-1.0, ub=1.0):
    """One TD3-PFPN forward/backward (twin_q=True).  P: eval params, PT: the target net's own copy of EVERY variable
    (actor, critics; its state-normaliser statistics are `norm_t`), ddpg.py:31-36, :201-219.
    target action = clip(mean_target(s') + target_noise * noise_t, lb, ub) (ddpg.py:136-158 with act_noise=target_noise);
    q_target = r + gamma * not_terminal * min(Q', twin_Q')(s', a') (:190-197, :216-219);
    value loss = mean_b((Q - y)^2 + (twin_Q - y)^2) * coef (:221-224, a2c.py:51-56) w.r.t. the critic variables;
    policy loss = -mean(Q(s, mean(s))) w.r.t. the actor variables, and only when `sync` (global_step % policy_delay
    == 0), else its gradient is zero (:199-207).  Returns ((loss, policy_loss, value_loss*coef), grads, aux)."""
    B = len(batch["state"])
    invB = F(1) / F(B)
    loc = P["actor/samples"]
    x = state_normalize(batch["state"], norm[0], norm[1], 5.0)
    logits, (h1, h2) = actor_forward(P, x, A, N)
    k, a_opt = mean_action(logits, loc, tanh_mode=True)
    xa = np.concatenate([x, a_opt], -1).astype(F)
    q_pi, acts_pi = q_forward(P, "q", xa)
    xh = np.concatenate([x, np.asarray(batch["action"], F)], -1).astype(F)
    q1r, acts1r = q_forward(P, "q", xh)
    q2r, acts2r = q_forward(P, "twin_q", xh)
    x2 = state_normalize(batch["state_"], norm_t[0], norm_t[1], 5.0)
    logits2, _ = actor_forward(PT, x2, A, N)
    _, a2 = mean_action(logits2, PT["actor/samples"], tanh_mode=True)
    a2 = np.clip((a2 + F(target_noise) * np.asarray(noise_t, F)).astype(F), F(lb), F(ub))
    xa2 = np.concatenate([x2, a2], -1).astype(F)
    v2 = np.minimum(q_forward(PT, "q", xa2)[0], q_forward(PT, "twin_q", xa2)[0])
    q_target = (np.asarray(batch["reward"], F) + F(gamma) * np.asarray(batch["not_terminal"], F) * v2).astype(F)
    value_loss = ((q1r - q_target) ** 2 + (q2r - q_target) ** 2).mean(dtype=F) * F(value_loss_coef)
    policy_loss = -q_pi.mean(dtype=F)
    G = {}
    _q_backward(P, "q", xh, acts1r, (F(2) * (q1r - q_target) * F(value_loss_coef) * invB).astype(F), G)
    _q_backward(P, "twin_q", xh, acts2r, (F(2) * (q2r - q_target) * F(value_loss_coef) * invB).astype(F), G)
    if sync:
        dxa = _q_backward(P, "q", xa, acts_pi, np.full(B, -invB, F))
        dlogits, dloc = deterministic_action_bwd(logits, loc, dxa[:, -A:])
        dlogits = dlogits.reshape(B, A * N)
        G["actor/samples"] = dloc
        dh2, G["actor/fc_policy/weight"], G["actor/fc_policy/bias"] = _fc_bwd(h2, P["actor/fc_policy/weight"], None, dlogits, False)
        dh1, G["actor/fc2/weight"], G["actor/fc2/bias"] = _fc_bwd(h1, P["actor/fc2/weight"], h2, dh2, True)
        _, G["actor/fc1/weight"], G["actor/fc1/bias"] = _fc_bwd(x, P["actor/fc1/weight"], h1, dh1, True)
    else:
        for n in TD3_ACTOR:
            G[n] = np.zeros_like(P[n])
    G["actor/samples_std"] = np.zeros_like(P["actor/samples_std"])        # fixed_sigma: not a trainable variable
    aux = {"a_opt": a_opt, "k": k, "q_target": q_target, "q_pi": q_pi, "a2": a2}
    return (F(policy_loss + value_loss), F(policy_loss), F(value_loss)), G, aux


class TD3State:
    def __init__(self, rng, S=197, A=36, N=35, hidden=(1024, 512), policy_delay=2, act_noise=0.1):
        self.S, self.A, self.N, self.policy_delay = S, A, N, policy_delay
        self.P = init_td3_params(rng, S, A, N, hidden, act_noise=act_noise)
        self.PT = {k: v.copy() for k, v in self.P.items()}                   # hard sync at init (ddpg.py:110-121)
        self.m = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.bp_c, self.bp_a = [F(0.9), F(0.999)], [F(0.9), F(0.999)]
        self.norm_mean, self.norm_std = np.zeros(S, F), np.ones(S, F)
        self.norm_mean_t, self.norm_std_t = np.zeros(S, F), np.ones(S, F)
        self.global_step = 0


def td3_learner_step(st, batch, noise_t, lr_c=1e-4, lr_a=1e-4, clip_norm=1.0, tau=0.005, gamma=0.95):
    """Single-worker TD3-PFPN update: gradients (policy gradient only on sync steps), ONE joint global-norm clip, critic
    Adam then actor Adam (base_worker.py:33-45, :97-118), normaliser EMA, then on sync steps the soft update of EVERY
    target variable incl. the normaliser statistics (ddpg.py:97-109).  `sync` uses the step counter BEFORE this apply:
    the tensor is evaluated once per run and the loss needs it before the optimiser increments the counter."""
    sync = st.global_step % st.policy_delay == 0 if st.policy_delay > 1 else True
    losses, G, aux = td3_grads(st.P, st.PT, (st.norm_mean, st.norm_std), (st.norm_mean_t, st.norm_std_t), batch, noise_t,
                               st.A, st.N, sync, gamma)
    trainable_a = [n for n in TD3_ACTOR if n != "actor/samples_std"]
    names = TD3_CRITIC + trainable_a
    clipped, _ = clip_by_global_norm([np.asarray(G[n], F) for n in names], clip_norm)
    new_mean, new_std = state_norm_update(batch["state"], st.norm_mean, st.norm_std, st.global_step)
    for n, g in zip(names, clipped):
        bp, lr = (st.bp_c, lr_c) if n in TD3_CRITIC else (st.bp_a, lr_a)
        st.P[n], st.m[n], st.v[n] = tf_adam(st.P[n], g.reshape(st.P[n].shape), st.m[n], st.v[n], bp[0], bp[1], lr)
    st.bp_c = [F(st.bp_c[0] * F(0.9)), F(st.bp_c[1] * F(0.999))]
    st.bp_a = [F(st.bp_a[0] * F(0.9)), F(st.bp_a[1] * F(0.999))]
    st.norm_mean, st.norm_std = new_mean, new_std
    st.global_step += 1
    if sync:
        t = F(tau)
        for n in st.PT:
            st.PT[n] = ((F(1) - t) * st.PT[n] + t * st.P[n]).astype(F)
        st.norm_mean_t = ((F(1) - t) * st.norm_mean_t + t * st.norm_mean).astype(F)
        st.norm_std_t = ((F(1) - t) * st.norm_std_t + t * st.norm_std).astype(F)
    return losses, aux


def vtrace(logp_target, logp_behavior, value, reward, not_terminal, E, gamma, rho_clip=1.0, pg_rho_clip=1.0,
           seq_len=None):
    """reference networks/actor_critic/vtrace.py:101-180.  Inputs flat [E*T] in episode-major order
    (to_batches :56-61 reshapes to [E,T] and transposes to time-major [T,E]); fp32 throughout, the reverse
    tf.scan (:143-153) restated as a loop.  Returns (vs, pg_adv, mask), flat in the input order (to_flat)."""
    tb = lambda a: np.asarray(a, F).reshape(E, -1).T.copy()          # [T, E]
    lt, lb, v, r, nt = tb(logp_target), tb(logp_behavior), tb(value), tb(reward), tb(not_terminal)
    T = v.shape[0]
    rho = np.exp((lt - lb).astype(F)).astype(F)                                   # :117-119
    v_rho = rho if rho_clip is None else np.minimum(F(rho_clip), rho)             # :121-126
    c = np.minimum(F(1.0), rho)                                                   # :134
    disc = (F(gamma) * nt).astype(F)                                              # :113-115
    end_mask = np.zeros((T, E), bool)
    if seq_len is not None:                                                       # :68-76 batched_episodic_end_mask
        for e in range(E):
            end_mask[int(seq_len[e]) - 1, e] = True

    def concat_bootstrap(x, val):                                                 # :91-97
        x1 = np.concatenate((x[1:], val[-1:]), axis=0)
        return np.where(end_mask, val, x1) if seq_len is not None else x1

    v_t1 = concat_bootstrap(v, v)                                                 # :127-128
    td = (v_rho * ((r + (disc * v_t1).astype(F)).astype(F) - v).astype(F)).astype(F)     # :129-132
    dsc = np.where(end_mask, F(0), disc).astype(F) if seq_len is not None else disc      # :136-141
    a = np.zeros(E, F)
    vmv = np.zeros((T, E), F)
    for t in range(T - 1, -1, -1):                                                # :143-153 reverse scan
        a = (td[t] + ((dsc[t] * c[t]).astype(F) * a).astype(F)).astype(F)
        vmv[t] = a
    vs = (vmv + v).astype(F)                                                      # :154
    pg_rho = rho if pg_rho_clip is None else np.minimum(F(pg_rho_clip), rho)      # :165-171
    vs_t1 = concat_bootstrap(vs, v)                                               # :172-173
    pg = (pg_rho * ((r + (disc * vs_t1).astype(F)).astype(F) - v).astype(F)).astype(F)   # :174-178
    mask = np.ones((T, E), F)
    if seq_len is not None:                                                       # :77-78 sequence_mask
        mask = (np.arange(T)[:, None] < np.asarray(seq_len)[None, :]).astype(F)
    flat = lambda x: np.ascontiguousarray(x.T).reshape(-1)                        # to_flat :63-64
    return flat(vs), flat(pg), flat(mask)


def a2c_loss(logp, adv, v, v_target, value_loss_coef=0.5, mask=None, normalize_adv=False, clip_advantage=None):
    """reference a2c.py:53-65 + actor_critic.py:132-184: policy_loss = -mean(valid(logp*adv_n)),
    value_loss = coef*mean(valid((v - vt)^2)).  Returns (loss, policy_loss, value_loss*coef, dlogp, dv)."""
    m = np.ones_like(logp, F) if mask is None else np.asarray(mask, F)
    n = F(m.sum())
    adv = np.asarray(adv, F)
    if normalize_adv:
        va = adv[m > 0]
        adv = ((adv - va.mean(dtype=F)) / (np.sqrt(va.var(dtype=F)) + F(1e-8))).astype(F)
    bounds = clip_advantage_bounds(clip_advantage)
    if bounds is not None:
        adv = np.maximum(np.minimum(adv, bounds[1]), bounds[0])
    pl = -F((m * (logp * adv)).sum(dtype=F) / n)
    d = (v - v_target).astype(F)
    vl = F((m * d * d).sum(dtype=F) / n) * F(value_loss_coef)
    dlogp = (-(adv * m) / n).astype(F)
    dv = (F(2.0 * value_loss_coef) * (d * m) / n).astype(F)
    return F(pl + vl), pl, vl, dlogp, dv


def a2c_grads(P, norm_mean, norm_std, batch, A, N, value_loss_coef=0.5, entropy_beta=0.0, vtrace_opts=None):
    """Forward + backward of the A2C loss (A3C: batch has value_target/advantage; IMPALA: vtrace_opts =
    dict(E, gamma, rho_clip, pg_rho_clip, seq_len) and batch has reward/log_prob/not_terminal -- the targets
    come from V-trace on the CURRENT value and target log-prob, stop-gradient: vtrace.py:101-180)."""
    x = state_normalize(batch["state"], norm_mean, norm_std, 5.0)
    logits, (h1, h2) = actor_forward(P, x, A, N)
    v, (c1, c2) = critic_forward(P, x)
    logp, _ = mixture_log_prob(logits, batch["action"], P["actor/samples"], P["actor/samples_std"])
    mask = None
    if vtrace_opts is not None:
        vt, adv, mask = vtrace(logp, batch["log_prob"], v, batch["reward"], batch["not_terminal"], vtrace_opts["E"],
                               vtrace_opts["gamma"], vtrace_opts.get("rho_clip", 1.0),
                               vtrace_opts.get("pg_rho_clip", 1.0), vtrace_opts.get("seq_len"))
        if vtrace_opts.get("seq_len") is None:
            mask = None
    else:
        vt, adv = batch["value_target"], batch["advantage"]
    loss, pl, vl, dlogp, dv = a2c_loss(logp, adv, v, vt, value_loss_coef, mask)
    dlogits, dloc, dlogstd, _ = mixture_log_prob_bwd(logits, batch["action"], P["actor/samples"],
                                                     P["actor/samples_std"], dlogp)
    ent = None
    if entropy_beta:
        m = np.ones(len(x), F) if mask is None else mask
        n = F(m.sum())
        H = categorical_entropy(logits)                               # [B,A]  (a2c.py:62-65: sum over A, mean over valid)
        ent = F((m * H.sum(1, dtype=F)).sum(dtype=F) / n)
        pl = F(pl - F(entropy_beta) * ent)
        loss = F(loss - F(entropy_beta) * ent)
        dH = np.broadcast_to((-(F(entropy_beta)) * m / n)[:, None], H.shape).astype(F)
        dlogits = (dlogits + categorical_entropy_bwd(logits, dH)).astype(F)
    G = {"actor/samples": dloc, "actor/samples_std": dlogstd}
    dl = dlogits.reshape(len(x), A * N)
    dh2, G["actor/fc_policy/weight"], G["actor/fc_policy/bias"] = _fc_bwd(h2, P["actor/fc_policy/weight"], None, dl, False)
    dh1, G["actor/fc2/weight"], G["actor/fc2/bias"] = _fc_bwd(h1, P["actor/fc2/weight"], h2, dh2, True)
    _, G["actor/fc1/weight"], G["actor/fc1/bias"] = _fc_bwd(x, P["actor/fc1/weight"], h1, dh1, True)
    dc2, G["critic/fc3/weight"], G["critic/fc3/bias"] = _fc_bwd(c2, P["critic/fc3/weight"], None, dv[:, None], False)
    dc1, G["critic/fc2/weight"], G["critic/fc2/bias"] = _fc_bwd(c1, P["critic/fc2/weight"], c2, dc2, True)
    _, G["critic/fc1/weight"], G["critic/fc1/bias"] = _fc_bwd(x, P["critic/fc1/weight"], c1, dc1, True)
    return (loss, ent, pl, vl), G, {"logp": logp, "value": v, "value_target": vt, "advantage": adv, "mask": mask}


def a2c_learner_step(st, batch, lr=1e-4, clip_norm=0.5, value_loss_coef=0.5, entropy_beta=0.00025, vtrace_opts=None):
    """One single-replica learner step of the A3C / IMPALA PFPN settings (deepmimic_a3c_base.py,
    deepmimic_impala_base.py: norm_clip 0.5, entropy_beta 0.00025; state-norm EMA as in the DPPO step)."""
    lm, ls = state_norm_update(batch["state"], st.norm_mean, st.norm_std, st.global_step)
    losses, G, aux = a2c_grads(st.P, st.norm_mean, st.norm_std, batch, st.A, st.N, value_loss_coef, entropy_beta,
                               vtrace_opts)
    clipped, _ = clip_by_global_norm([G[n] for n in PARAM_ORDER], clip_norm)
    st.norm_mean, st.norm_std = lm, ls
    for n, g in zip(PARAM_ORDER, clipped):
        st.P[n], st.m[n], st.v[n] = tf_adam(st.P[n], g.astype(F), st.m[n], st.v[n], st.b1p, st.b2p, lr)
    st.b1p, st.b2p = F(st.b1p * F(0.9)), F(st.b2p * F(0.999))
    st.global_step += 1
    st.train_flag += 1
    return losses, aux
