"""CPU checks of the oracle against independent float64 torch-autograd formulations
(the oracle is op-for-op fp32; these pin its math, not its rounding)."""
import numpy as np
import pytest
import torch


def _inputs(oracle, B=13, A=6, N=35, seed=0, tanh_mode=False):
    rng = np.random.default_rng(seed)
    loc, logstd = oracle.init_particles(A, N, tanh_mode)
    loc = (loc + rng.normal(0, 0.05, loc.shape)).astype(np.float32)
    logstd = (logstd + rng.normal(0, 0.1, logstd.shape)).astype(np.float32)
    logits = rng.normal(0, 1, (B, A, N)).astype(np.float32)
    x = rng.uniform(-1.2, 1.2, (B, A)).astype(np.float32)
    return logits, x, loc, logstd


def _torch_logp(logits, x, loc, logstd, tanh_mode):
    lw = torch.log_softmax(logits, -1)
    comp = torch.distributions.Normal(loc, logstd.exp()).log_prob(x.unsqueeze(-1))
    lp = torch.logsumexp(lw + comp, -1)
    if tanh_mode:
        lp = lp - torch.log1p(-torch.tanh(x) ** 2)
    return lp.sum(-1)


@pytest.mark.parametrize("tanh_mode", [False, True])
def test_log_prob_matches_logsumexp(oracle, tanh_mode):
    logits, x, loc, logstd = _inputs(oracle, tanh_mode=tanh_mode)
    lp, p = oracle.mixture_log_prob(logits, x, loc, logstd, tanh_mode)
    t = [torch.tensor(a, dtype=torch.float64) for a in (logits, x, loc, logstd)]
    ref = _torch_logp(*t, tanh_mode).numpy()
    np.testing.assert_allclose(lp, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("tanh_mode", [False, True])
def test_log_prob_bwd_matches_autograd(oracle, tanh_mode):
    logits, x, loc, logstd = _inputs(oracle, seed=3, tanh_mode=tanh_mode)
    g = np.random.default_rng(1).normal(0, 1, len(x)).astype(np.float32)
    dl, dloc, dls, dx = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, g, tanh_mode)
    t = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (logits, x, loc, logstd)]
    (_torch_logp(*t, tanh_mode) * torch.tensor(g, dtype=torch.float64)).sum().backward()
    for got, ref in zip((dl, dx, dloc, dls), (t[0].grad, t[1].grad, t[2].grad, t[3].grad)):
        ref = ref.numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max())


def test_entropy_and_bwd(oracle):
    logits, *_ = _inputs(oracle, seed=5)
    H = oracle.categorical_entropy(logits)
    t = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    ref = torch.distributions.Categorical(logits=t).entropy()
    np.testing.assert_allclose(H, ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    g = np.random.default_rng(2).normal(0, 1, H.shape).astype(np.float32)
    (ref * torch.tensor(g, dtype=torch.float64)).sum().backward()
    np.testing.assert_allclose(oracle.categorical_entropy_bwd(logits, g), t.grad.numpy(), rtol=1e-4, atol=1e-6)


def test_multinomial_rule(oracle):
    """First index whose running fp64 sum of exp(l - max) exceeds u * total."""
    rng = np.random.default_rng(0)
    logits = rng.normal(0, 2, (500, 35)).astype(np.float32)
    u = rng.random(500)
    k = oracle.multinomial_index(logits, u)
    w = np.exp(logits.astype(np.float64) - logits.max(-1, keepdims=True))
    cdf = np.cumsum(w, -1)
    for i in range(500):
        exp_k = int(np.searchsorted(cdf[i], u[i] * cdf[i, -1], side="right"))
        assert k[i] == min(exp_k, 34)
    # edge: u = 0 picks the first particle with non-zero mass; u -> 1 picks the last
    assert oracle.multinomial_index(logits[:1], np.array([0.0]))[0] == 0
    assert oracle.multinomial_index(logits[:1], np.array([np.nextafter(1.0, 0)]))[0] == 34
    # several draws per row
    k2 = oracle.multinomial_index(logits[:4], rng.random((4, 35)))
    assert k2.shape == (4, 35) and k2.dtype == np.int32
    # empirical frequencies follow softmax
    l1 = np.tile(logits[:1], (20000, 1))
    kk = oracle.multinomial_index(l1, rng.random(20000))
    freq = np.bincount(kk, minlength=35) / 20000
    np.testing.assert_allclose(freq, oracle.softmax(logits[0]), atol=0.012)


def test_rsample_bwd_matches_custom_grads(oracle):
    """utils.py:156-186: d a/d w = tanh p - tanh p_k ; d s_/d w = that / max(1e-6, 1 - tanh^2 p_k);
    d/d p = onehot; then softmax(logits + g) and p = loc + scale * eps via autograd."""
    logits, _, loc, logstd = _inputs(oracle, B=7, A=4, N=35, seed=7, tanh_mode=True)
    rng = np.random.default_rng(4)
    gum = -np.log(-np.log(rng.uniform(1e-9, 1, logits.shape))).astype(np.float32)
    eps = rng.normal(0, 1, logits.shape).astype(np.float32)
    da = rng.normal(0, 1, logits.shape[:2]).astype(np.float32)
    ds = rng.normal(0, 1, logits.shape[:2]).astype(np.float32)
    a, s_, k = oracle.rsample(logits, loc, logstd, gum, eps)
    dl, dloc, dls = oracle.rsample_bwd(logits, loc, logstd, gum, eps, da, ds)

    T = lambda v, g=False: torch.tensor(v, dtype=torch.float64, requires_grad=g)
    tl, tloc, tls = T(logits, True), T(loc, True), T(logstd, True)
    w = torch.softmax(tl + T(gum), -1)
    p = tloc + tls.exp() * T(eps)
    kk = torch.tensor(k.astype(np.int64))
    m = torch.nn.functional.one_hot(kk, logits.shape[-1]).double()
    tp = torch.tanh(p)
    tk = (m * tp).sum(-1, keepdim=True).detach()

    class Mask(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w, p, gap):
            ctx.save_for_backward(gap)
            return m * p
        @staticmethod
        def backward(ctx, dy):
            (gap,) = ctx.saved_tensors
            return gap * dy, m * dy, None
    s_t = Mask.apply(w, p, ((tp - tk) / torch.clamp(1 - tk ** 2, min=1e-6)).detach()).sum(-1)
    a_t = Mask.apply(w, tp, (tp - tk).detach()).sum(-1)
    np.testing.assert_allclose(a, a_t.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(s_, s_t.detach().numpy(), rtol=1e-5, atol=1e-6)
    ((a_t * T(da)).sum() + (s_t * T(ds)).sum()).backward()
    for got, ref in ((dl, tl.grad), (dloc, tloc.grad), (dls, tls.grad)):
        ref = ref.numpy()
        np.testing.assert_allclose(got, ref, rtol=3e-4, atol=3e-5 * (np.abs(ref).max() + 1e-30))


def test_gae_closed_form(oracle):
    rng = np.random.default_rng(0)
    T_ = 37
    r = rng.normal(0, 1, T_).astype(np.float32)
    v = rng.normal(0, 1, T_ + 1).astype(np.float32)
    adv = oracle.gae(r, v, 0.95, 0.95 * 0.95)
    td = r.astype(np.float64) + 0.95 * v[1:] - v[:-1]
    ref = np.array([sum(td[t + i] * 0.9025 ** i for i in range(T_ - t)) for t in range(T_)])
    np.testing.assert_allclose(adv, ref, rtol=1e-5, atol=1e-6)
    assert adv.dtype == np.float32


def test_ppo_loss_bwd_matches_autograd(oracle):
    rng = np.random.default_rng(3)
    B = 64
    lp = rng.normal(-30, 1, B).astype(np.float32)
    lp_old = (lp + rng.normal(0, 0.2, B)).astype(np.float32)
    adv = rng.normal(0, 1, B).astype(np.float32)
    v, v_old = rng.normal(0, 1, B).astype(np.float32), rng.normal(0, 1, B).astype(np.float32)
    loss, pl, vl, dlogp, dv = oracle.ppo_loss(lp, lp_old, adv, v, v_old)
    T = lambda a, g=False: torch.tensor(a, dtype=torch.float64, requires_grad=g)
    tlp, tv = T(lp, True), T(v, True)
    ta = T(adv)
    na = (ta - ta.mean()) / (ta.var(unbiased=False).sqrt() + 1e-8)
    ratio = (tlp - T(lp_old)).exp()
    l_pi = -torch.minimum(ratio * na, ratio.clamp(0.8, 1.2) * na).mean()
    l_v = 0.5 * ((tv - (ta + T(v_old))) ** 2).mean()
    (l_pi + l_v).backward()
    np.testing.assert_allclose([pl, vl, loss], [l_pi.item(), l_v.item(), (l_pi + l_v).item()], rtol=1e-5)
    np.testing.assert_allclose(dlogp, tlp.grad.numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(dv, tv.grad.numpy(), rtol=1e-5, atol=1e-9)


def test_tf_adam_differs_from_torch_adam_in_eps_placement(oracle):
    """TF: p -= lr_t * m / (sqrt(v) + eps);  torch: p -= lr * m_hat / (sqrt(v_hat) + eps)."""
    rng = np.random.default_rng(0)
    p = rng.normal(0, 1, 50).astype(np.float32)
    g = (rng.normal(0, 1, 50) * 1e-6).astype(np.float32)      # tiny grads: eps placement matters
    p1, m1, v1 = oracle.tf_adam(p, g, np.zeros_like(p), np.zeros_like(p), 0.9, 0.999)
    m = 0.1 * g.astype(np.float64); v = 0.001 * g.astype(np.float64) ** 2
    alpha = 1e-4 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(p1, p - alpha * m / (np.sqrt(v) + 1e-8), rtol=1e-6, atol=1e-9)
    tp = torch.tensor(p.copy(), requires_grad=True)
    opt = torch.optim.Adam([tp], lr=1e-4, eps=1e-8)
    tp.grad = torch.tensor(g)
    opt.step()
    assert np.abs(tp.detach().numpy() - p1).max() > 1e-6       # genuinely different update


def test_clip_by_global_norm(oracle):
    rng = np.random.default_rng(1)
    gs = [rng.normal(0, 1, s).astype(np.float32) for s in ((5, 3), (7,), (2, 2))]
    out, norm = oracle.clip_by_global_norm(gs, 1.0)
    ref = np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in gs))
    np.testing.assert_allclose(norm, ref, rtol=1e-6)
    np.testing.assert_allclose(np.sqrt(sum((o.astype(np.float64) ** 2).sum() for o in out)), 1.0, rtol=1e-5)
    small = [g * np.float32(1e-3) for g in gs]
    out2, _ = oracle.clip_by_global_norm(small, 1.0)
    for a, b in zip(out2, small):
        assert np.array_equal(a, b)                              # scale == 1 exactly


def test_sac_grads_match_autograd(oracle):
    """oracle.sac_grads (hand-written backward) vs float64 torch autograd of the same graph with the
    reference's custom gradients (utils.py:165-183) and stop-gradients (sac.py:36-39, 123-126, 137-139)."""
    rng = np.random.default_rng(5)
    S, A, N, B = 11, 3, 35, 9
    st = oracle.SACState(rng, S, A, N, hidden=(16, 8))
    for k in st.P:                       # larger weights so every path carries signal
        if k.endswith("weight"):
            st.P[k] = (st.P[k] * 30).astype(np.float32)
    st.PT = {k: (v * 1.1).astype(np.float32) for k, v in st.P.items() if k.startswith("critic")}
    st.P["alpha/log_alpha"] = np.float32(-0.3)
    batch = {"state": rng.normal(0, 1, (B, S)).astype(np.float32), "action": np.tanh(rng.normal(0, 1, (B, A))).astype(np.float32),
             "reward": rng.random(B).astype(np.float32), "not_terminal": (rng.random(B) < 0.8).astype(np.float32),
             "state_": rng.normal(0, 1, (B, S)).astype(np.float32)}
    noise = {k: (-np.log(-np.log(rng.uniform(1e-9, 1, (B, A, N)))) if k.startswith("g") else rng.normal(0, 1, (B, A, N))).astype(np.float32)
             for k in ("gumbel", "eps", "gumbel_", "eps_")}
    (loss, pl, vl), G, aux = oracle.sac_grads(st.P, st.PT, st.norm_mean, st.norm_std, batch, noise, A, N)

    T = lambda v, g=False: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=g)
    P = {k: T(v, True) for k, v in st.P.items()}
    PT = {k: T(v) for k, v in st.PT.items()}
    relu6 = lambda t: torch.clamp(t, 0, 6)

    def actor(x):
        h = relu6(x @ P["actor/fc1/weight"] + P["actor/fc1/bias"])
        h = relu6(h @ P["actor/fc2/weight"] + P["actor/fc2/bias"])
        return (h @ P["actor/fc_policy/weight"] + P["actor/fc_policy/bias"]).view(-1, A, N)

    def qnet(Pd, q, xa):
        h = relu6(xa @ Pd["critic/%s/fc1/weight" % q] + Pd["critic/%s/fc1/bias" % q])
        h = relu6(h @ Pd["critic/%s/fc2/weight" % q] + Pd["critic/%s/fc2/bias" % q])
        return (h @ Pd["critic/%s/fc3/weight" % q] + Pd["critic/%s/fc3/bias" % q])[:, 0]

    def rsample(logits, gum, eps):
        w = torch.softmax(logits + T(gum), -1)
        p = P["actor/samples"] + P["actor/samples_std"].exp() * T(eps)
        k = (logits + T(gum)).argmax(-1)
        m = torch.nn.functional.one_hot(k, N).double()
        tp = torch.tanh(p)
        tk = (m * tp).sum(-1, keepdim=True).detach()

        class Mask(torch.autograd.Function):
            @staticmethod
            def forward(ctx, w, p, gap):
                ctx.save_for_backward(gap)
                return m * p
            @staticmethod
            def backward(ctx, dy):
                return ctx.saved_tensors[0] * dy, m * dy, None
        s_ = Mask.apply(w, p, ((tp - tk) / torch.clamp(1 - tk ** 2, min=1e-6)).detach()).sum(-1)
        a = Mask.apply(w, tp, (tp - tk).detach()).sum(-1)
        return a, s_

    def logprob(logits, s_):
        lw = torch.log_softmax(logits, -1)
        comp = torch.distributions.Normal(P["actor/samples"], P["actor/samples_std"].exp()).log_prob(s_.unsqueeze(-1))
        return (torch.logsumexp(lw + comp, -1) - torch.log1p(-torch.tanh(s_) ** 2)).sum(-1)

    x, x2 = T(batch["state"]).clamp(-5, 5), T(batch["state_"]).clamp(-5, 5)     # mean 0, std 1 normaliser
    alpha = P["alpha/log_alpha"].exp().detach()
    logits = actor(x)
    a, s_ = rsample(logits, noise["gumbel"], noise["eps"])
    logp = logprob(logits, s_)
    q1p, q2p = qnet(P, "q1", torch.cat([x, a], -1)), qnet(P, "q2", torch.cat([x, a], -1))
    xh = torch.cat([x, T(batch["action"])], -1)
    q1r, q2r = qnet(P, "q1", xh), qnet(P, "q2", xh)
    with torch.no_grad():
        l2 = actor(x2)
        a2, s2 = rsample(l2, noise["gumbel_"], noise["eps_"])
        vf2 = torch.minimum(qnet(PT, "q1", torch.cat([x2, a2], -1)), qnet(PT, "q2", torch.cat([x2, a2], -1))) - alpha * logprob(l2, s2)
        qt = T(batch["reward"]) + 0.95 * T(batch["not_terminal"]) * vf2
    value_loss = 0.5 * ((qt - q1r) ** 2 + (qt - q2r) ** 2).mean()
    policy_loss = (alpha * logp - torch.minimum(q1p, q2p) - P["alpha/log_alpha"] * (logp - A).detach()).mean()
    np.testing.assert_allclose([pl, vl], [policy_loss.item(), value_loss.item()], rtol=2e-4, atol=1e-5)
    crit = [k for k in P if k.startswith("critic")]
    act = [k for k in P if not k.startswith("critic")]
    gc = torch.autograd.grad(value_loss, [P[k] for k in crit], retain_graph=True, allow_unused=True)
    ga = torch.autograd.grad(policy_loss, [P[k] for k in act], allow_unused=True)
    for k, g in list(zip(crit, gc)) + list(zip(act, ga)):
        ref = g.numpy()
        got = np.asarray(G[k]).reshape(ref.shape)
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-4 * (np.abs(ref).max() + 1e-12), err_msg=k)


def test_resample_topk_sources_and_mass_split():
    """oracle `resample = k` branch (a2c.py:405-408): sources lie in the top-k set, ties go to the lower index, and the
    bias split keeps the softmax mass of {source + clones} equal to the source's old mass."""
    import oracle.pfpn_oracle as o
    rng = np.random.default_rng(3)
    A, N, H, k = 4, 12, 8, 3
    loc, logstd = o.init_particles(A, N, False)
    bias = rng.normal(0, 0.5, A * N).astype(np.float32)
    weight = rng.normal(0, 0.1, (H, A * N)).astype(np.float32)
    sum_active = (rng.random((A, N)) * 10 + 1).astype(np.float32)
    sum_active[0, 2] = sum_active[0, 7] = 100.0                      # tie at the top of dim 0
    max_active = np.ones((A, N), np.float32)
    max_active[:, 9:] = 0.0                                          # three dead particles per dim
    sum_active[:, 9:] = 0.01
    u = rng.random((A, N)); un = rng.uniform(-1, 1, (A, N)).astype(np.float32)
    u[0, 9], u[0, 10] = 0.0, 0.34                                    # candidate 0 and candidate 1 of dim 0
    nl, nls, nb, nw, info = o.resample_particles(loc, logstd, bias, weight, max_active, sum_active, u, un, topk=k)
    src, dead = info["src"], info["dead"]
    assert src[0, 9] == 2 and src[0, 10] == 7                        # tie broken towards the lower index
    for a in range(A):
        top = set(np.argsort(-sum_active[a], kind="stable")[:k].tolist())
        assert set(src[a, 9:].tolist()) <= top
        for s in set(src[a, 9:].tolist()):
            grp = [s] + [j for j in range(9, N) if src[a, j] == s]
            new_mass = np.exp(nb[a * N:(a + 1) * N][grp].astype(np.float64)).sum()
            np.testing.assert_allclose(new_mass, np.exp(np.float64(bias[a * N + s])), rtol=1e-5)
    assert np.array_equal(nw[:, 9], weight[:, src[0, 9]])


def test_torch_cpu_step_is_the_oracle_step(oracle):
    """bench.py's all-core CPU baseline (oracle/torch_cpu_step.py: torch-CPU ops + autograd) computes the same learner
    step as the op-for-op NumPy oracle: losses, every gradient, parameters after two steps."""
    import torch
    from oracle import torch_cpu_step as tcs
    rng = np.random.default_rng(3)
    st = oracle.DPPOState(rng)
    for k in ("actor/fc1/weight", "actor/fc2/weight", "actor/fc_policy/weight", "critic/fc1/weight", "critic/fc2/weight"):
        st.P[k] = (st.P[k] * 8).astype(np.float32)          # off the near-linear regime of the 0.01-sigma init
    B = 64
    ts = tcs.TorchCPUState(st)
    for it in range(2):
        b = {"state": (rng.normal(0, 1, (B, 197)) * 2 + 0.3).astype(np.float32),
             "action": rng.uniform(-1.1, 1.1, (B, 36)).astype(np.float32),
             "value": rng.normal(0, 0.1, B).astype(np.float32), "advantage": rng.normal(0, 1, B).astype(np.float32)}
        x = oracle.state_normalize(b["state"], st.norm_mean, st.norm_std, 5.0)
        logits, _ = oracle.actor_forward(st.P, x, 36, 35)
        lp, _ = oracle.mixture_log_prob(logits, b["action"], st.P["actor/samples"], st.P["actor/samples_std"])
        b["log_prob"] = (lp + rng.normal(0, 0.1, B)).astype(np.float32)
        tb = {k: torch.from_numpy(v) for k, v in b.items()}
        (l, pl, vl), G, aux = oracle.dppo_grads(st.P, st.norm_mean, st.norm_std, b, 36, 35)
        loss, tpl, tvl, tlogp, tval = tcs.losses(ts, tb)
        grads = torch.autograd.grad(loss, [ts.P[k] for k in tcs.ORDER])
        np.testing.assert_allclose([loss.item(), tpl.item(), tvl.item()], [l, pl, vl], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(tlogp.detach().numpy(), aux["logp"], rtol=1e-5)
        for k, g in zip(tcs.ORDER, grads):
            ref = G[k]
            np.testing.assert_allclose(g.numpy().reshape(ref.shape), ref, rtol=0, atol=2e-4 * np.abs(ref).max())
        oracle.dppo_learner_step(st, [b])
        tcs.learner_step(ts, tb)
        for k in tcs.ORDER:
            # from zero moments Adam's first steps are ~lr * sign(g): entries with |g| ~ 1e-7 amplify rounding
            np.testing.assert_allclose(ts.P[k].detach().numpy().reshape(st.P[k].shape), st.P[k], rtol=0, atol=1e-5)
        np.testing.assert_allclose(ts.mean.numpy(), st.norm_mean, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("sync", [True, False])
def test_td3_grads_match_autograd(oracle, sync):
    """oracle.td3_grads (hand-written backward through the deterministic particle action) vs float64 torch autograd of
    the same graph with the reference's custom gradient of `mean()` (utils.py:205-221), the target network's own actor
    (ddpg.py:201-219) and the policy-delay gate (ddpg.py:199-207)."""
    rng = np.random.default_rng(8)
    S, A, N, B = 11, 3, 35, 9
    st = oracle.TD3State(rng, S, A, N, hidden=(16, 8))
    for k in st.P:
        if k.endswith("weight"):
            st.P[k] = (st.P[k] * 30).astype(np.float32)
    st.PT = {k: (v * 1.05).astype(np.float32) if k.endswith("weight") else v.copy() for k, v in st.P.items()}
    batch = {"state": rng.normal(0, 1, (B, S)).astype(np.float32), "action": np.tanh(rng.normal(0, 1, (B, A))).astype(np.float32),
             "reward": rng.random(B).astype(np.float32), "not_terminal": (rng.random(B) < 0.8).astype(np.float32),
             "state_": rng.normal(0, 1, (B, S)).astype(np.float32)}
    noise_t = rng.normal(0, 1, (B, A)).astype(np.float32)
    norm = (st.norm_mean, st.norm_std)
    (loss, pl, vl), G, aux = oracle.td3_grads(st.P, st.PT, norm, norm, batch, noise_t, A, N, sync)

    T = lambda v, g=False: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=g)
    P = {k: T(v, True) for k, v in st.P.items()}
    PT = {k: T(v) for k, v in st.PT.items()}
    relu6 = lambda t: torch.clamp(t, 0, 6)

    def actor(Pd, x):
        h = relu6(x @ Pd["actor/fc1/weight"] + Pd["actor/fc1/bias"])
        h = relu6(h @ Pd["actor/fc2/weight"] + Pd["actor/fc2/bias"])
        return (h @ Pd["actor/fc_policy/weight"] + Pd["actor/fc_policy/bias"]).view(-1, A, N)

    def qnet(Pd, q, xa):
        h = relu6(xa @ Pd["critic/%s/fc1/weight" % q] + Pd["critic/%s/fc1/bias" % q])
        h = relu6(h @ Pd["critic/%s/fc2/weight" % q] + Pd["critic/%s/fc2/bias" % q])
        return (h @ Pd["critic/%s/fc3/weight" % q] + Pd["critic/%s/fc3/bias" % q])[:, 0]

    def mean_action(Pd, logits):
        w = torch.softmax(logits, -1)
        p = torch.tanh(Pd["actor/samples"]).unsqueeze(0)                       # [1,A,N]
        m = torch.nn.functional.one_hot(w.argmax(-1), N).double()

        class Mask(torch.autograd.Function):
            @staticmethod
            def forward(ctx, w, p):
                y = m * p
                ctx.save_for_backward((p - y.sum(-1, keepdim=True)).detach())
                return y
            @staticmethod
            def backward(ctx, dy):
                return ctx.saved_tensors[0] * dy, (m * dy).sum(0, keepdim=True)
        return Mask.apply(w, p).sum(-1)

    x, x2 = T(batch["state"]).clamp(-5, 5), T(batch["state_"]).clamp(-5, 5)
    a_opt = mean_action(P, actor(P, x))
    q_pi = qnet(P, "q", torch.cat([x, a_opt], -1))
    xh = torch.cat([x, T(batch["action"])], -1)
    q1r, q2r = qnet(P, "q", xh), qnet(P, "twin_q", xh)
    with torch.no_grad():
        a2 = (mean_action(PT, actor(PT, x2)) + 0.2 * T(noise_t)).clamp(-1, 1)
        xa2 = torch.cat([x2, a2], -1)
        qt = T(batch["reward"]) + 0.95 * T(batch["not_terminal"]) * torch.minimum(qnet(PT, "q", xa2), qnet(PT, "twin_q", xa2))
    value_loss = 0.5 * ((q1r - qt) ** 2 + (q2r - qt) ** 2).mean()
    policy_loss = -q_pi.mean()
    np.testing.assert_allclose([pl, vl], [policy_loss.item(), value_loss.item()], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(aux["a_opt"], a_opt.detach().numpy(), rtol=1e-5, atol=1e-6)
    crit = [k for k in P if k.startswith("critic")]
    act = [k for k in P if k.startswith("actor") and k != "actor/samples_std"]
    gc = torch.autograd.grad(value_loss, [P[k] for k in crit], retain_graph=True)
    ga = torch.autograd.grad(policy_loss, [P[k] for k in act])
    for k, g in list(zip(crit, gc)) + list(zip(act, ga)):
        ref = g.numpy() if (sync or k in crit) else np.zeros_like(g.numpy())
        got = np.asarray(G[k]).reshape(ref.shape)
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-4 * (np.abs(g.numpy()).max() + 1e-12), err_msg=k)
    assert not np.any(G["actor/samples_std"])                                  # fixed_sigma
