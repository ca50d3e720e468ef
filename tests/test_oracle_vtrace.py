"""CPU: the oracle's V-trace / A2C restatement against independent formulations (float64 loops, torch autograd)."""
import numpy as np
import torch

F = np.float32


def vtrace_reference_f64(lt, lb, v, r, nt, E, gamma, rho_clip, pg_clip, seq_len):
    """Per-episode scalar recursion straight from the IMPALA paper + the reference's bootstrap quirks
    (vtrace.py:91-97: the value after the last step of a segment is that step's own value)."""
    T = len(v) // E
    vs, pg = np.zeros(E * T), np.zeros(E * T)
    for e in range(E):
        end = (seq_len[e] if seq_len is not None else T) - 1
        sl = slice(e * T, (e + 1) * T)
        rho = np.exp(lt[sl].astype(np.float64) - lb[sl])
        V, R, D = v[sl].astype(np.float64), r[sl].astype(np.float64), gamma * nt[sl].astype(np.float64)
        a_next, out = 0.0, np.zeros(T)
        for t in range(T - 1, -1, -1):
            last = t == T - 1 or t == end
            v1 = V[t] if last else V[t + 1]
            td = min(rho_clip, rho[t]) * (R[t] + D[t] * v1 - V[t])
            a = td + (0.0 if t == end and seq_len is not None else D[t]) * min(1.0, rho[t]) * a_next
            out[t] = a + V[t]
            a_next = a
        for t in range(T):
            last = t == T - 1 or t == end
            vs1 = V[t] if last else out[t + 1]
            pg[e * T + t] = min(pg_clip, rho[t]) * (R[t] + D[t] * vs1 - V[t])
        vs[sl] = out
    return vs, pg


def test_vtrace_matches_f64_recursion(oracle):
    rng = np.random.default_rng(0)
    for E, T, sl in ((1, 64, None), (4, 64, [64, 50, 1, 33]), (3, 7, [7, 4, 6])):
        n = E * T
        lt, lb = rng.normal(-30, 1, n).astype(F), rng.normal(-30, 1, n).astype(F)
        v, r = rng.normal(0, 1, n).astype(F), rng.normal(0.5, 0.3, n).astype(F)
        nt = (rng.uniform(size=n) > 0.05).astype(F)
        vs, pg, mask = oracle.vtrace(lt, lb, v, r, nt, E, 0.95, 1.0, 1.0, sl)
        rvs, rpg = vtrace_reference_f64(lt, lb, v, r, nt, E, 0.95, 1.0, 1.0, sl)
        m = mask > 0
        np.testing.assert_allclose(vs[m], rvs[m], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(pg[m], rpg[m], rtol=2e-5, atol=2e-5)
        if sl is not None:
            assert mask.reshape(E, T).sum(1).tolist() == sl
    # on-policy, no clipping active, not_terminal=1: vs - V is the discounted sum of TD errors (lambda = 1)
    E, T = 2, 16
    v, r = rng.normal(0, 1, E * T).astype(F), rng.normal(0, 1, E * T).astype(F)
    lp = rng.normal(-20, 1, E * T).astype(F)
    vs, pg, _ = oracle.vtrace(lp, lp, v, r, np.ones(E * T, F), E, 0.9)
    for e in range(E):
        V, R = v[e * T:(e + 1) * T].astype(np.float64), r[e * T:(e + 1) * T].astype(np.float64)
        V1 = np.append(V[1:], V[-1])
        delta = R + 0.9 * V1 - V
        want = np.array([sum(0.9 ** (k - t) * delta[k] for k in range(t, T)) for t in range(T)]) + V
        np.testing.assert_allclose(vs[e * T:(e + 1) * T], want, rtol=1e-5, atol=1e-5)


def test_a2c_loss_gradients_vs_autograd(oracle):
    rng = np.random.default_rng(1)
    B = 37
    logp, adv = rng.normal(-20, 2, B).astype(F), rng.normal(0, 1, B).astype(F)
    v, vt = rng.normal(0, 1, B).astype(F), rng.normal(0, 1, B).astype(F)
    for mask in (None, (rng.uniform(size=B) > 0.3).astype(F)):
        loss, pl, vl, dlogp, dv = oracle.a2c_loss(logp, adv, v, vt, 0.5, mask)
        tl = torch.tensor(logp, dtype=torch.float64, requires_grad=True)
        tv = torch.tensor(v, dtype=torch.float64, requires_grad=True)
        m = torch.ones(B, dtype=torch.float64) if mask is None else torch.tensor(mask, dtype=torch.float64)
        sel = m > 0
        tpl = -(tl * torch.tensor(adv, dtype=torch.float64))[sel].mean()
        tvl = 0.5 * ((tv - torch.tensor(vt, dtype=torch.float64))[sel] ** 2).mean()
        (tpl + tvl).backward()
        np.testing.assert_allclose([pl, vl, loss], [tpl.item(), tvl.item(), (tpl + tvl).item()], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dlogp, tl.grad.numpy(), rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(dv, tv.grad.numpy(), rtol=1e-5, atol=1e-8)


def test_a2c_grads_entropy_term(oracle):
    """entropy_beta path of a2c_grads: finite differences of the full loss w.r.t. the policy bias."""
    rng = np.random.default_rng(2)
    S, A, N, B = 11, 3, 5, 6
    P = oracle.init_params(rng, S, A, N, hidden=(16, 8))
    for k in P:
        if k.endswith("weight"):
            P[k] = (P[k] * 30).astype(F)
    batch = {"state": rng.normal(0, 1, (B, S)).astype(F), "action": rng.uniform(-1, 1, (B, A)).astype(F),
             "value_target": rng.normal(0, 1, B).astype(F), "advantage": rng.normal(0, 1, B).astype(F)}
    nm, ns = np.zeros(S, F), np.ones(S, F)
    beta = 0.3
    (loss, ent, pl, vl), G, _ = oracle.a2c_grads(P, nm, ns, batch, A, N, 0.5, beta)
    assert ent is not None and abs(loss - (pl + vl)) < 1e-6
    g = G["actor/fc_policy/bias"]
    for j in (0, 7, 14):
        Pp = {k: v.copy() for k, v in P.items()}; Pm = {k: v.copy() for k, v in P.items()}
        Pp["actor/fc_policy/bias"][j] += 1e-2; Pm["actor/fc_policy/bias"][j] -= 1e-2
        lp = oracle.a2c_grads(Pp, nm, ns, batch, A, N, 0.5, beta)[0][0]
        lm = oracle.a2c_grads(Pm, nm, ns, batch, A, N, 0.5, beta)[0][0]
        assert abs((lp - lm) / 2e-2 - g[j]) < 5e-3 * max(1.0, abs(g[j]))
