"""CPU: the oracle's C twin (oracle/pfpn_cpu_twin.c -- host twins of three C-ABI entry points, test / bench infrastructure)
against the NumPy oracle: two independent restatements of the same reference lines must agree to fp32 rounding."""
import numpy as np
import pytest

from oracle import cpu_twin

F = np.float32


def inputs(rng, B, A, N, spread=1.0):
    logits = rng.normal(0, 1.5, (B, A, N)).astype(F)
    x = rng.uniform(-1.05, 1.05, (B, A)).astype(F) * F(spread)
    loc = (np.linspace(-1, 1, N)[None, :] + rng.normal(0, 0.01, (A, N))).astype(F)
    logstd = (np.log(2.0 / max(N - 1, 1)) + rng.normal(0, 0.1, (A, N))).astype(F)
    return logits, x, loc, logstd


@pytest.mark.parametrize("B,A,N", [(1, 1, 1), (7, 3, 5), (64, 36, 35), (33, 8, 100)])
@pytest.mark.parametrize("tanh_mode", [False, True])
def test_logprob_forward_and_backward_match_the_numpy_oracle(oracle, B, A, N, tanh_mode):
    rng = np.random.default_rng(B * 131 + N)
    logits, x, loc, logstd = inputs(rng, B, A, N)
    g = rng.normal(0, 1, B).astype(F)
    lp, p = cpu_twin.logprob_fwd(logits, x, loc, logstd, tanh_mode)
    ref_lp, ref_p = oracle.mixture_log_prob(logits, x, loc, logstd, tanh_mode)
    np.testing.assert_allclose(p, ref_p, rtol=3e-6, atol=0)
    np.testing.assert_allclose(lp, ref_lp, rtol=1e-5, atol=1e-5)
    dl, dloc, dls, dx = cpu_twin.logprob_bwd(logits, x, loc, logstd, g, tanh_mode)
    r_dl, r_dloc, r_dls, r_dx = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, g, tanh_mode)
    np.testing.assert_allclose(dl, r_dl, rtol=1e-4, atol=2e-6 * max(np.abs(r_dl).max(), 1e-30))
    np.testing.assert_allclose(dx, r_dx, rtol=1e-4, atol=1e-5 * max(np.abs(r_dx).max(), 1e-30))
    for got, ref in ((dloc, r_dloc), (dls, r_dls)):
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5 * max(np.abs(ref).max(), 1e-30))


def test_underflow_guard_like_the_reference(oracle):
    """x far outside the grid: p underflows to 0, log p = -inf, and the guard turns d log p / dp = g / 0 into 0."""
    rng = np.random.default_rng(3)
    logits, x, loc, logstd = inputs(rng, 6, 4, 35)
    x[2, 1] = F(40.0)
    g = rng.normal(0, 1, 6).astype(F)
    lp, p = cpu_twin.logprob_fwd(logits, x, loc, logstd)
    ref_lp, ref_p = oracle.mixture_log_prob(logits, x, loc, logstd)
    assert p[2, 1] == 0 and ref_p[2, 1] == 0 and np.isneginf(lp[2]) and np.isneginf(ref_lp[2])
    dl, dloc, dls, dx = cpu_twin.logprob_bwd(logits, x, loc, logstd, g)
    r = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, g)
    assert not dl[2, 1].any() and not r[0][2, 1].any() and np.isfinite(dl).all() and np.isfinite(dloc).all()
    np.testing.assert_allclose(dl, r[0], rtol=1e-4, atol=2e-6 * np.abs(r[0]).max())


def test_gae_matches_the_oracle_on_ragged_episodes(oracle):
    rng = np.random.default_rng(5)
    lens = [1, 2, 17, 64, 5, 300]
    off = np.concatenate([[0], np.cumsum(lens)])
    reward = rng.normal(0, 1, off[-1]).astype(F)
    value = rng.normal(0, 1, off[-1] + len(lens)).astype(F)
    adv = cpu_twin.gae(reward, value, off, 0.95, 0.95 * 0.95)
    ref = np.concatenate([oracle.gae(reward[off[e]:off[e + 1]], value[off[e] + e:off[e + 1] + e + 1], 0.95, 0.95 * 0.95)
                          for e in range(len(lens))])
    assert np.array_equal(adv, ref)          # same fp32 td, same float64 scan: bit for bit


def test_the_product_never_loads_the_twin():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d, _, files in os.walk(os.path.join(root, "pfpn_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert "cpu_twin" not in open(os.path.join(d, f), errors="ignore").read(), os.path.join(d, f)
