"""GPU parity: sampling / statistics / entropy / rsample / resample kernels vs the NumPy oracle.
Indices (sampled particle, argmax, resample source) must be BIT-EXACT."""
import numpy as np
import pytest
import torch

from test_gpu_mixture import make_inputs, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,A,N", [(1, 36, 35), (8, 36, 35), (64, 36, 35), (4096, 36, 35), (33, 5, 50), (7, 3, 256)])
def test_sample_plain(oracle, dev, B, A, N):
    from pfpn_amd import kernels
    logits, _, loc, logstd = make_inputs(oracle, B, A, N, 11 + B)
    logits *= 2.5
    rng = np.random.default_rng(B + N)
    u = rng.random((B, A))
    eps = rng.normal(0, 1, (B, A)).astype(np.float32)
    k_ref, a_ref = oracle.sample_plain(logits, loc, logstd, u, eps)
    lp_ref, _ = oracle.mixture_log_prob(logits, a_ref, loc, logstd)
    mx0 = rng.random((A, N)).astype(np.float32) * 0.05
    sm0 = rng.random((A, N)).astype(np.float32)
    mx_ref, sm_ref = oracle.active_stats(logits, mx0, sm0)
    dl, dloc, dls, du, de, dmx, dsm = to_dev(dev, logits, loc, logstd, u, eps, mx0, sm0)
    k, a, lp = kernels.sample(dl, dloc, dls, du, de, dmx, dsm)
    torch.cuda.synchronize()
    assert np.array_equal(k.cpu().numpy(), k_ref), "sampled particle indices must be bit-exact"
    np.testing.assert_allclose(a.cpu().numpy(), a_ref, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dmx.cpu().numpy(), mx_ref, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dsm.cpu().numpy(), sm_ref, rtol=1e-5, atol=1e-5)


def test_sample_edge_uniforms(oracle, dev):
    """u = 0 and u -> 1, plus a one-hot-like row: the first/last particle rule of upper_bound."""
    from pfpn_amd import kernels
    B, A, N = 4, 36, 35
    logits, _, loc, logstd = make_inputs(oracle, B, A, N, 3)
    logits[2, :, :] = -50.0
    logits[2, :, 17] = 30.0
    u = np.zeros((B, A))
    u[1] = np.nextafter(1.0, 0.0)
    u[2] = 0.5
    u[3] = np.random.default_rng(0).random(A)
    eps = np.zeros((B, A), np.float32)
    k_ref, a_ref = oracle.sample_plain(logits, loc, logstd, u, eps)
    assert np.all(k_ref[0] == 0) and np.all(k_ref[1] == N - 1) and np.all(k_ref[2] == 17)
    d = to_dev(dev, logits, loc, logstd, u, eps)
    k, a, _ = kernels.sample(*d)
    assert np.array_equal(k.cpu().numpy(), k_ref)
    np.testing.assert_allclose(a.cpu().numpy(), a_ref, rtol=1e-6)


def test_active_stats_only(oracle, dev):
    from pfpn_amd import kernels
    B, A, N = 300, 36, 35
    logits, *_ = make_inputs(oracle, B, A, N, 21)
    mx0 = np.zeros((A, N), np.float32); sm0 = np.zeros((A, N), np.float32)
    mx_ref, sm_ref = oracle.active_stats(logits, mx0, sm0)
    dl, dmx, dsm = to_dev(dev, logits, mx0, sm0)
    kernels.active_stats(dl, dmx, dsm)
    np.testing.assert_allclose(dmx.cpu().numpy(), mx_ref, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(dsm.cpu().numpy(), sm_ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dsm.cpu().numpy().sum(-1), B, rtol=1e-4)      # softmax rows sum to 1


@pytest.mark.parametrize("tanh_mode", [False, True])
def test_mean_action(oracle, dev, tanh_mode):
    from pfpn_amd import kernels
    B, A, N = 50, 36, 35
    logits, _, loc, _ = make_inputs(oracle, B, A, N, 31, tanh_mode)
    logits[3, 4, 7] = logits[3, 4].max() + 1; logits[3, 4, 20] = logits[3, 4, 7]     # tie -> first index
    k_ref, a_ref = oracle.mean_action(logits, loc, tanh_mode)
    dl, dloc = to_dev(dev, logits, loc)
    k, a = kernels.mean_action(dl, dloc, tanh_mode)
    assert np.array_equal(k.cpu().numpy(), k_ref) and k_ref[3, 4] == 7
    np.testing.assert_allclose(a.cpu().numpy(), a_ref, rtol=1e-6)


def test_entropy(oracle, dev):
    from pfpn_amd import kernels
    B, A, N = 77, 36, 35
    logits, *_ = make_inputs(oracle, B, A, N, 41)
    H_ref = oracle.categorical_entropy(logits)
    g = np.random.default_rng(1).normal(0, 1, (B, A)).astype(np.float32)
    d_ref = oracle.categorical_entropy_bwd(logits, g)
    dl, dg = to_dev(dev, logits, g)
    H = kernels.entropy_fwd(dl.view(B, A * N), A, N)
    dlog = kernels.entropy_bwd(dl.view(B, A * N), dg, A, N)
    np.testing.assert_allclose(H.cpu().numpy(), H_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dlog.cpu().numpy().reshape(B, A, N), d_ref, rtol=1e-4, atol=1e-6)
    # accumulate flavour
    base = torch.ones(B, A * N, device=dev)
    kernels.entropy_bwd(dl.view(B, A * N), dg, A, N, dlogits=base)
    np.testing.assert_allclose(base.cpu().numpy().reshape(B, A, N), d_ref + 1, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,A,N", [(1, 36, 35), (256, 36, 35), (1024, 36, 35), (9, 4, 50)])
def test_rsample(oracle, dev, B, A, N):
    from pfpn_amd import kernels
    logits, _, loc, logstd = make_inputs(oracle, B, A, N, 51 + B, tanh_mode=True)
    rng = np.random.default_rng(B)
    gum = (-np.log(-np.log(rng.uniform(1e-12, 1, logits.shape)))).astype(np.float32)
    eps = rng.normal(0, 1, logits.shape).astype(np.float32)
    da = rng.normal(0, 1, (B, A)).astype(np.float32)
    ds = rng.normal(0, 1, (B, A)).astype(np.float32)
    a_ref, s_ref, k_ref = oracle.rsample(logits, loc, logstd, gum, eps)
    dl_ref, dloc_ref, dls_ref = oracle.rsample_bwd(logits, loc, logstd, gum, eps, da, ds)
    d = to_dev(dev, logits, loc, logstd, gum, eps, da, ds)
    a, s_, k = kernels.rsample_fwd(*d[:5])
    assert np.array_equal(k.cpu().numpy(), k_ref), "argmax(logits+gumbel) must be bit-exact"
    np.testing.assert_allclose(s_.cpu().numpy(), s_ref, rtol=1e-6, atol=1e-6)   # loc + scale*eps cancels
    np.testing.assert_allclose(a.cpu().numpy(), a_ref, rtol=1e-5, atol=1e-6)
    dlg, dloc, dls = kernels.rsample_bwd(*d)
    sc = np.abs(dl_ref).max() + 1e-30
    np.testing.assert_allclose(dlg.cpu().numpy().reshape(B, A, N), dl_ref, rtol=2e-4, atol=2e-6 * sc)
    np.testing.assert_allclose(dloc.cpu().numpy(), dloc_ref, rtol=1e-4, atol=1e-5 * (np.abs(dloc_ref).max() + 1e-30))
    np.testing.assert_allclose(dls.cpu().numpy(), dls_ref, rtol=1e-4, atol=1e-5 * (np.abs(dls_ref).max() + 1e-30))


def _resample_case(oracle, A, N, H, seed, tanh_mode, frac_dead=0.3):
    rng = np.random.default_rng(seed)
    loc, logstd = oracle.init_particles(A, N, tanh_mode)
    loc = (loc + rng.normal(0, 0.02, loc.shape)).astype(np.float32)
    logstd = (logstd + rng.normal(0, 0.1, logstd.shape)).astype(np.float32)
    bias = rng.normal(0, 0.5, A * N).astype(np.float32)
    weight = rng.normal(0, 0.01, (H, A * N)).astype(np.float32)
    sum_active = (rng.random((A, N)) * 100).astype(np.float32)
    thr = np.float32(0.05 / N)
    max_active = (thr * (1 + rng.random((A, N)) * 5)).astype(np.float32)
    dead = rng.random((A, N)) < frac_dead
    max_active[dead] = (thr * rng.random(dead.sum()) * 0.99).astype(np.float32)
    sum_active[dead] *= 0.01
    if A > 2:
        max_active[1] = thr * 2            # a dim with no dead particle
        max_active[2, :] = 0; max_active[2, 5] = 1.0   # all but one dead
    u_res = rng.random((A, N))
    u_noise = rng.uniform(-1, 1, (A, N)).astype(np.float32)
    return loc, logstd, bias, weight, max_active, sum_active, u_res, u_noise


@pytest.mark.parametrize("A,N,H,tanh_mode", [(36, 35, 512, False), (36, 35, 512, True), (6, 50, 64, False), (3, 200, 32, False)])
def test_resample(oracle, dev, A, N, H, tanh_mode):
    from pfpn_amd import kernels
    case = _resample_case(oracle, A, N, H, 7 + A + N, tanh_mode)
    loc, logstd, bias, weight, mx, sm, ur, un = case
    r_loc, r_ls, r_bias, r_w, info = oracle.resample_particles(loc, logstd, bias, weight, mx, sm, ur, un,
                                                               tanh_mode=tanh_mode)
    d = to_dev(dev, *case)
    dead, src = kernels.resample(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], 0.0, tanh_mode)
    torch.cuda.synchronize()
    dead = dead.cpu().numpy().astype(bool)
    assert np.array_equal(dead, info["dead"])
    assert np.array_equal(src.cpu().numpy(), info["src"]), "resample source indices must be bit-exact"
    assert dead.sum() > 0
    np.testing.assert_allclose(d[0].cpu().numpy(), r_loc, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(d[1].cpu().numpy(), r_ls, rtol=0, atol=0)
    np.testing.assert_allclose(d[2].cpu().numpy(), r_bias, rtol=1e-6, atol=1e-6)
    assert np.array_equal(d[3].cpu().numpy(), r_w), "weight columns are copies: exact"


def test_resample_preserves_policy(oracle, dev):
    """Invariant: for every dim, the softmax mass of {source + its clones} after the
    update equals the source's mass before (bias split by log(count+1-delta))."""
    from pfpn_amd import kernels
    A, N, H = 36, 35, 128
    case = _resample_case(oracle, A, N, H, 99, False)
    loc, logstd, bias, weight, mx, sm, ur, un = case
    rng = np.random.default_rng(5)
    h = rng.normal(0, 1, (16, H)).astype(np.float32)
    before = oracle.softmax((h @ weight + bias).reshape(16, A, N))
    d = to_dev(dev, *case)
    dead, src = kernels.resample(*d[:8], 0.0, False)
    dead = dead.cpu().numpy().astype(bool); src = src.cpu().numpy()
    after = oracle.softmax((h @ d[3].cpu().numpy() + d[2].cpu().numpy()).reshape(16, A, N))
    for a in range(A):
        alive_mass_before = before[:, a, ~dead[a]].sum(-1)
        # dead particles' own mass is discarded and renormalised away: compare ratios among sources
        for s in np.unique(src[a][dead[a]]):
            if dead[a, s]:
                continue
            grp = [s] + [j for j in np.nonzero(dead[a])[0] if src[a, j] == s]
            np.testing.assert_allclose(after[:, a, grp].sum(-1) / after[:, a, ~dead[a] | True].sum(-1),
                                       after[:, a, grp].sum(-1), rtol=1e-5)
            # ratio to an untouched live particle is unchanged
            live = [n for n in range(N) if not dead[a, n] and n not in src[a][dead[a]]]
            if live:
                l0 = live[0]
                np.testing.assert_allclose(after[:, a, grp].sum(-1) / after[:, a, l0],
                                           before[:, a, s] / before[:, a, l0], rtol=2e-4)


def test_resample_noop_when_all_alive(oracle, dev):
    from pfpn_amd import kernels
    A, N, H = 36, 35, 64
    case = list(_resample_case(oracle, A, N, H, 123, False))
    case[4] = np.ones((A, N), np.float32)
    d = to_dev(dev, *case)
    dead, _ = kernels.resample(*d[:8], 0.0, False)
    assert dead.sum().item() == 0
    for got, ref in zip(d[:4], case[:4]):
        assert np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("A,N,H,k,tanh_mode", [(36, 35, 128, 3, False), (36, 35, 128, 35, False), (6, 50, 64, 1, True),
                                               (3, 200, 32, 17, False)])
def test_resample_topk_branch(oracle, dev, A, N, H, k, tanh_mode):
    """`resample = k > 0` (a2c.py:405-408): source drawn uniformly among the k most active particles; indices
    bit-exact against the oracle, ties (equal averages) to the lower index."""
    from pfpn_amd import kernels
    case = list(_resample_case(oracle, A, N, H, 31 + A + N + k, tanh_mode))
    case[5][0, 3] = case[5][0, 4] = case[5][0].max() * 2            # a tie at the top of dim 0
    loc, logstd, bias, weight, mx, sm, ur, un = case
    r_loc, r_ls, r_bias, r_w, info = oracle.resample_particles(loc, logstd, bias, weight, mx, sm, ur, un,
                                                               tanh_mode=tanh_mode, topk=k)
    d = to_dev(dev, *case)
    dead, src = kernels.resample(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], 0.0, tanh_mode, topk=k)
    dead = dead.cpu().numpy().astype(bool); src = src.cpu().numpy()
    assert np.array_equal(dead, info["dead"]) and dead.sum() > 0
    assert np.array_equal(src, info["src"]), "top-k source indices must be bit-exact"
    # every source is one of the k most active particles of its action dimension
    avg = sm / sm.sum(1, keepdims=True)
    for a in range(A):
        kth = np.sort(avg[a])[::-1][min(k, N) - 1]
        assert np.all(avg[a, src[a]] >= kth)
    if k == 1:
        assert np.all(src == src[:, :1])
    np.testing.assert_allclose(d[0].cpu().numpy(), r_loc, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(d[1].cpu().numpy(), r_ls, rtol=0, atol=0)
    np.testing.assert_allclose(d[2].cpu().numpy(), r_bias, rtol=1e-6, atol=1e-6)
    assert np.array_equal(d[3].cpu().numpy(), r_w)


def test_network_resample_topk_option(dev):
    """the reference default `resample=3` (a2c.py:311) builds and ticks."""
    from test_gpu_dppo_step import make_net
    net = make_net(dev, resample=3, resample_interval=1)
    net.max_active.fill_(1.0); net.sum_active.copy_(torch.rand_like(net.sum_active) + 0.1)
    net.max_active[:, :5] = 0.0
    before = net.loc.clone()
    dead, src = net.resample_now()
    assert dead[:, :5].all() and dead.sum().item() == 5 * net.n_actions
    top3 = torch.topk(net.sum_active, 3, dim=1).indices
    assert all(int(src[a, j]) in top3[a].tolist() for a in range(net.n_actions) for j in range(5))
    assert not torch.equal(before[:, :5], net.loc[:, :5]) and torch.equal(before[:, 5:], net.loc[:, 5:])


@pytest.mark.parametrize("tanh_mode", [False, True])
def test_resample_fixed_sigma(oracle, dev, tanh_mode):
    """`fixed_sigma` resampling (a2c.py:430-440, :463-465): jitter with the fixed std (around tanh(loc) in tanh mode), the
    log-std variable is not touched."""
    from pfpn_amd import kernels
    A, N, H = 36, 35, 64
    case = _resample_case(oracle, A, N, H, 17, tanh_mode)
    loc, logstd, bias, weight, mx, sm, ur, un = case
    r_loc, r_ls, r_bias, r_w, info = oracle.resample_particles(loc, logstd, bias, weight, mx, sm, ur, un,
                                                               tanh_mode=tanh_mode, fixed_sigma=0.07)
    d = to_dev(dev, *case)
    dead, src = kernels.resample(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], 0.0, tanh_mode, fixed_sigma=0.07)
    assert np.array_equal(src.cpu().numpy(), info["src"]) and np.array_equal(dead.cpu().numpy().astype(bool), info["dead"])
    assert np.array_equal(d[1].cpu().numpy(), logstd) and np.array_equal(r_ls, logstd)
    np.testing.assert_allclose(d[0].cpu().numpy(), r_loc, rtol=3e-6, atol=2e-7)
    np.testing.assert_allclose(d[2].cpu().numpy(), r_bias, rtol=1e-6, atol=1e-6)
    assert np.array_equal(d[3].cpu().numpy(), r_w)
    dd = info["dead"]
    if not tanh_mode:      # the jitter is bounded by the fixed std (+1e-4)
        assert np.all(np.abs(d[0].cpu().numpy()[dd] - loc[np.nonzero(dd)[0], info["src"][dd]]) <= 0.07 + 1.1e-4)
