"""GPU parity: GAE, state normaliser, dense epilogues, value head, clip + TF-Adam vs the oracle."""
import numpy as np
import pytest
import torch

from test_gpu_mixture import to_dev

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.mark.parametrize("lens", [[512] * 8, [1], [3, 600, 1, 77], [5000, 2049, 2048]])
def test_gae_bit_exact(oracle, dev, lens):
    """fp32 deltas + sequential float64 scan in the reference's op order -> bit-exact fp32 output."""
    from pfpn_amd import kernels
    rng = np.random.default_rng(len(lens))
    rs = [rng.normal(0, 1, T).astype(F) for T in lens]
    vs = [rng.normal(0, 3, T + 1).astype(F) for T in lens]
    ref = np.concatenate([oracle.gae(r, v, 0.95, 0.95 * 0.95) for r, v in zip(rs, vs)])
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dr, dv, do = to_dev(dev, np.concatenate(rs), np.concatenate(vs), offs)
    adv = kernels.gae(dr, dv, do, 0.95, 0.95 * 0.95)
    assert np.array_equal(adv.cpu().numpy(), ref)


@pytest.mark.parametrize("B", [1, 32, 4096])
def test_state_norm_and_update(oracle, dev, B):
    from pfpn_amd import kernels
    S = 197
    rng = np.random.default_rng(B)
    s = (rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F)
    s[0, :5] = 100.0
    mean = rng.normal(0, 0.1, S).astype(F)
    std = (1 + rng.random(S)).astype(F)
    x_ref = oracle.state_normalize(s, mean, std, 5.0)
    ds, dm, dsd = to_dev(dev, s, mean, std)
    x = kernels.state_norm(ds, dm, dsd, 5.0)
    np.testing.assert_allclose(x.cpu().numpy(), x_ref, rtol=1e-6, atol=1e-7)
    assert x.max().item() <= 5.0 and x.min().item() >= -5.0
    for step in (0, 5, 100000):
        m_ref, s_ref = oracle.state_norm_update(s, mean, std, step)
        m, sd = kernels.state_norm_update(ds, step, dm, dsd)
        np.testing.assert_allclose(m.cpu().numpy(), m_ref, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sd.cpu().numpy(), s_ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("rows,H,G", [(4096, 2048, 1), (33, 512, 1), (2 * 4096, 512, 2), (2 * 7, 512, 2), (5, 1260, 1)])
def test_bias_act_and_bwd(dev, rows, H, G):
    from pfpn_amd import kernels
    rng = np.random.default_rng(rows + H)
    y0 = rng.normal(0, 3, (rows, H)).astype(F)
    bias = rng.normal(0, 1, (G, H)).astype(F)
    gr = rows // G if G > 1 else 0
    ref = np.clip(y0 + np.repeat(bias, rows // G, axis=0), 0, 6)
    dy0, db0 = to_dev(dev, y0, bias)
    y = kernels.bias_act(dy0, db0, True, gr)
    np.testing.assert_array_equal(y.cpu().numpy(), ref.astype(F))
    g = rng.normal(0, 1, (rows, H)).astype(F)
    mask = (ref > 0) & (ref < 6)
    g_ref = np.where(mask, g, 0).astype(F)
    db_ref = g_ref.reshape(G, rows // G, H).sum(1, dtype=np.float64)
    dg, = to_dev(dev, g)
    db = torch.empty(G, H, device=dev)
    kernels.act_bwd_bias_grad(y, dg, db, True, gr)
    np.testing.assert_array_equal(dg.cpu().numpy(), g_ref)
    np.testing.assert_allclose(db.cpu().numpy(), db_ref, rtol=1e-4, atol=1e-4)
    # linear layer (no activation): dy untouched, db = column sums
    dg2, = to_dev(dev, g)
    kernels.act_bwd_bias_grad(None, dg2, db, False, gr)
    np.testing.assert_array_equal(dg2.cpu().numpy(), g)
    np.testing.assert_allclose(db.cpu().numpy(), g.reshape(G, rows // G, H).sum(1, dtype=np.float64), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B", [1, 100, 4096])
def test_value_head(dev, B):
    from pfpn_amd import kernels
    H = 512
    rng = np.random.default_rng(B)
    h = np.clip(rng.normal(1, 3, (B, H)), 0, 6).astype(F)
    w = rng.normal(0, 0.05, H).astype(F)
    b = np.array([0.3], F)
    dv = rng.normal(0, 1, B).astype(F)
    dh_, dw_, db_, ddv = to_dev(dev, h, w, b, dv)
    v = kernels.value_head_fwd(dh_, dw_, db_)
    np.testing.assert_allclose(v.cpu().numpy(), h.astype(np.float64) @ w + 0.3, rtol=1e-5, atol=1e-5)
    dh = torch.empty(B, H, device=dev); dw = torch.empty(H, device=dev); db = torch.empty(1, device=dev)
    kernels.value_head_bwd(dh_, dw_, ddv, dh, dw, db, True)
    mask = (h > 0) & (h < 6)
    np.testing.assert_allclose(dh.cpu().numpy(), np.where(mask, dv[:, None] * w[None], 0), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(dw.cpu().numpy(), h.astype(np.float64).T @ dv, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), [dv.sum(dtype=np.float64)], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,scale", [(2104517, 1.0), (2104517, 1e-4), (1003, 1.0)])
def test_clip_and_tf_adam(oracle, dev, n, scale):
    from pfpn_amd import kernels
    rng = np.random.default_rng(n)
    p = rng.normal(0, 0.01, n).astype(F)
    g = (rng.normal(0, 1, n) * scale / np.sqrt(n) * 3).astype(F)
    m = (rng.normal(0, 1, n) * 1e-3).astype(F)
    v = (rng.random(n) * 1e-6).astype(F)
    (gc,), norm = oracle.clip_by_global_norm([g], 1.0)
    p_ref, m_ref, v_ref = oracle.tf_adam(p, gc, m, v, F(0.9) ** 3, F(0.999) ** 3, 1e-4)
    dp, dg, dm, dv = to_dev(dev, p, g, m, v)
    bp = torch.tensor([F(0.9) ** 3, F(0.999) ** 3], device=dev, dtype=torch.float32)
    ns = torch.empty(2, device=dev)
    kernels.clip_by_global_norm(dg, 1.0, ns, apply=True)
    np.testing.assert_allclose(ns.cpu().numpy()[0], norm, rtol=2e-6)
    np.testing.assert_allclose(dg.cpu().numpy(), gc, rtol=2e-6, atol=0)
    kernels.tf_adam(dp, dg, dm, dv, bp, 1e-4)
    if norm < 1.0:
        # scale == 1 exactly -> the whole update is bit-exact against the oracle
        assert np.array_equal(dg.cpu().numpy(), g)
        assert np.array_equal(dm.cpu().numpy(), m_ref) and np.array_equal(dv.cpu().numpy(), v_ref)
        assert np.array_equal(dp.cpu().numpy(), p_ref)
    else:
        np.testing.assert_allclose(dm.cpu().numpy(), m_ref, rtol=1e-5, atol=1e-9)     # |m| ~ 1e-3, fp32 resolution
        np.testing.assert_allclose(dv.cpu().numpy(), v_ref, rtol=1e-5, atol=1e-12)
        np.testing.assert_allclose(dp.cpu().numpy(), p_ref, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(bp.cpu().numpy(), [F(0.9) ** 4, F(0.999) ** 4], rtol=1e-7)


def test_adam_fused_scale_matches_two_pass(dev):
    """norm_scale/grad_mul folded into the Adam kernel == scaling the gradient first."""
    from pfpn_amd import kernels
    n = 40001
    rng = np.random.default_rng(0)
    p, g = rng.normal(0, 1, n).astype(F), rng.normal(0, 1, n).astype(F)
    a = to_dev(dev, p, g, np.zeros(n, F), np.zeros(n, F))
    b = to_dev(dev, p, g, np.zeros(n, F), np.zeros(n, F))
    ns = torch.empty(2, device=dev)
    kernels.clip_by_global_norm(a[1], 1.0, ns, apply=False)
    bp1 = torch.tensor([0.9, 0.999], device=dev); bp2 = bp1.clone()
    kernels.tf_adam(a[0], a[1], a[2], a[3], bp1, 1e-4, norm_scale=ns, grad_mul=0.125)
    kernels.clip_by_global_norm(b[1], 1.0, ns, apply=True)
    b[1].mul_(0.125)
    kernels.tf_adam(b[0], b[1], b[2], b[3], bp2, 1e-4)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])


def test_soft_update(dev):
    from pfpn_amd import kernels
    t = torch.randn(10007, device=dev); s = torch.randn(10007, device=dev)
    ref = (1 - 0.005) * t + 0.005 * s
    kernels.soft_update(t, s, 0.005)
    torch.testing.assert_close(t, ref, rtol=1e-6, atol=1e-7)


def test_ppo_loss_unfused(oracle, dev):
    from pfpn_amd import kernels
    B = 3000
    rng = np.random.default_rng(2)
    lp = rng.normal(-30, 1, B).astype(F)
    lp_old = (lp + rng.normal(0, 0.2, B)).astype(F)
    adv, v, v_old = (rng.normal(0, 1, B).astype(F) for _ in range(3))
    loss, pl, vl, dlogp, dv = oracle.ppo_loss(lp, lp_old, adv, v, v_old)
    d = to_dev(dev, lp, lp_old, adv, v, v_old)
    st = kernels.adv_stats(d[2])
    g_lp, g_v, l3 = kernels.ppo_loss(*d, st)
    np.testing.assert_allclose(l3.cpu().numpy(), [loss, pl, vl], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(g_lp.cpu().numpy(), dlogp, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(g_v.cpu().numpy(), dv, rtol=1e-5, atol=1e-10)


def test_fused_prologue_matches_separate_launches(dev):
    """pfpn_state_prologue + pfpn_state_norm_ema == pfpn_state_norm + pfpn_state_norm_update + pfpn_adv_stats."""
    from pfpn_amd import kernels
    rng = np.random.default_rng(5)
    for B, S in ((4096, 197), (33, 7), (1, 197)):
        s = torch.from_numpy((rng.normal(0, 1, (B, S)) * 3 + 0.5).astype(np.float32)).to(dev)
        mean = torch.from_numpy(rng.normal(0, 0.3, S).astype(np.float32)).to(dev)
        std = torch.from_numpy(rng.uniform(0.5, 2, S).astype(np.float32)).to(dev)
        adv = torch.from_numpy(rng.normal(0.2, 1.5, B).astype(np.float32)).to(dev)
        step = torch.tensor([3.0], device=dev)
        x_ref = kernels.state_norm(s, mean, std, 5.0)
        m_ref, sd_ref = kernels.state_norm_update(s, step, mean, std)
        st_ref = kernels.adv_stats(adv, normalize=B > 1)
        x = torch.empty_like(s)
        st = kernels.new_adv_stats(dev)
        kernels.state_prologue(s, mean, std, 5.0, x, True, adv, B > 1, st)
        m, sd = torch.empty_like(mean), torch.empty_like(std)
        kernels.state_norm_ema(B, S, step, mean, std, m, sd)
        assert torch.equal(x, x_ref)                                   # same arithmetic
        assert torch.equal(m, m_ref) and torch.equal(sd, sd_ref)       # same float64 moments, same order
        np.testing.assert_allclose(st.cpu().numpy(), st_ref.cpu().numpy(), rtol=2e-6)   # 256- vs 1024-thread sums
        # without the optional parts
        x2 = torch.empty_like(s)
        kernels.state_prologue(s, mean, std, 0.0, x2, False)
        assert torch.equal(x2, kernels.state_norm(s, mean, std, 0.0))


def test_clip_adam_matches_clip_then_adam(dev):
    """pfpn_clip_adam (2 launches) == pfpn_clip_by_global_norm + pfpn_tf_adam per group + power advance, bit for bit."""
    import ctypes
    from pfpn_amd import kernels
    rng = np.random.default_rng(6)
    n = 40000
    cuts = [(0, 12000, 1e-4), (12000, 28000, 3e-4)]
    base = [torch.from_numpy(rng.normal(0, s, n).astype(np.float32)).to(dev) for s in (1.0, 0.05, 0.01)]
    vabs = torch.from_numpy(np.abs(rng.normal(0, 0.01, n)).astype(np.float32)).to(dev)
    for clip in (1.0, 1e6):
        p1, g, m1 = [t.clone() for t in base]
        v1 = vabs.clone()
        p2, m2, v2 = p1.clone(), m1.clone(), v1.clone()
        bp1 = [torch.tensor([0.9 ** 3, 0.999 ** 3], device=dev) for _ in cuts]
        bp2 = torch.tensor([[0.9 ** 3, 0.999 ** 3, 0, 0]] * len(cuts), device=dev)
        step1, step2 = torch.tensor([7.0], device=dev), torch.tensor([7.0], device=dev)
        ns1, ns2 = torch.zeros(2, device=dev), torch.zeros(2, device=dev)
        kernels.clip_by_global_norm(g, clip, ns1, apply=False)
        for gi, (off, ln, lr) in enumerate(cuts):
            sl = slice(off, off + ln)
            kernels.tf_adam(p1[sl], g[sl], m1[sl], v1[sl], bp1[gi], lr, advance=True, norm_scale=ns1,
                            step_counter=step1 if gi == len(cuts) - 1 else None)
        G = len(cuts)
        kernels.clip_adam(p2, g, m2, v2, (ctypes.c_longlong * G)(*[c[0] for c in cuts]),
                          (ctypes.c_longlong * G)(*[c[1] for c in cuts]), (ctypes.c_float * G)(*[c[2] for c in cuts]),
                          bp2, clip, ns2, step_counter=step2)
        assert torch.equal(ns1, ns2) and torch.equal(step1, step2)
        assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2)
        for gi in range(G):
            assert torch.equal(bp2[gi, :2], bp1[gi])                   # TF's beta*_power after the step
            assert torch.allclose(bp2[gi, 2:], torch.tensor([0.9 ** 3, 0.999 ** 3], device=dev))


def test_dp_split_step_matches_separate_kernels(dev):
    """pfpn_clip_scale_inplace + pfpn_adam_groups (the data-parallel split of the optimiser step) ==
    pfpn_clip_by_global_norm(apply) + pfpn_tf_adam(grad_mul) + power advance, bit for bit."""
    import ctypes
    from pfpn_amd import kernels
    rng = np.random.default_rng(8)
    n, W = 30000, 4
    cuts = [(0, 30000, 1e-4)]
    p0 = torch.from_numpy(rng.normal(0, 1, n).astype(np.float32)).to(dev)
    g0 = torch.from_numpy(rng.normal(0, 0.05, n).astype(np.float32)).to(dev)
    p1, g1, m1, v1 = p0.clone(), g0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    p2, g2, m2, v2 = p0.clone(), g0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    ns1, ns2 = torch.zeros(2, device=dev), torch.zeros(2, device=dev)
    bp1 = torch.tensor([0.9, 0.999], device=dev)
    bp2 = torch.tensor([[0.9, 0.999, 0, 0]], device=dev)
    s1, s2 = torch.tensor([0.0], device=dev), torch.tensor([0.0], device=dev)
    kernels.clip_by_global_norm(g1, 1.0, ns1, apply=True)
    kernels.clip_scale_inplace(g2, 1.0, ns2, bp2)
    assert torch.equal(g1, g2) and torch.equal(ns1, ns2)
    g1.mul_(W); g2.mul_(W)                                    # stands in for the all-reduce sum of W equal replicas
    kernels.tf_adam(p1, g1, m1, v1, bp1, 1e-4, advance=True, grad_mul=1.0 / W, step_counter=s1)
    G = 1
    kernels.adam_groups(p2, g2, m2, v2, (ctypes.c_longlong * G)(0), (ctypes.c_longlong * G)(n), (ctypes.c_float * G)(1e-4),
                        bp2, 1.0 / W, step_counter=s2)
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2) and torch.equal(s1, s2)
    assert torch.equal(bp2[0, :2], bp1)


@pytest.mark.parametrize("B,H", [(4096, 512), (32, 512), (8, 64), (12, 260)])
def test_last_layer_epilogue_with_value_head_matches_separate_launches(dev, B, H):
    """bias+ReLU6 with the critic head folded in == pfpn_bias_act followed by pfpn_value_head_fwd, bit for bit."""
    from pfpn_amd import kernels
    rng = np.random.default_rng(B + H)
    y0 = rng.normal(0, 3, (2 * B, H)).astype(F)
    bias = rng.normal(0, 1, (2, H)).astype(F)
    w = rng.normal(0, 0.1, H).astype(F)
    b3 = np.array([0.37], F)
    ya, yb, db, dw, dbb = to_dev(dev, y0, y0, bias, w, b3)
    kernels.bias_act(ya, db, True, B)
    v_ref = kernels.value_head_fwd(ya[B:], dw, dbb)
    v = torch.empty(B, dtype=torch.float32, device=dev)
    kernels.bias_act_value(yb, db, B, 1, dw, dbb, v, relu6=True)
    assert torch.equal(ya, yb)
    assert torch.equal(v, v_ref)
    # group 0 carrying the head instead
    yc = to_dev(dev, y0)[0]
    kernels.bias_act_value(yc, db, B, 0, dw, dbb, v, relu6=True)
    assert torch.equal(v, kernels.value_head_fwd(ya[:B], dw, dbb))


@pytest.mark.parametrize("B,H", [(4096, 512), (32, 512), (7, 64), (300, 260)])
def test_deferred_reductions_match_immediate_ones(dev, B, H):
    """partials + ONE reduce_jobs launch == pfpn_act_bwd_bias_grad / pfpn_value_head_bwd, bit for bit, including the
    critic head's backward folded into the last layer's pass and the state-normaliser EMA job."""
    from pfpn_amd import kernels
    rng = np.random.default_rng(B * 3 + H)
    y = np.clip(rng.normal(2, 3, (2 * B, H)), 0, 6).astype(F)
    g = rng.normal(0, 1, (2 * B, H)).astype(F)
    w = rng.normal(0, 0.1, H).astype(F)
    dv = rng.normal(0, 1, B).astype(F)
    dy, dg_a, dg_b, dw, ddv = to_dev(dev, y, g, g, w, dv)
    # immediate: critic head backward writes group 1's gradient, then the layer epilogue
    dw_ref, db3_ref = torch.empty(H, device=dev), torch.empty(1, device=dev)
    kernels.value_head_bwd(dy[B:], dw, ddv, dg_a[B:], dw_ref, db3_ref, relu6=True)
    db_ref = torch.empty(2, H, device=dev)
    kernels.act_bwd_bias_grad(dy, dg_a, db_ref, relu6=True, group_rows=B)
    # deferred
    ws, wsv = kernels.learner_workspace(2 * B, H, dev), kernels.learner_workspace(B, H, dev)
    cpg = kernels.act_bwd_partial(dy, dg_b, B, ws, relu6=True, value=(1, dw, ddv, wsv))
    db, dw3, db3 = torch.zeros(2, H, device=dev), torch.zeros(H, device=dev), torch.zeros(1, device=dev)
    S = 197
    s = to_dev(dev, (rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F))[0]
    mean, std = to_dev(dev, rng.normal(0, 0.1, S).astype(F), (1 + rng.random(S)).astype(F))
    step = torch.tensor([5.0], device=dev)
    wsm = kernels.learner_workspace(B, S, dev)
    x = torch.empty_like(s)
    kernels.state_prologue(s, mean, std, 5.0, x, True, ws=wsm)
    m_out, s_out = torch.empty_like(mean), torch.empty_like(std)
    jobs = [kernels.job_colsum(ws, gi * cpg * H, cpg, H, H, db[gi]) for gi in range(2)]
    jobs += [kernels.job_colsum(wsv, 0, cpg, H, H, dw3), kernels.job_vecsum(ddv, db3),
             kernels.job_norm_ema(wsm, B, S, step, mean, std, m_out, s_out)]
    kernels.reduce_jobs(jobs)
    assert torch.equal(dg_a, dg_b)
    assert torch.equal(db, db_ref)
    assert torch.equal(dw3, dw_ref)
    assert torch.equal(db3, db3_ref)
    m_ref, s_ref = kernels.state_norm_update(s, step, mean, std)
    assert torch.equal(m_out, m_ref) and torch.equal(s_out, s_ref)
    # the no-head form of the partial kernel
    dg_c = to_dev(dev, g)[0]
    dg_c[B:].copy_(dg_a[B:])
    cpg2 = kernels.act_bwd_partial(dy, dg_c, B, ws, relu6=True)
    db2 = torch.zeros(2, H, device=dev)
    kernels.reduce_jobs([kernels.job_colsum(ws, gi * cpg2 * H, cpg2, H, H, db2[gi]) for gi in range(2)])
    assert torch.equal(db2, db_ref)


@pytest.mark.parametrize("rows,H,G", [(4096, 2048, 1), (33, 512, 1), (2 * 40, 1024, 2), (7, 256, 1)])
def test_relu6_mask_bits_match_activation_reads(dev, rows, H, G):
    """bias+ReLU6 that leaves the Relu6Grad mask as bits, and the backward reading the bits: the same bits as the pair
    that re-reads the activations (outputs, masked gradient, bias-gradient sums)."""
    from pfpn_amd import kernels
    rng = np.random.default_rng(rows + H)
    y0 = rng.normal(1.5, 3, (rows, H)).astype(F)
    y0[0, :4] = [0.0, 6.0, -0.0, 5.9999995]                      # the edges of the open interval (0, 6)
    bias = rng.normal(0, 1, (G, H)).astype(F)
    bias[0, :4] = 0.0
    g = rng.normal(0, 1, (rows, H)).astype(F)
    gr = rows // G if G > 1 else 0
    ya, yb, db, ga, gb = to_dev(dev, y0, y0, bias, g, g)
    kernels.bias_act(ya, db, True, gr)
    mask = kernels.relu6_mask(rows, H, dev)
    kernels.bias_act_mask(yb, db, mask, gr)
    assert torch.equal(ya, yb)
    # decode the bits on the host and compare with the activations' open-interval test
    m64 = mask.cpu().numpy().view(np.uint64)[: ((rows * H // 4 + 63) // 64) * 4].reshape(-1, 4)
    i = np.arange(rows * H // 4)
    bits = np.stack([(m64[i >> 6, k] >> (i & 63).astype(np.uint64)) & np.uint64(1) for k in range(4)], 1).reshape(rows, H)
    ref = ya.cpu().numpy()
    assert np.array_equal(bits.astype(bool), (ref > 0) & (ref < 6))
    ws1, ws2 = kernels.learner_workspace(rows, H, dev), kernels.learner_workspace(rows, H, dev)
    c1 = kernels.act_bwd_partial(ya, ga, gr, ws1, relu6=True)
    c2 = kernels.act_bwd_partial(None, gb, gr, ws2, mask=mask)
    assert c1 == c2 and torch.equal(ga, gb)
    n = G * c1 * H
    assert torch.equal(ws1.view(torch.float32)[:n], ws2.view(torch.float32)[:n])


def test_prologue_normalises_a_second_batch_with_the_same_moments(dev):
    """pfpn_state_prologue(s2, x2): the second batch (SAC's next states) comes out as pfpn_state_norm would give it, and
    neither the first batch's x nor the float64 moments change."""
    from pfpn_amd import kernels
    rng = np.random.default_rng(15)
    for B, S in ((1024, 197), (33, 7)):
        s = torch.from_numpy((rng.normal(0, 1, (B, S)) * 3 + 0.5).astype(np.float32)).to(dev)
        s2 = torch.from_numpy((rng.normal(0, 1, (B, S)) * 2 - 0.3).astype(np.float32)).to(dev)
        mean = torch.from_numpy(rng.normal(0, 0.3, S).astype(np.float32)).to(dev)
        std = torch.from_numpy(rng.uniform(0.5, 2, S).astype(np.float32)).to(dev)
        step = torch.tensor([5.0], device=dev)
        xa, xb, x2 = torch.empty_like(s), torch.empty_like(s), torch.empty_like(s)
        kernels.state_prologue(s, mean, std, 5.0, xa, True)
        ma, sa = torch.empty_like(mean), torch.empty_like(std)
        kernels.state_norm_ema(B, S, step, mean, std, ma, sa)
        kernels.state_prologue(s, mean, std, 5.0, xb, True, s2=s2, out_x2=x2)
        mb, sb = torch.empty_like(mean), torch.empty_like(std)
        kernels.state_norm_ema(B, S, step, mean, std, mb, sb)
        assert torch.equal(xa, xb) and torch.equal(ma, mb) and torch.equal(sa, sb)
        assert torch.equal(x2, kernels.state_norm(s2, mean, std, 5.0))


def test_soft_target_update_inside_the_adam_launch(dev):
    """pfpn_adam_groups / pfpn_clip_adam_prenorm with soft_target == the same launch followed by pfpn_soft_update on the
    updated parameters, bit for bit; parameters outside [soft_off, soft_off + soft_len) leave the targets alone."""
    import ctypes
    from pfpn_amd import kernels
    rng = np.random.default_rng(16)
    n, off, ln, tau = 30000, 4000, 16000, 0.005
    p0 = torch.from_numpy(rng.normal(0, 1, n).astype(np.float32)).to(dev)
    g0 = torch.from_numpy(rng.normal(0, 0.05, n).astype(np.float32)).to(dev)
    t0 = torch.from_numpy(rng.normal(0, 1, ln).astype(np.float32)).to(dev)
    G = 2
    offs, lens, lrs = (ctypes.c_longlong * G)(0, 12000), (ctypes.c_longlong * G)(12000, 18000), (ctypes.c_float * G)(1e-4, 3e-4)
    res = []
    for fused in (False, True):
        p, g, m, v, t = p0.clone(), g0.clone(), torch.zeros_like(p0), torch.zeros_like(p0), t0.clone()
        bp = torch.tensor([[0.9, 0.999, 0.9, 0.999]] * G, device=dev)
        kernels.adam_groups(p, g, m, v, offs, lens, lrs, bp, 0.5, soft=(t, off, ln, tau) if fused else None)
        if not fused:
            kernels.soft_update(t, p[off:off + ln], tau)
        res.append((p, m, v, t))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert not torch.equal(res[0][3], t0)


def test_column_sum_job_over_two_slab_sets(dev):
    """PFPN_JOB_COLSUM with a second slab set: out = sum(first set) + sum(second set), each summed as a job of its own
    sums it."""
    from pfpn_amd import kernels
    rng = np.random.default_rng(17)
    n, k1, s1, k2, s2 = 1260, 64, 2524, 16, 2520
    w1 = torch.from_numpy(rng.normal(0, 1, k1 * s1).astype(np.float32)).to(dev)
    w2 = torch.from_numpy(rng.normal(0, 1, 100 + k2 * s2).astype(np.float32)).to(dev)
    b1, b2 = w1.view(torch.uint8), w2.view(torch.uint8)
    o1, o2, o12 = (torch.empty(n, device=dev) for _ in range(3))
    kernels.reduce_jobs([kernels.job_colsum(b1, 4, k1, s1, n, o1), kernels.job_colsum(b2, 100, k2, s2, n, o2),
                         kernels.job_colsum(b1, 4, k1, s1, n, o12, second=(b2, 100, k2, s2))])
    assert torch.equal(o12, o1 + o2)
    ref = w1.view(k1, s1)[:, 4:4 + n].double().sum(0) + w2[100:].view(k2, s2)[:, :n].double().sum(0)
    torch.testing.assert_close(o12.double(), ref, rtol=1e-5, atol=1e-4)
