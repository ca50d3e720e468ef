"""GPU parity: the small-row dense job kernel (csrc/rowdense_kernels.hip; reference networks/ops.py:82-118 fc_layer and its
tf.gradients expansion) against float64 matmul -- every product kind, every tile arrangement, unaligned operands, ragged
edges, fused epilogues, the scalar-head helpers, several jobs in one launch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# the tolerance: an fp32 dot product of length K carries <= ~K * 2^-24 * sum|a||b| rounding error whatever its order
# (measured ~1e-7 * sum|a||b|); bound each element by 4e-7 * (|A| |B|)[i, j] + a floor
def _close(out, ref, bound):
    err = (out.double() - ref).abs()
    lim = 4e-7 * bound + 1e-7
    bad = err > lim
    assert not bad.any(), "max err %.3e at limit %.3e (%d elements)" % (
        float(err[bad].max()), float(lim[bad].min()), int(bad.sum()))


SHAPES = [  # G, M, K, N
    (1, 512, 197, 2048), (2, 512, 1024, 512), (1, 512, 512, 1260), (1, 32, 197, 2048), (2, 32, 1024, 512),
    (1, 1, 512, 1260), (2, 100, 236, 132), (1, 1000, 64, 36), (2, 257, 33, 68), (1, 64, 8, 4), (1, 7, 5, 12),
]


def _rand(dev, seed, *shape, scale=1.0):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.randn(*shape, device=dev, generator=g) * scale


@pytest.mark.parametrize("G,M,K,N", SHAPES)
@pytest.mark.parametrize("cfg", [0, 1, 2, -1])
def test_nn_bias_relu6(dev, G, M, K, N, cfg):
    from pfpn_amd import kernels
    x = _rand(dev, 1, G, M, K)
    w = _rand(dev, 2, G, K, N, scale=0.3)
    b = _rand(dev, 3, G, N)
    out = torch.full((G, M, N), float("nan"), device=dev)
    kernels.dense_jobs([kernels.dense_job_nn(x, w, out, bias=b, relu6=True, cfg=cfg)])
    ref = (torch.bmm(x.double(), w.double()) + b.double().unsqueeze(1))
    bound = torch.bmm(x.double().abs(), w.double().abs()) + b.double().abs().unsqueeze(1)
    _close(out, ref.clamp(0, 6), bound)
    # no epilogue, shared input
    out2 = torch.full((G, M, N), float("nan"), device=dev)
    kernels.dense_jobs([kernels.dense_job_nn(x[0], w, out2, cfg=cfg)])
    _close(out2, torch.bmm(x[:1].expand(G, M, K).double(), w.double()),
           torch.bmm(x[:1].expand(G, M, K).double().abs(), w.double().abs()))


def test_nn_identity_layout(dev):
    """A = I with an ASYMMETRIC B catches row/col swaps of the MFMA C/D map; exact."""
    from pfpn_amd import kernels
    for cfg in (0, 1, 2):
        K = N = 128
        x = torch.eye(K, device=dev)
        w = (torch.arange(K * N, device=dev, dtype=torch.float32).view(K, N) % 997) * 0.25
        out = torch.empty(K, N, device=dev)
        kernels.dense_jobs([kernels.dense_job_nn(x, w, out, cfg=cfg)])
        assert torch.equal(out, w)


@pytest.mark.parametrize("G,M,K,N", SHAPES)
@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_nt_mask(dev, G, M, K, N, cfg):
    from pfpn_amd import kernels
    if K % 4:
        pytest.skip("NT needs K % 4 == 0 (dY rows are activations / logits: always padded to 4)")
    a = _rand(dev, 4, G, M, K)
    bt = _rand(dev, 5, G, N, K, scale=0.3)
    h = (_rand(dev, 6, G, M, N) * 4).clamp(0, 6)
    out = torch.full((G, M, N), float("nan"), device=dev)
    kernels.dense_jobs([kernels.dense_job_nt(a, bt, out, mask=h, cfg=cfg)])
    ref = torch.bmm(a.double(), bt.double().transpose(1, 2))
    bound = torch.bmm(a.double().abs(), bt.double().abs().transpose(1, 2))
    keep = ((h > 0) & (h < 6)).double()
    _close(out, ref * keep, bound)
    kernels.dense_jobs([kernels.dense_job_nt(a, bt, out, cfg=cfg)])
    _close(out, ref, bound)


@pytest.mark.parametrize("G,M,K,N", SHAPES)
@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_tn_colsum(dev, G, M, K, N, cfg):
    """dW = X^T dY: the reduction runs over the rows (here K plays the batch)."""
    from pfpn_amd import kernels
    rows, Kd = M, K                      # X [rows, Kd], dY [rows, N] -> dW [Kd, N]
    x = _rand(dev, 7, G, rows, Kd)
    dy = _rand(dev, 8, G, rows, N)
    dw = torch.full((G, Kd, N), float("nan"), device=dev)
    db = torch.full((G, N), float("nan"), device=dev)
    kernels.dense_jobs([kernels.dense_job_tn(x, dy, dw, colsum=db, cfg=cfg)])
    _close(dw, torch.bmm(x.double().transpose(1, 2), dy.double()),
           torch.bmm(x.double().abs().transpose(1, 2), dy.double().abs()))
    _close(db, dy.double().sum(1), dy.double().abs().sum(1))


def test_strided_groups_like_the_merged_first_layer(dev):
    """Layer 1 reads layer 0's merged output [B, G*H0] through the group-strided view (lda = G*H0, group stride H0) and
    writes its input gradient back the same way."""
    from pfpn_amd import kernels
    B, G, H0, H1 = 96, 2, 1024, 512
    act0 = _rand(dev, 9, B, G * H0).clamp(0, 6)
    a0 = act0.view(B, G, H0).transpose(0, 1)             # [G, B, H0] strided
    w1 = _rand(dev, 10, G, H0, H1, scale=0.05)
    b1 = _rand(dev, 11, G, H1)
    act1 = torch.empty(G, B, H1, device=dev)
    kernels.dense_jobs([kernels.dense_job_nn(a0, w1, act1, bias=b1, relu6=True)])
    ref = (torch.bmm(a0.double(), w1.double()) + b1.double().unsqueeze(1)).clamp(0, 6)
    _close(act1, ref, torch.bmm(a0.double().abs(), w1.double().abs()) + b1.double().abs().unsqueeze(1))
    dy1 = _rand(dev, 12, G, B, H1)
    dy0 = torch.full((B, G * H0), float("nan"), device=dev)
    d0 = dy0.view(B, G, H0).transpose(0, 1)
    gw = torch.empty(G, H0, H1, device=dev)
    gb = torch.empty(G, H1, device=dev)
    kernels.dense_jobs([kernels.dense_job_tn(a0, dy1, gw, colsum=gb), kernels.dense_job_nt(dy1, w1, d0, mask=a0)])
    keep = ((a0 > 0) & (a0 < 6)).double()
    _close(d0, torch.bmm(dy1.double(), w1.double().transpose(1, 2)) * keep,
           torch.bmm(dy1.double().abs(), w1.double().abs().transpose(1, 2)))
    _close(gw, torch.bmm(a0.double().transpose(1, 2), dy1.double()),
           torch.bmm(a0.double().abs().transpose(1, 2), dy1.double().abs()))
    _close(gb, dy1.double().sum(1), dy1.double().abs().sum(1))
    assert not torch.isnan(dy0).any()


@pytest.mark.parametrize("M", [1, 32, 100, 512])
@pytest.mark.parametrize("hgroup", [1, -1])
def test_scalar_head_forward_and_backward(dev, M, hgroup):
    """critic fc3 (actor_critic.py:206-210) / twin Q heads (sac.py:123-139) riding on the last trunk layer."""
    from pfpn_amd import kernels
    G, K, N = 2, 256, 512
    x = _rand(dev, 13, G, M, K)
    w = _rand(dev, 14, G, K, N, scale=0.1)
    b = _rand(dev, 15, G, N)
    hw = _rand(dev, 16, G, N) if hgroup < 0 else _rand(dev, 16, N)
    hb = _rand(dev, 17, G, 1) if hgroup < 0 else _rand(dev, 17, 1)
    P = (N + 31) // 32
    act = torch.empty(G, M, N, device=dev)
    part = torch.full((G, M, P) if hgroup < 0 else (M, P), float("nan"), device=dev)
    v = torch.full((G, M) if hgroup < 0 else (M,), float("nan"), device=dev)
    kernels.dense_jobs([kernels.dense_job_nn(x, w, act, bias=b, relu6=True, head=(hgroup, hw, part))])
    kernels.dense_jobs([kernels.dense_job_head_final(part, hb, v)])
    a64 = (torch.bmm(x.double(), w.double()) + b.double().unsqueeze(1)).clamp(0, 6)
    if hgroup < 0:
        ref = (a64 * hw.double().unsqueeze(1)).sum(-1) + hb.double()
        bound = (a64.abs() * hw.double().abs().unsqueeze(1)).sum(-1) + hb.double().abs()
    else:
        ref = (a64[hgroup] * hw.double()).sum(-1) + hb.double()
        bound = (a64[hgroup].abs() * hw.double().abs()).sum(-1) + hb.double().abs()
    _close(v, ref, bound * 8)           # (the activations themselves carry fp32 error into the head)
    # backward helpers
    dv = _rand(dev, 18, G, M) if hgroup < 0 else _rand(dev, 18, M)
    h = act if hgroup < 0 else act[hgroup]
    dh = torch.full_like(h, float("nan"))
    dw = torch.full_like(hw, float("nan"))
    dbh = torch.full_like(hb, float("nan"))
    kernels.dense_jobs([kernels.dense_job_rank1(dv, hw, h, dh), kernels.dense_job_head_wgrad(dv, h, dw, dbh)])
    keep = ((h > 0) & (h < 6)).double()
    ref_dh = dv.double().unsqueeze(-1) * hw.double().unsqueeze(-2) * keep
    assert torch.allclose(dh.double(), ref_dh, rtol=1e-6, atol=0)
    ref_dw = (dv.double().unsqueeze(-1) * h.double()).sum(-2)
    _close(dw, ref_dw, (dv.double().abs().unsqueeze(-1) * h.double().abs()).sum(-2))
    _close(dbh.view(-1), dv.double().sum(-1).view(-1), dv.double().abs().sum(-1).view(-1))


def test_bitwise_reproducible_and_cfg_independent_of_launch_mates(dev):
    """same job alone or beside other jobs, launched twice: identical bits (fixed summation order)."""
    from pfpn_amd import kernels
    x = _rand(dev, 19, 512, 512)
    w = _rand(dev, 20, 512, 1260, scale=0.1)
    dl = _rand(dev, 21, 512, 1260)
    o1, o2 = torch.empty(512, 1260, device=dev), torch.empty(512, 1260, device=dev)
    g1, g2 = torch.empty(512, 1260, device=dev), torch.empty(512, 1260, device=dev)
    dx = torch.empty(512, 512, device=dev)
    kernels.dense_jobs([kernels.dense_job_nn(x, w, o1, cfg=1)])
    kernels.dense_jobs([kernels.dense_job_nn(x, w, o2, cfg=1)])
    assert torch.equal(o1, o2)
    kernels.dense_jobs([kernels.dense_job_tn(x, dl, g1, cfg=0)])
    kernels.dense_jobs([kernels.dense_job_nt(dl, w, dx, mask=x, cfg=2), kernels.dense_job_tn(x, dl, g2, cfg=0)])
    assert torch.equal(g1, g2)


def test_argument_validation(dev):
    from pfpn_amd import kernels, _lib
    x = torch.zeros(8, 6, device=dev)
    w = torch.zeros(6, 6, device=dev)          # N % 4 != 0: B rows are not float4-aligned
    out = torch.zeros(8, 6, device=dev)
    with pytest.raises(_lib.PfpnLibraryError):
        kernels.dense_jobs([kernels.dense_job_nn(x, w, out)])
    with pytest.raises(ValueError):
        kernels.dense_job_nn(x, torch.zeros(5, 8, device=dev), out)
    with pytest.raises(ValueError):
        kernels.dense_jobs([])


@pytest.mark.parametrize("B", [32, 256, 600])
def test_learner_step_through_the_family_matches_the_library_path(dev, monkeypatch, B):
    """the whole DPPO step with every dense layer on the job kernel (10 launches) against the same step on PyTorch-ROCm
    GEMMs + epilogue kernels: same losses, gradients and updated parameters up to fp32 summation order."""
    from test_gpu_dppo_step import make_net, make_batch
    from pfpn_amd.models import SyncReplicasApplier
    from pfpn_amd.networks.ops import DenseStack
    monkeypatch.setenv("PFPN_GRAPH_TRAIN", "0")
    res = []
    for rmax in (0, 1 << 20):
        monkeypatch.setattr(DenseStack, "ROW_MAX", rmax)
        net = make_net(dev, seed=3, resample_interval=10 ** 9)
        opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
        rng = np.random.default_rng(B)
        for it in range(2):
            b = make_batch(rng, B)
            b["log_prob"] = np.full(B, -40.0, np.float32)
            loss, _ = net.train(None, opt, [], b["state"], b["action"], b["value"], b["log_prob"], b["advantage"])
        assert net.trunk.small(B) == (rmax > 0)
        res.append((loss, net.store.grad.clone(), net.store.flat.clone(), net.value_buf.clone(),
                    net.train_out["logp"].clone()))
    (l0, g0, p0, v0, lp0), (l1, g1, p1, v1, lp1) = res
    np.testing.assert_allclose([l1[0], l1[2], l1[3]], [l0[0], l0[2], l0[3]], rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(lp1, lp0, rtol=1e-5, atol=0)
    assert float((v1 - v0).abs().max()) <= 1e-6
    # gradients: 1e-4 of the largest entry of the buffer (Relu6Grad masks flip nowhere: both paths mask by the same
    # activations up to rounding, and the batches keep the pre-activations off the kinks with probability ~1)
    scale = float(g0.abs().max())
    assert float((g1 - g0).abs().max()) <= 1e-4 * scale
    assert float((p1 - p0).abs().max()) <= 2e-6
