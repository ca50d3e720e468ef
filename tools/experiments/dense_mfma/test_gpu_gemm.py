"""GPU parity: the hand-written fp32 MFMA dense kernels (fwd / dgrad / wgrad, fused epilogues) vs float64 matmul."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # G, rows, K, N
    (2, 4096, 200, 1024), (2, 4096, 1024, 512), (1, 4096, 512, 1260), (1, 33, 200, 1024), (2, 257, 64, 132),
    (1, 128, 128, 128), (1, 1, 8, 4), (2, 300, 236, 1024), (1, 500, 512, 4),
]


@pytest.mark.parametrize("G,rows,K,N", SHAPES)
@pytest.mark.parametrize("shared", [True, False])
def test_dense_fwd(dev, G, rows, K, N, shared):
    from pfpn_amd import kernels
    g = torch.Generator(device=dev); g.manual_seed(rows + K)
    x = torch.randn(rows, K, device=dev, generator=g) if shared else torch.randn(G, rows, K, device=dev, generator=g)
    w = torch.randn(G, K, N, device=dev, generator=g) * 0.2
    b = torch.randn(G, N, device=dev, generator=g)
    for relu6 in (True, False):
        out = torch.empty(G, rows, N, device=dev)
        kernels.dense_fwd(x, w, b, out, relu6)
        xe = x.unsqueeze(0).expand(G, rows, K) if shared else x
        ref = torch.bmm(xe.double(), w.double()) + b.double().unsqueeze(1)
        if relu6:
            ref = ref.clamp(0, 6)
        torch.testing.assert_close(out.double(), ref, rtol=2e-5, atol=2e-4)


def test_dense_fwd_identity_layout(dev):
    """A = I with an ASYMMETRIC B catches row/col swaps of the MFMA C/D map."""
    from pfpn_amd import kernels
    K = N = 128
    x = torch.eye(K, device=dev)
    w = (torch.arange(K * N, device=dev, dtype=torch.float32).view(1, K, N) % 997) * 0.01
    out = torch.empty(1, K, N, device=dev)
    kernels.dense_fwd(x, w, torch.zeros(1, N, device=dev), out, relu6=False)
    assert torch.equal(out[0], w[0])


@pytest.mark.parametrize("G,rows,K,N", SHAPES)
def test_dense_dgrad(dev, G, rows, K, N):
    from pfpn_amd import kernels
    g = torch.Generator(device=dev); g.manual_seed(rows + N)
    dy = torch.randn(G, rows, N, device=dev, generator=g)
    w = torch.randn(G, K, N, device=dev, generator=g) * 0.2
    hin = (torch.randn(G, rows, K, device=dev, generator=g) * 4).clamp(0, 6)
    ref = torch.bmm(dy.double(), w.double().transpose(1, 2))
    dx = torch.empty(G, rows, K, device=dev)
    kernels.dense_dgrad(dy, w, None, dx)
    torch.testing.assert_close(dx.double(), ref, rtol=2e-5, atol=3e-4)
    kernels.dense_dgrad(dy, w, hin, dx)
    mask = ((hin > 0) & (hin < 6)).double()
    torch.testing.assert_close(dx.double(), ref * mask, rtol=2e-5, atol=3e-4)


@pytest.mark.parametrize("G,rows,K,N", SHAPES)
@pytest.mark.parametrize("shared", [True, False])
def test_dense_wgrad(dev, G, rows, K, N, shared):
    from pfpn_amd import kernels
    g = torch.Generator(device=dev); g.manual_seed(K + N)
    x = torch.randn(rows, K, device=dev, generator=g) if shared else torch.randn(G, rows, K, device=dev, generator=g)
    dy = torch.randn(G, rows, N, device=dev, generator=g)
    dw = torch.empty(G, K, N, device=dev); db = torch.empty(G, N, device=dev)
    kernels.dense_wgrad(x, dy, dw, db)
    xe = x.unsqueeze(0).expand(G, rows, K) if shared else x
    ref = torch.bmm(xe.double().transpose(1, 2), dy.double())
    torch.testing.assert_close(dw.double(), ref, rtol=3e-5, atol=2e-3 * (rows ** 0.5) / 64)
    torch.testing.assert_close(db.double(), dy.double().sum(1), rtol=3e-5, atol=1e-3)
    dw2 = torch.empty_like(dw); db2 = torch.empty_like(db)
    kernels.dense_wgrad(x, dy, dw2, db2)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "split-K slab reduction must be bitwise reproducible"
    kernels.dense_wgrad(x, dy, dw2, None)
    assert torch.equal(dw, dw2)


def test_dense_padded_input(dev):
    """K = 197 is served by zero-padding the input to ldx = 200 and the weight to 200 rows."""
    from pfpn_amd import kernels
    rows, K, Kp, N = 64, 197, 200, 1024
    x = torch.zeros(rows, Kp, device=dev); x[:, :K] = torch.randn(rows, K, device=dev)
    w = torch.zeros(1, Kp, N, device=dev); w[0, :K] = torch.randn(K, N, device=dev) * 0.1
    out = torch.empty(1, rows, N, device=dev)
    kernels.dense_fwd(x, w, torch.zeros(1, N, device=dev), out, relu6=False)
    torch.testing.assert_close(out[0].double(), x[:, :K].double() @ w[0, :K].double(), rtol=2e-5, atol=1e-4)
