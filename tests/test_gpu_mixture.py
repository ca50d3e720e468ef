"""GPU parity: mixture log-prob kernels (through the C ABI) vs the NumPy oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-5   # north_star: fp32 log-prob within 1e-5 rtol


def make_inputs(oracle, B, A, N, seed, tanh_mode=False, spread=1.2):
    rng = np.random.default_rng(seed)
    loc, logstd = oracle.init_particles(A, N, tanh_mode)
    loc = (loc + rng.normal(0, 0.05, loc.shape)).astype(np.float32)
    logstd = (logstd + rng.normal(0, 0.1, logstd.shape)).astype(np.float32)
    logits = rng.normal(0, 1, (B, A, N)).astype(np.float32)
    x = rng.uniform(-spread, spread, (B, A)).astype(np.float32)
    return logits, x, loc, logstd


def to_dev(dev, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


SHAPES = [(1, 36, 35), (5, 36, 35), (7, 36, 35), (8, 36, 35), (64, 36, 35), (257, 36, 35), (4096, 36, 35),
          (33, 6, 50), (19, 36, 16), (40, 3, 8), (9, 5, 35), (3, 1, 35),
          (17, 36, 23), (6, 4, 100), (5, 3, 256), (4, 7, 64), (3, 2, 2),
          # more tiles per block than LDS planes (the loader ring wraps), ragged last tiles
          (6001, 36, 35), (20003, 8, 16), (9001, 12, 8), (5000, 4, 50), (1030, 20, 35),
          # the tile kernel's widest block (A = 44: 11 compute waves) and the first shape beyond it (wave kernel)
          (13, 44, 8), (7, 48, 16), (3, 1, 8)]


@pytest.mark.parametrize("B,A,N", SHAPES)
@pytest.mark.parametrize("tanh_mode", [False, True])
def test_logprob_fwd(oracle, dev, B, A, N, tanh_mode):
    from pfpn_amd import kernels
    if tanh_mode and N <= 3:
        pytest.skip("tanh grid needs N > 3")
    logits, x, loc, logstd = make_inputs(oracle, B, A, N, 100 + B + N, tanh_mode)
    ref_lp, ref_p = oracle.mixture_log_prob(logits, x, loc, logstd, tanh_mode)
    dl, dx, dloc, dls = to_dev(dev, logits, x, loc, logstd)
    lp, p = kernels.logprob_fwd(dl, dx, dloc, dls, tanh_mode, want_pmix=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(p.cpu().numpy(), ref_p, rtol=2e-5, atol=0)
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp, rtol=RTOL, atol=1e-5)


@pytest.mark.parametrize("B,A,N", SHAPES)
@pytest.mark.parametrize("tanh_mode", [False, True])
def test_logprob_bwd(oracle, dev, B, A, N, tanh_mode):
    from pfpn_amd import kernels
    if tanh_mode and N <= 3:
        pytest.skip("tanh grid needs N > 3")
    logits, x, loc, logstd = make_inputs(oracle, B, A, N, 200 + B + N, tanh_mode)
    rng = np.random.default_rng(B * 7 + N)
    g = rng.normal(0, 1, B).astype(np.float32)
    r_dl, r_dloc, r_dls, r_dx = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, g, tanh_mode)
    dl, dx, dloc, dls, dg = to_dev(dev, logits, x, loc, logstd, g)
    o_dl, o_dloc, o_dls, o_dx = kernels.logprob_bwd(dl, dx, dloc, dls, dg, tanh_mode, want_dx=True)
    torch.cuda.synchronize()
    sc = np.abs(r_dl).max() + 1e-30
    # dlogit = q*w*(phi - p): a difference of two terms of the size of `sc`, each good to a few 1e-7 relative
    np.testing.assert_allclose(o_dl.cpu().numpy(), r_dl, rtol=1e-4, atol=3e-6 * sc)
    np.testing.assert_allclose(o_dx.cpu().numpy(), r_dx, rtol=1e-4, atol=1e-5 * (np.abs(r_dx).max() + 1e-30))
    # sums over B of fp32 terms: tolerance scales with the summed magnitude
    for got, ref in ((o_dloc, r_dloc), (o_dls, r_dls)):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * (np.abs(ref).max() + 1e-30))


def _config5_inputs(dev, B, A=128, N=256, seed=5):
    """BASELINE.json config 5 inputs generated on the device (the full tensor is 1 GiB)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    logits = torch.randn(B, A * N, device=dev, generator=g)
    x = torch.rand(B, A, device=dev, generator=g) * 2.0 - 1.0      # inside the grid: sigma = 2/255, p underflows outside
    grid = torch.linspace(-1, 1, N, device=dev)[None, :].expand(A, N)
    loc = (grid + 0.003 * torch.randn(A, N, device=dev, generator=g)).contiguous()
    logstd = (float(np.log(2.0 / (N - 1))) + 0.1 * torch.randn(A, N, device=dev, generator=g)).contiguous()
    dlogp = torch.randn(B, device=dev, generator=g)
    return logits.view(B, A, N), x, loc, logstd, dlogp


def test_wide_kernel_config5_slice_vs_oracle(oracle, dev):
    """config 5 (128 action dims x 256 particles): the wide kernels against the oracle on a 1024-row shard -- every
    wave walks several 4-row groups, all 128 dims, the multi-chunk row split."""
    from pfpn_amd import kernels
    B, A, N = 1024, 128, 256
    logits, x, loc, logstd, g = _config5_inputs(dev, B)
    h = [t.cpu().numpy() for t in (logits, x, loc, logstd, g)]
    ref_lp, ref_p = oracle.mixture_log_prob(*h[:4])
    lp, p = kernels.logprob_fwd(logits, x, loc, logstd, False, want_pmix=True)
    np.testing.assert_allclose(p.cpu().numpy(), ref_p, rtol=2e-5, atol=0)
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp, rtol=RTOL, atol=0)      # |logp| ~ 1e2: pure rtol
    r_dl, r_dloc, r_dls, r_dx = oracle.mixture_log_prob_bwd(*h)
    o_dl, o_dloc, o_dls, o_dx = kernels.logprob_bwd(logits, x, loc, logstd, g, False, want_dx=True)
    sc = np.abs(r_dl).max()
    np.testing.assert_allclose(o_dl.cpu().numpy().reshape(B, A, N), r_dl, rtol=1e-4, atol=3e-6 * sc)
    np.testing.assert_allclose(o_dx.cpu().numpy(), r_dx, rtol=1e-4, atol=1e-5 * np.abs(r_dx).max())
    for got, ref in ((o_dloc, r_dloc), (o_dls, r_dls)):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * np.abs(ref).max())


def test_wide_kernel_config5_full_size_properties(oracle, dev):
    """The full 8192 x 128 x 256 shape (1 GiB of logits, BASELINE.json config 5) through size-independent properties:
    rows are independent (full launch == two half launches, bit for bit, for the per-row outputs), a segment's logit
    gradients sum to zero (softmax Jacobian), the particle gradients of the halves add up to the full ones, and a
    strided sample of rows matches the oracle."""
    from pfpn_amd import kernels
    B, A, N = 8192, 128, 256
    logits, x, loc, logstd, g = _config5_inputs(dev, B, seed=6)
    lp, p = kernels.logprob_fwd(logits, x, loc, logstd, False, want_pmix=True)
    assert bool(torch.isfinite(lp).all()) and bool((p > 0).all())
    dl, dloc, dls = kernels.logprob_bwd(logits, x, loc, logstd, g)
    dl = dl.view(B, A, N)
    seg = dl.sum(-1).abs().max().item()
    mag = dl.abs().sum(-1).max().item()
    assert seg <= 5e-6 * mag, (seg, mag)
    H = B // 2
    halves = []
    for sl in (slice(0, H), slice(H, B)):
        lp_h = kernels.logprob_fwd(logits[sl].contiguous(), x[sl].contiguous(), loc, logstd)
        assert torch.equal(lp_h, lp[sl])
        dl_h, dloc_h, dls_h = kernels.logprob_bwd(logits[sl].contiguous(), x[sl].contiguous(), loc, logstd,
                                                  g[sl].contiguous())
        assert torch.equal(dl_h.view(H, A, N), dl[sl])
        halves.append((dloc_h.double(), dls_h.double()))
    for full, parts in ((dloc, 0), (dls, 1)):
        tot = halves[0][parts] + halves[1][parts]
        err = (full.double() - tot).abs().max().item()
        assert err <= 2e-5 * tot.abs().max().item(), err
    rows = torch.arange(0, B, 64, device=dev)                      # 128 rows spread over the whole batch
    h = [t.cpu().numpy() for t in (logits[rows], x[rows], loc, logstd, g[rows])]
    ref_lp, _ = oracle.mixture_log_prob(*h[:4])
    np.testing.assert_allclose(lp[rows].cpu().numpy(), ref_lp, rtol=RTOL, atol=0)
    r_dl = oracle.mixture_log_prob_bwd(*h)[0]
    np.testing.assert_allclose(dl[rows].cpu().numpy(), r_dl, rtol=1e-4, atol=3e-6 * np.abs(r_dl).max())


@pytest.mark.parametrize("fused", [True, False])
def test_wide_ppo_composition_config5_full_size_vs_oracle(oracle, dev, monkeypatch, fused):
    """Round 5: the N >= 64 learner composition -- [forward + clipped-PPO head in ONE launch] -> backward
    (+ the head's bias plane) -- at BASELINE.json config 5's FULL shape, 8192 x 128 x 256 (1 GiB of logits), against the
    oracle: log-prob and d loss / d value on EVERY row (oracle log-probs computed in 256-row chunks; reference
    networks/utils.py:108-134), the loss sums against oracle.ppo_loss on those log-probs (networks/actor_critic/ppo.py:31-54,
    actor_critic.py:132-184), d loss / d logp on every row, and the backward (utils.py autodiff) on 128 rows spread over the
    batch with the oracle fed the oracle's own d loss / d logp."""
    from _parity import away_from_clip
    from pfpn_amd import kernels
    monkeypatch.setattr(kernels, "WIDE_FUSED", fused)       # the one-launch (2L) form | [forward + head] -> backward (3L)
    B, A, N = 8192, 128, 256
    logits, x, loc, logstd, _ = _config5_inputs(dev, B, seed=8)
    loc_h, ls_h, x_h = loc.cpu().numpy(), logstd.cpu().numpy(), x.cpu().numpy()
    ref_lp = np.concatenate([oracle.mixture_log_prob(logits[i:i + 256].cpu().numpy(), x_h[i:i + 256], loc_h, ls_h)[0]
                             for i in range(0, B, 256)])
    assert np.isfinite(ref_lp).all()
    rng = np.random.default_rng(85)
    lp_old = away_from_clip(ref_lp, (ref_lp + rng.normal(0, 0.15, B)).astype(np.float32))
    adv, v, v_old = (rng.normal(0, 1, B).astype(np.float32) for _ in range(3))
    loss, pl, vl, r_dlogp, r_dv = oracle.ppo_loss(ref_lp, lp_old, adv, v, v_old, 0.2, 0.5, normalize_adv=True)
    d = to_dev(dev, lp_old, adv, v, v_old)
    stats = kernels.adv_stats(d[1])
    lg2, dl2 = logits.view(B, A * N), torch.empty(B, A * N, device=dev)
    out = {"logp": torch.empty(B, device=dev), "dlogits": dl2, "dloc": torch.empty(A, N, device=dev),
           "dlogstd": torch.empty(A, N, device=dev), "dvalue": torch.empty(B, device=dev), "loss3": torch.empty(3, device=dev)}
    dbias = torch.empty(A * N, device=dev)
    kernels.ppo_mixture_fwd_bwd(lg2, x, loc, logstd, d[0], d[1], d[2], d[3], stats, 0.2, 0.5, out=out, dbias=dbias)
    np.testing.assert_allclose(out["logp"].cpu().numpy(), ref_lp, rtol=RTOL, atol=0)              # every row, pure rtol
    np.testing.assert_allclose(out["loss3"].cpu().numpy(), [loss, pl, vl], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["dvalue"].cpu().numpy(), r_dv, rtol=1e-5, atol=1e-10)
    got_dlogp = kernels._dlogp_scratch(B, dev).cpu().numpy()
    # d loss / d logp = -(A_norm / B) * ratio on the rows whose surrogate passes: the ratio carries the log-prob's 1e-5
    np.testing.assert_allclose(got_dlogp, r_dlogp, rtol=2e-3, atol=1e-9)
    assert np.array_equal(got_dlogp != 0, r_dlogp != 0)                                           # same clip decisions
    rows = torch.arange(0, B, 64, device=dev)
    h = [logits[rows].cpu().numpy(), x_h[::64], loc_h, ls_h, r_dlogp[::64]]
    r_dl = oracle.mixture_log_prob_bwd(*h)[0]
    got = out["dlogits"].view(B, A, N)[rows].cpu().numpy()
    np.testing.assert_allclose(got, r_dl, rtol=2e-3, atol=3e-6 * np.abs(r_dl).max())
    # ... and with the kernel's own d loss / d logp (so that only the backward's arithmetic is in the comparison)
    r_dl2 = oracle.mixture_log_prob_bwd(h[0], h[1], h[2], h[3], got_dlogp[::64])[0]
    np.testing.assert_allclose(got, r_dl2, rtol=1e-4, atol=3e-6 * np.abs(r_dl2).max())
    # the sums over the batch: column sums of dlogits == dbias (float64 on the device), softmax Jacobian rows sum to zero
    cs = out["dlogits"].double().sum(0)
    assert (dbias.double() - cs).abs().max().item() <= 1e-4 * cs.abs().max().item()
    assert bool(torch.isfinite(out["dloc"]).all()) and bool(torch.isfinite(out["dlogstd"]).all())


@pytest.mark.parametrize("B,A,N", [(1030, 8, 128), (77, 12, 256), (4099, 4, 512), (517, 128, 256), (9, 16, 64)])
def test_wide_in_launch_handoffs(oracle, dev, B, A, N):
    """Round 4: the wide kernels finish their own reductions (no k_row_sum / k_finalize launch).  Forward: a block's four
    waves add their rows' log p, the last block of a row chunk adds the chunk's blocks in order; backward: the last wave of an
    action dim adds the C slabs in k_finalize's order.  Checked (a) against the oracle, (b) the backward BITWISE against the
    slabs of the un-finalised launch summed on the host in that order, (c) over repeated launches on fresh data (the tickets
    must come back zeroed) with ragged row counts, several float4s per lane and more blocks than CUs."""
    from pfpn_amd import kernels
    for it in range(3):
        logits, x, loc, logstd = make_inputs(oracle, B, A, N, 900 + 7 * it + B, spread=1.0)
        g = np.random.default_rng(it).normal(0, 1, B).astype(np.float32)
        d = to_dev(dev, logits, x, loc, logstd, g)
        ref_lp, _ = oracle.mixture_log_prob(logits, x, loc, logstd)
        lp = kernels.logprob_fwd(d[0], d[1], d[2], d[3])
        ok = np.isfinite(ref_lp)
        assert ok.sum() > B // 2
        got = lp.cpu().numpy()
        assert np.array_equal(np.isfinite(got), ok)
        # (a row's A segment terms are of size ~5-8 each and may cancel: 3e-5 absolute is a few ulps of the terms)
        np.testing.assert_allclose(got[ok], ref_lp[ok], rtol=RTOL, atol=3e-5)
        assert torch.equal(lp, kernels.logprob_fwd(d[0], d[1], d[2], d[3]))                # same bits on the next launch
        o_dl, o_dloc, o_dls = kernels.logprob_bwd(*d)
        ws = kernels.mixture_workspace(B, A, N, dev)
        s_dl, _, _ = kernels.logprob_bwd(*d, slabs_ws=ws)                                   # slabs left unreduced
        assert torch.equal(o_dl, s_dl)
        nslab, stride = kernels.logprob_bwd_slabs(d[0], s_dl, B, A, N)
        slabs = ws.view(torch.float32)[:nslab * stride].view(nslab, stride).cpu().numpy()
        run = np.zeros((16, 2 * A * N), np.float32)
        for k in range(nslab):
            run[k % 16] += slabs[k, :2 * A * N]
        tot = run[0].copy()
        for r in range(1, 16):
            tot += run[r]
        assert np.array_equal(o_dloc.cpu().numpy().reshape(-1), tot[:A * N])
        assert np.array_equal(o_dls.cpu().numpy().reshape(-1), tot[A * N:])
        r_dloc, r_dls = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, g)[1:3]
        for gotp, ref in ((o_dloc, r_dloc), (o_dls, r_dls)):
            np.testing.assert_allclose(gotp.cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * (np.abs(ref).max() + 1e-30))
        # the backward that also leaves the head's bias gradient (a third slab plane): every other output keeps its bits,
        # dbias = the column sums of dlogits
        dbias = torch.full((A * N,), float("nan"), device=dev)
        b_dl, b_dloc, b_dls = kernels.logprob_bwd(*d, dbias=dbias)
        assert torch.equal(b_dl, o_dl) and torch.equal(b_dloc, o_dloc) and torch.equal(b_dls, o_dls)
        r_db = o_dl.view(B, A * N).double().sum(0).cpu().numpy()
        np.testing.assert_allclose(dbias.cpu().numpy(), r_db, rtol=1e-4, atol=2e-5 * (np.abs(r_db).max() + 1e-30))
        dbias2 = torch.empty_like(dbias)
        kernels.logprob_bwd(*d, dbias=dbias2)
        assert torch.equal(dbias, dbias2)


@pytest.mark.parametrize("B,A,N", [(1030, 8, 128), (517, 128, 256)])
def test_wide_handoff_is_reentrant_across_streams(oracle, dev, B, A, N):
    """Round 5 (SURVEY 8b "re-entrant"; include/pfpn_hip.h "Concurrency"): the hand-off tickets of the wide kernels live in the
    CALLER'S workspace, not in the code object.  Forward (+ PPO head) and backward launched on two streams at once, each with
    its own workspace and outputs, 50 rounds, interleaved from the host so that launches of the two streams overlap on the
    chip: every result of every round is bitwise the single-stream one (compared on the device, stream-ordered, no host sync
    inside the rounds)."""
    from pfpn_amd import kernels
    logits, x, loc, logstd = make_inputs(oracle, B, A, N, 4242 + B, spread=1.0)
    rng = np.random.default_rng(B)
    g = rng.normal(0, 1, B).astype(np.float32)
    sets = []
    for k in range(2):                                          # two independent problems (same shape, different data)
        lg = logits if k == 0 else np.roll(logits, 1, axis=0).copy()
        d = to_dev(dev, lg, x, loc, logstd, g)
        ref_lp = kernels.logprob_fwd(d[0], d[1], d[2], d[3]).clone()
        ref = [t.clone() for t in kernels.logprob_bwd(*d)]
        dbias_ref = torch.empty(A * N, device=dev)
        kernels.logprob_bwd(*d, dbias=dbias_ref)
        sets.append((d, ref_lp, ref, dbias_ref))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    outs, bad, wss = [], [], []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            outs.append({"lp": torch.empty(B, device=dev), "dlogits": torch.empty(B, A, N, device=dev),
                         "dloc": torch.empty(A, N, device=dev), "dlogstd": torch.empty(A, N, device=dev),
                         "dbias": torch.empty(A * N, device=dev)})
            bad.append(torch.zeros(1, dtype=torch.int64, device=dev))
            d = sets[k][0]
            kernels.logprob_fwd(d[0], d[1], d[2], d[3], out=outs[k]["lp"])     # (sizes this stream's workspace)
            wss.append(kernels.workspace(1, dev))
    assert wss[0].data_ptr() != wss[1].data_ptr()              # one workspace per stream
    torch.cuda.synchronize()
    for it in range(50):
        for k, st in enumerate(streams):
            d, ref_lp, ref, dbias_ref = sets[k]
            o = outs[k]
            with torch.cuda.stream(st):
                kernels.logprob_fwd(d[0], d[1], d[2], d[3], out=o["lp"])
                bad[k] += (o["lp"] != ref_lp).sum()
                kernels.logprob_bwd(*d, out=o, dbias=o["dbias"] if it % 2 else None)
                bad[k] += (o["dlogits"] != ref[0]).sum() + (o["dloc"] != ref[1]).sum() + (o["dlogstd"] != ref[2]).sum()
                if it % 2:
                    bad[k] += (o["dbias"] != dbias_ref).sum()
    torch.cuda.synchronize()
    assert int(bad[0].item()) == 0 and int(bad[1].item()) == 0, (int(bad[0].item()), int(bad[1].item()))


def test_wide_handoff_repairs_a_poisoned_workspace_and_refuses_a_short_one(oracle, dev):
    """The ticket area needs no initialisation: a workspace full of 0xFF bytes (what torch.empty may hand out, or what a launch
    that died mid-way leaves) gives the bits of a clean one -- the entry point zeroes the tickets it uses in front of its
    launch.  A workspace shorter than pfpn_mixture_workspace_bytes(): the forward still answers (launch-per-piece form, same
    bits to 1e-6), the bias-plane backward -- which exists in the hand-off only -- is refused with PFPN_ERR_WORKSPACE."""
    import ctypes
    from pfpn_amd import _lib, kernels
    B, A, N = 517, 12, 256
    logits, x, loc, logstd = make_inputs(oracle, B, A, N, 99, spread=1.0)
    g = np.random.default_rng(9).normal(0, 1, B).astype(np.float32)
    d = to_dev(dev, logits, x, loc, logstd, g)
    lp0 = kernels.logprob_fwd(d[0], d[1], d[2], d[3]).clone()
    r0 = [t.clone() for t in kernels.logprob_bwd(*d)]
    db0 = torch.empty(A * N, device=dev)
    kernels.logprob_bwd(*d, dbias=db0)
    for fill in (0xFF, 0x01):
        kernels.workspace(1, dev).fill_(fill)                   # the current stream's workspace, tickets included
        assert torch.equal(kernels.logprob_fwd(d[0], d[1], d[2], d[3]), lp0)
        kernels.workspace(1, dev).fill_(fill)
        db = torch.empty(A * N, device=dev)
        r = kernels.logprob_bwd(*d, dbias=db)
        assert all(torch.equal(a, b) for a, b in zip(r, r0)) and torch.equal(db, db0)
    lib = _lib.load()
    need = int(lib.pfpn_mixture_workspace_bytes(B, A, N))
    short = torch.empty(need - 64, dtype=torch.uint8, device=dev)
    lp = torch.empty(B, device=dev)
    st = kernels._stream()
    assert lib.pfpn_logprob_fwd(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), B, A, N, 0, lp.data_ptr(),
                                None, short.data_ptr(), short.numel(), st) == 0
    np.testing.assert_allclose(lp.cpu().numpy(), lp0.cpu().numpy(), rtol=2e-6, atol=2e-5)
    dl, dloc, dls, db = (torch.empty(B, A * N, device=dev), torch.empty(A, N, device=dev), torch.empty(A, N, device=dev),
                         torch.empty(A * N, device=dev))
    rc = lib.pfpn_logprob_bwd_bias(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), B, A, N,
                                   0, dl.data_ptr(), dloc.data_ptr(), dls.data_ptr(), None, db.data_ptr(), short.data_ptr(),
                                   short.numel(), st)
    assert rc == 3                                              # PFPN_ERR_WORKSPACE
    # ... and the plain backward with the short workspace takes the k_finalize launch: same bits (same summation order)
    assert lib.pfpn_logprob_bwd(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), B, A, N, 0,
                                dl.data_ptr(), dloc.data_ptr(), dls.data_ptr(), None, 1, short.data_ptr(), short.numel(), st) == 0
    assert torch.equal(dl.view(B, A, N), r0[0]) and torch.equal(dloc, r0[1]) and torch.equal(dls, r0[2])


@pytest.mark.parametrize("B,A,N", [(517, 128, 256), (1030, 8, 256), (64, 12, 256), (4099, 4, 256),
                                   (1030, 8, 128), (333, 20, 64), (77, 12, 100),         # (N < 256: the lanes past N / 4 idle)
                                   (33, 8, 256), (35, 4, 128)])       # (8 chunks of 5 consecutive rows: the last one is EMPTY)
@pytest.mark.parametrize("safe", [0, 1, 2])
def test_wide_fused_one_launch_ppo_pass(oracle, dev, monkeypatch, B, A, N, safe):
    """Round 5: the 256-particle PPO pass as ONE launch that reads the logits once (pfpn_ppo_wide_fused; the waves of a row
    chunk exchange their per-dimension log p through memory once per four rows) against the oracle on EVERY row and output,
    against the two-launch composition (equal to summation order), bitwise run to run, from a poisoned workspace -- with the
    exchange inside the chunk's XCD (the placement check passes); `safe` = 1: with the memory-side exchange forced
    (PFPN_WIDE_F2L_SAFE); `safe` = 2: with every chunk made of CONSECUTIVE blocks, i.e. spread over all eight XCDs -- the
    in-kernel placement check must notice by itself and the memory-side exchange must carry the launch ACROSS XCDs."""
    from _parity import away_from_clip
    from pfpn_amd import kernels
    if safe:
        monkeypatch.setenv("PFPN_WIDE_F2L_SAFE", str(safe))
    else:
        monkeypatch.delenv("PFPN_WIDE_F2L_SAFE", raising=False)
    logits, x, loc, logstd = make_inputs(oracle, B, A, N, 5000 + B, spread=1.0)
    rng = np.random.default_rng(B + A)
    ref_lp, _ = oracle.mixture_log_prob(logits, x, loc, logstd)
    assert np.isfinite(ref_lp).all()
    lp_old = away_from_clip(ref_lp, (ref_lp + rng.normal(0, 0.15, B)).astype(np.float32))
    adv, v, v_old = (rng.normal(0, 1, B).astype(np.float32) for _ in range(3))
    loss, pl, vl, r_dlogp, r_dv = oracle.ppo_loss(ref_lp, lp_old, adv, v, v_old, 0.2, 0.5, normalize_adv=True)
    r_dl, r_dloc, r_dls, _ = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, r_dlogp)
    d = to_dev(dev, logits, x, loc, logstd, lp_old, adv, v, v_old)
    stats = kernels.adv_stats(d[5])

    def run(fill=None):
        if fill is not None:
            kernels.workspace(1, dev).fill_(fill)
        dbias = torch.full((A * N,), float("nan"), device=dev)
        out, chunks, err = kernels.ppo_wide_fused(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], stats, 0.2, 0.5, dbias=dbias)
        torch.cuda.synchronize()
        assert int(err.item()) == 0 and chunks % 8 == 0 and chunks >= 8
        return {k: t.clone() for k, t in out.items()}, dbias
    o, db = run()
    np.testing.assert_allclose(o["logp"].cpu().numpy(), ref_lp, rtol=RTOL, atol=3e-5)
    np.testing.assert_allclose(o["loss3"].cpu().numpy(), [loss, pl, vl], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(o["dvalue"].cpu().numpy(), r_dv, rtol=1e-5, atol=1e-9)
    got_dlogp = o["dlogp"].cpu().numpy()
    assert np.array_equal(got_dlogp != 0, r_dlogp != 0)
    np.testing.assert_allclose(got_dlogp, r_dlogp, rtol=2e-3, atol=1e-9)
    # the backward's own arithmetic: the oracle fed the kernel's d loss / d logp
    r2 = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, got_dlogp)
    sc = np.abs(r2[0]).max() + 1e-30
    np.testing.assert_allclose(o["dlogits"].cpu().numpy().reshape(B, A, N), r2[0], rtol=1e-4, atol=3e-6 * sc)
    for k, ref in (("dloc", r2[1]), ("dlogstd", r2[2])):
        np.testing.assert_allclose(o[k].cpu().numpy(), ref, rtol=1e-4, atol=5e-5 * (np.abs(ref).max() + 1e-30))
    r_db = o["dlogits"].double().sum(0).cpu().numpy()
    np.testing.assert_allclose(db.cpu().numpy(), r_db, rtol=1e-4, atol=2e-5 * (np.abs(r_db).max() + 1e-30))
    # against the two-launch composition
    monkeypatch.setattr(kernels, "WIDE_FUSED", False)
    c = kernels.ppo_mixture_fwd_bwd(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], stats, 0.2, 0.5)
    for k in ("logp", "dvalue", "dloc", "dlogstd", "loss3"):
        ref = c[k].double()
        assert float((o[k].double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-12, k
    assert float((o["dlogits"].double() - c["dlogits"].double()).abs().max()) <= 1e-4 * float(c["dlogits"].abs().max())
    # run to run, and from a workspace full of 0xFF / 0x01 bytes: the same bits
    for fill in (None, 0xFF, 0x01):
        o2, db2 = run(fill)
        assert all(torch.equal(o[k], o2[k]) for k in o) and torch.equal(db, db2), fill


def test_wide_fused_pass_fails_loudly_when_its_waves_cannot_meet(oracle, dev, monkeypatch):
    """The one-launch pass's waves wait for each other.  If a chunk's partners are NOT all resident -- here forced: a grid twice
    the device's capacity (PFPN_WIDE_F2L_OVERSUBSCRIBE) whose chunks take every C-th block (PFPN_WIDE_F2L_SAFE=3), so that half
    of every chunk's partners wait behind the resident half -- the bounded waits expire: the launch ENDS (no hang), the flag is
    set, the gradients are NaN (the step's global norm goes NaN: loud), and the next normal launch is clean and bitwise the
    one before.  (With the shipped block -> chunk mapping even the oversubscribed grid is fine: a chunk is complete within
    8 * A / 4 consecutive blocks of the dispatch order.)"""
    from pfpn_amd import kernels
    B, A, N = 2048, 128, 256
    logits, x, loc, logstd, _ = _config5_inputs(dev, B, seed=21)
    lp0 = kernels.logprob_fwd(logits, x, loc, logstd).clone()
    g = torch.Generator(device=dev).manual_seed(3)
    adv = torch.randn(B, device=dev, generator=g)
    v = torch.zeros(B, device=dev)
    stats = kernels.adv_stats(adv)

    def run():
        out, chunks, err = kernels.ppo_wide_fused(logits.view(B, A * N), x, loc, logstd, lp0, adv, v, v, stats)
        torch.cuda.synchronize()
        return {k: t.clone() for k, t in out.items()}, chunks, int(err.item())
    o0, c0, e0 = run()
    assert e0 == 0 and bool(torch.isfinite(o0["dloc"]).all())
    monkeypatch.setenv("PFPN_WIDE_F2L_OVERSUBSCRIBE", "1")
    o1, c1, e1 = run()                                       # shipped mapping, oversubscribed: chunks complete in dispatch order
    assert e1 == 0 and c1 == 2 * c0 and bool(torch.isfinite(o1["dloc"]).all())
    np.testing.assert_allclose(o1["logp"].cpu().numpy(), o0["logp"].cpu().numpy(), rtol=1e-6, atol=1e-5)
    monkeypatch.setenv("PFPN_WIDE_F2L_SAFE", "3")
    o2, c2, e2 = run()
    assert e2 == 1 and c2 == 2 * c0 and int(torch.isnan(o2["dloc"]).sum().item()) > 0       # loud, and it ended
    monkeypatch.delenv("PFPN_WIDE_F2L_OVERSUBSCRIBE")
    monkeypatch.delenv("PFPN_WIDE_F2L_SAFE")
    o3, c3, e3 = run()
    assert e3 == 0 and c3 == c0 and all(torch.equal(o0[k], o3[k]) for k in o0)
    # the per-launch flag is gone with the next launch (e3 == 0), the CALLER'S fault word is not: the expired launch stays on
    # record however many clean launches follow, and the package raises at its next host synchronisation (loss read-back,
    # end of a training round, checkpoint save) instead of training on with NaN parameters
    assert int(kernels.fault_word(dev).item()) == 1
    with pytest.raises(kernels.DeviceFault, match="PFPN_WIDE_FUSED=0"):
        kernels.check_faults()
    with pytest.raises(kernels.DeviceFault):
        from pfpn_amd import checkpoint
        checkpoint.network_tensors(None)                     # (refuses before it looks at the network)
    kernels.clear_faults()
    kernels.check_faults()


def test_wide_fused_refuses_a_chunk_beyond_its_buffer_window(dev):
    """A wave addresses its chunk's rows through ONE 32-bit buffer window: a batch whose chunks would span 2 GiB or more
    (here 2^19 rows of 128 x 256: 32768 rows = 4 GiB per chunk at 16 chunks) is refused with PFPN_ERR_ARG before anything is
    launched or touched (the caller composes the two launches) -- the pointers below are small dummies, never dereferenced."""
    import ctypes
    from pfpn_amd import _lib, kernels
    lib = _lib.load()
    B, A, N = 1 << 19, 128, 256
    assert (B // 32) * A * N * 4 >= 1 << 31                  # even at the largest chunk count the form takes
    t = torch.zeros(64, device=dev)
    nbytes = int(lib.pfpn_mixture_workspace_bytes(B, A, N))
    ws = kernels.workspace(nbytes, dev)
    p = ctypes.c_void_p(t.data_ptr())
    chunks = ctypes.c_int(-1)
    rc = lib.pfpn_ppo_wide_fused(p, p, p, p, p, p, p, p, p, B, A, N, ctypes.c_float(0.2), ctypes.c_float(0.5), p, p, p, p, p, p,
                                 None, p, ctypes.addressof(chunks), None, ctypes.c_void_p(ws.data_ptr()),
                                 ctypes.c_size_t(ws.numel()), None)
    torch.cuda.synchronize()
    assert rc == 1 and chunks.value == -1 and float(t.abs().sum()) == 0.0


def test_wide_fused_is_what_the_learner_composition_takes(oracle, dev, monkeypatch):
    """ppo_mixture_fwd_bwd at a 256-particle shape takes the one-launch form by itself (PFPN_WIDE_FUSED=1, the default) and the
    two-launch composition where the form does not exist (a batch of fewer than 8 row chunks; A % 4 != 0)."""
    from pfpn_amd import _lib, kernels
    assert kernels.WIDE_FUSED
    calls = []
    lib = _lib.load()
    real = lib.pfpn_ppo_wide_fused

    class Spy(object):
        def __getattr__(self, name):
            if name == "pfpn_ppo_wide_fused":
                def f(*a):
                    rc = real(*a)
                    calls.append(rc)
                    return rc
                return f
            return getattr(lib, name)
    monkeypatch.setattr(_lib, "load", lambda: Spy())
    for (B, A, N), want in (((517, 8, 256), 0), ((9, 8, 256), 1), ((200, 6, 256), 1)):
        logits, x, loc, logstd = make_inputs(oracle, B, A, N, 31 + B, spread=1.0)
        rng = np.random.default_rng(B)
        ref_lp, _ = oracle.mixture_log_prob(logits, x, loc, logstd)
        lp_old = (ref_lp + rng.normal(0, 0.1, B)).astype(np.float32)
        adv, v, v_old = (rng.normal(0, 1, B).astype(np.float32) for _ in range(3))
        d = to_dev(dev, logits, x, loc, logstd, lp_old, adv, v, v_old)
        out = kernels.ppo_mixture_fwd_bwd(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], kernels.adv_stats(d[5]), 0.2, 0.5)
        assert calls[-1] == want, ((B, A, N), calls)
        np.testing.assert_allclose(out["logp"].cpu().numpy(), ref_lp, rtol=RTOL, atol=3e-5)


@pytest.mark.parametrize("fused", [False, True])
def test_wide_ppo_composition_is_two_launches_into_the_callers_buffers(oracle, dev, monkeypatch, fused):
    """N >= 64: ppo_mixture_fwd_bwd composes [forward + PPO head in one launch] -> backward (`fused` = False; with the
    one-launch form of round 5 on -- 64 <= N <= 256 -- the same call is ONE launch); every result lands in `out` directly;
    the in-launch head gives the per-row outputs of the stand-alone head bit for bit."""
    from pfpn_amd import kernels
    monkeypatch.setattr(kernels, "WIDE_FUSED", fused)
    B, A, N = 300, 8, 128
    logits, x, loc, logstd = make_inputs(oracle, B, A, N, 77, spread=1.0)
    rng = np.random.default_rng(3)
    ref_lp, _ = oracle.mixture_log_prob(logits, x, loc, logstd)
    lp_old = (ref_lp + rng.normal(0, 0.15, B)).astype(np.float32)
    adv, v, v_old = (rng.normal(0, 1, B).astype(np.float32) for _ in range(3))
    loss, pl, vl, dlogp, dv = oracle.ppo_loss(ref_lp, lp_old, adv, v, v_old, 0.2, 0.5, normalize_adv=True)
    r_dl, r_dloc, r_dls, _ = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, dlogp)
    d = to_dev(dev, logits, x, loc, logstd, lp_old, adv, v, v_old)
    stats = kernels.adv_stats(d[5])
    out = {"logp": torch.empty(B, device=dev), "dlogits": torch.empty(B, A * N, device=dev),
           "dloc": torch.empty(A, N, device=dev), "dlogstd": torch.empty(A, N, device=dev),
           "dvalue": torch.empty(B, device=dev), "loss3": torch.empty(3, device=dev)}
    ptrs = {k: t.data_ptr() for k, t in out.items()}
    res = kernels.ppo_mixture_fwd_bwd(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], stats, 0.2, 0.5, out=out)
    assert all(res[k].data_ptr() == ptrs[k] for k in ptrs)
    np.testing.assert_allclose(out["logp"].cpu().numpy(), ref_lp, rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(out["loss3"].cpu().numpy(), [loss, pl, vl], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(out["dvalue"].cpu().numpy(), dv, rtol=1e-5, atol=1e-9)
    sc = np.abs(r_dl).max() + 1e-30
    np.testing.assert_allclose(out["dlogits"].cpu().numpy().reshape(B, A, N), r_dl, rtol=2e-4, atol=3e-6 * sc)
    for k, ref in (("dloc", r_dloc), ("dlogstd", r_dls)):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref, rtol=1e-4, atol=5e-5 * (np.abs(ref).max() + 1e-30))
    # the head inside the forward launch == the stand-alone head on the same log-probs (sums: another order)
    h_dlogp, h_dv, h_l3 = kernels.ppo_loss(out["logp"], d[4], d[5], d[6], d[7], stats, 0.2, 0.5)
    assert torch.equal(kernels._dlogp_scratch(B, dev), h_dlogp) and torch.equal(out["dvalue"], h_dv)
    np.testing.assert_allclose(out["loss3"].cpu().numpy(), h_l3.cpu().numpy(), rtol=2e-6, atol=1e-7)
    again = kernels.ppo_mixture_fwd_bwd(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], stats, 0.2, 0.5)
    for k in out:
        assert torch.equal(out[k], again[k]), k                # the tickets came back zeroed; fixed-order sums


def test_stream_pair_placement_changes_no_bits(dev):
    """logits / dlogits of large shapes come out of ONE allocation at a fixed distance (kernels.stream_pair: where the two
    streams of the backward sit relative to each other decides its speed); smaller shapes get two plain
    allocations.  Placement must not change a single bit of the results."""
    from pfpn_amd import kernels
    B, A, N = 2048, 128, 256                                        # 256 MiB of logits: the first paired size
    lg, dl = kernels.stream_pair(B, A * N, dev)
    assert lg.untyped_storage().data_ptr() == dl.untyped_storage().data_ptr()
    assert (dl.data_ptr() - lg.data_ptr()) % (256 << 20) == 4096 and dl.data_ptr() - lg.data_ptr() >= lg.numel() * 4
    assert lg.is_contiguous() and dl.is_contiguous() and lg.shape == dl.shape == (B, A * N)
    s1, s2 = kernels.stream_pair(B // 2, A * N, dev)
    assert s1.untyped_storage().data_ptr() != s2.untyped_storage().data_ptr()
    src, x, loc, logstd, g = _config5_inputs(dev, B, seed=8)
    lg.copy_(src.view(B, A * N))
    outs = []
    for dlogits in (dl, torch.empty(B, A * N, device=dev)):
        o = {"dlogits": dlogits, "dloc": torch.empty(A, N, device=dev), "dlogstd": torch.empty(A, N, device=dev)}
        kernels.logprob_bwd(lg.view(B, A, N), x, loc, logstd, g, out=o)
        outs.append(o)
    for k in ("dlogits", "dloc", "dlogstd"):
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert bool(torch.isfinite(outs[0]["dlogits"]).all())


@pytest.mark.parametrize("A,N", [(36, 35), (8, 256), (4, 100)])
def test_row_maximum_found_wherever_it_sits(oracle, dev, A, N):
    """The softmax shift is a cross-lane maximum (DPP / lane-swap reductions written partly in inline asm).  ANY finite shift
    gives the same result in exact arithmetic, so a wrong maximum would pass every ordinary comparison -- except where the
    spread is large: one logit at +60 among -60s overflows exp() unless the maximum really is that logit.  The hot particle
    visits every position (every lane and register of the reduction) and sits under the action so that p stays normal."""
    from pfpn_amd import kernels
    B = N                                                            # row b: hot particle b % N
    loc = np.tile(np.linspace(-1, 1, N, dtype=np.float32), (A, 1))
    logstd = np.full((A, N), np.log(2.0 / (N - 1)), np.float32)
    logits = np.full((B, A, N), -60.0, np.float32)
    hot = (np.arange(B)[:, None] + np.arange(A)[None, :]) % N        # a different position per segment of a row too
    np.put_along_axis(logits, hot[..., None], 60.0, axis=-1)
    x = np.take_along_axis(np.broadcast_to(loc, (B, A, N)), hot[..., None], axis=-1)[..., 0].astype(np.float32)
    g = np.ones(B, np.float32)
    t = [torch.from_numpy(a).to(dev) for a in (logits, x, loc, logstd, g)]
    ref_lp, ref_p = oracle.mixture_log_prob(logits, x, loc, logstd)
    lp, p = kernels.logprob_fwd(t[0], t[1], t[2], t[3], False, want_pmix=True)
    assert np.isfinite(ref_lp).all() and bool(torch.isfinite(lp).all())
    np.testing.assert_allclose(p.cpu().numpy(), ref_p, rtol=2e-5, atol=0)
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp, rtol=RTOL, atol=1e-5)
    r_dl = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, g)[0]
    dl = kernels.logprob_bwd(t[0], t[1], t[2], t[3], t[4])[0].view(B, A, N)
    assert bool(torch.isfinite(dl).all())
    np.testing.assert_allclose(dl.cpu().numpy(), r_dl, rtol=1e-4, atol=3e-6 * max(np.abs(r_dl).max(), 1e-30))


def test_logprob_underflow_guard(oracle, dev):
    """x far from every particle: p underflows to 0, logp = -inf in the forward (reference computes in
    probability space), and the backward guard zeroes the gradients (utils.py:109-117)."""
    from pfpn_amd import kernels
    B, A, N = 9, 36, 35
    logits, x, loc, logstd = make_inputs(oracle, B, A, N, 5)
    x[3, 7] = 40.0
    x[5, :] = -30.0
    g = np.ones(B, np.float32)
    ref_lp, _ = oracle.mixture_log_prob(logits, x, loc, logstd)
    r_dl, r_dloc, r_dls, _ = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, g)
    assert np.isneginf(ref_lp[3]) and np.isneginf(ref_lp[5])
    dl, dx, dloc, dls, dg = to_dev(dev, logits, x, loc, logstd, g)
    lp = kernels.logprob_fwd(dl, dx, dloc, dls)
    o_dl, o_dloc, o_dls = kernels.logprob_bwd(dl, dx, dloc, dls, dg)
    lp = lp.cpu().numpy()
    assert np.isneginf(lp[3]) and np.isneginf(lp[5])
    ok = np.isfinite(ref_lp)
    np.testing.assert_allclose(lp[ok], ref_lp[ok], rtol=RTOL, atol=1e-5)
    o = o_dl.cpu().numpy()
    assert np.all(np.isfinite(o))
    assert np.all(o[3, 7] == 0) and np.all(o[5] == 0)
    np.testing.assert_allclose(o, r_dl, rtol=1e-4, atol=1e-6 * np.abs(r_dl).max())
    assert np.all(np.isfinite(o_dloc.cpu().numpy())) and np.all(np.isfinite(o_dls.cpu().numpy()))


@pytest.mark.parametrize("B", [1, 7, 100, 4096, 6001])
def test_ppo_fused(oracle, dev, B):
    """Fused fwd + PPO loss + bwd kernel == oracle log_prob -> ppo_loss -> log_prob_bwd."""
    from pfpn_amd import kernels
    A, N = 36, 35
    logits, x, loc, logstd = make_inputs(oracle, B, A, N, 300 + B, spread=1.05)
    rng = np.random.default_rng(B)
    ref_lp, _ = oracle.mixture_log_prob(logits, x, loc, logstd)
    lp_old = (ref_lp + rng.normal(0, 0.15, B)).astype(np.float32)     # ratios on both sides of the clip
    adv = rng.normal(0, 1, B).astype(np.float32)
    v = rng.normal(0, 1, B).astype(np.float32)
    v_old = rng.normal(0, 1, B).astype(np.float32)
    loss, pl, vl, dlogp, dv = oracle.ppo_loss(ref_lp, lp_old, adv, v, v_old, 0.2, 0.5, normalize_adv=B > 1)
    r_dl, r_dloc, r_dls, _ = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, dlogp)
    d = to_dev(dev, logits, x, loc, logstd, lp_old, adv, v, v_old)
    stats = kernels.adv_stats(d[5], normalize=B > 1)
    dbias = torch.full((A * N,), float("nan"), device=dev)
    out = kernels.ppo_mixture_fwd_bwd(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], stats, 0.2, 0.5, dbias=dbias)
    torch.cuda.synchronize()
    # fc_policy bias gradient = column sums of dlogits (a2c.py:541-554 BiasAdd gradient)
    r_db = r_dl.reshape(B, A * N).astype(np.float64).sum(0)
    np.testing.assert_allclose(dbias.cpu().numpy(), r_db, rtol=2e-4, atol=2e-5 * (np.abs(r_db).max() + 1e-30))
    out2 = kernels.ppo_mixture_fwd_bwd(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], stats, 0.2, 0.5)
    for k in ("logp", "dlogits", "dloc", "dlogstd", "dvalue", "loss3"):          # dbias is optional
        assert torch.equal(out[k], out2[k]), k
    np.testing.assert_allclose(out["logp"].cpu().numpy(), ref_lp, rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(out["dvalue"].cpu().numpy(), dv, rtol=1e-5, atol=1e-9)
    got = out["loss3"].cpu().numpy()
    # means of O(1) terms with mixed signs: absolute tolerance at fp32 resolution of the terms
    np.testing.assert_allclose(got, [loss, pl, vl], rtol=2e-5, atol=5e-6)
    sc = np.abs(r_dl).max() + 1e-30
    np.testing.assert_allclose(out["dlogits"].cpu().numpy().reshape(B, A, N), r_dl, rtol=2e-4, atol=2e-6 * sc)
    for k, ref in (("dloc", r_dloc), ("dlogstd", r_dls)):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref, rtol=1e-4, atol=5e-5 * (np.abs(ref).max() + 1e-30))


def test_bitwise_reproducible(oracle, dev):
    """Fixed-order slab reductions: two launches give identical bits."""
    from pfpn_amd import kernels
    B, A, N = 4096, 36, 35
    logits, x, loc, logstd = make_inputs(oracle, B, A, N, 9)
    g = np.random.default_rng(1).normal(0, 1, B).astype(np.float32)
    d = to_dev(dev, logits, x, loc, logstd, g)
    a = kernels.logprob_bwd(*d)
    b = kernels.logprob_bwd(*d)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("clip", [0.8, (-0.3, 1.5)])
def test_clip_advantage_in_every_loss_head(oracle, dev, clip):
    """clip_advantage (actor_critic.py:158-170): the normalised advantage is clipped inside the fused PPO kernel, the
    unfused PPO head and the A2C head alike (bounds ride in adv_stats[2:4]); value targets are not clipped."""
    from pfpn_amd import kernels
    B, A, N = 512, 36, 35
    logits, x, loc, logstd = make_inputs(oracle, B, A, N, 41, spread=1.05)
    rng = np.random.default_rng(41)
    ref_lp, _ = oracle.mixture_log_prob(logits, x, loc, logstd)
    lp_old = (ref_lp + rng.normal(0, 0.15, B)).astype(np.float32)
    adv = rng.normal(0, 1.5, B).astype(np.float32)
    v = rng.normal(0, 1, B).astype(np.float32)
    v_old = rng.normal(0, 1, B).astype(np.float32)
    loss, pl, vl, dlogp, dv = oracle.ppo_loss(ref_lp, lp_old, adv, v, v_old, 0.2, 0.5, clip_advantage=clip)
    loss_nc = oracle.ppo_loss(ref_lp, lp_old, adv, v, v_old, 0.2, 0.5)[0]
    assert abs(loss - loss_nc) > 1e-3                      # the bounds bite on this batch
    r_dl, _, _, _ = oracle.mixture_log_prob_bwd(logits, x, loc, logstd, dlogp)
    d = to_dev(dev, logits, x, loc, logstd, lp_old, adv, v, v_old)
    stats = kernels.new_adv_stats(dev, clip)
    kernels.adv_stats(d[5], True, out=stats)
    lo, hi = (min(clip), max(clip)) if hasattr(clip, "__len__") else (-clip, clip)
    assert stats[2].item() == np.float32(lo) and stats[3].item() == np.float32(hi)      # moments leave the bounds alone
    out = kernels.ppo_mixture_fwd_bwd(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], stats, 0.2, 0.5)
    np.testing.assert_allclose(out["loss3"].cpu().numpy(), [loss, pl, vl], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(out["dvalue"].cpu().numpy(), dv, rtol=1e-5, atol=1e-9)
    sc = np.abs(r_dl).max() + 1e-30
    np.testing.assert_allclose(out["dlogits"].cpu().numpy().reshape(B, A, N), r_dl, rtol=2e-4, atol=2e-6 * sc)
    g_lp, g_v, l3 = kernels.ppo_loss(to_dev(dev, ref_lp)[0], d[4], d[5], d[6], d[7], stats)
    np.testing.assert_allclose(l3.cpu().numpy(), [loss, pl, vl], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(g_lp.cpu().numpy(), dlogp, rtol=2e-5, atol=1e-9)
    # A2C head on host-computed targets
    vt = rng.normal(0, 1, B).astype(np.float32)
    a_ref = oracle.a2c_loss(ref_lp, adv, v, vt, 0.5, normalize_adv=True, clip_advantage=clip)
    dlp, dvv, l4 = kernels.a2c_loss(to_dev(dev, ref_lp)[0], d[5], d[6], to_dev(dev, vt)[0], stats, 0.5)
    np.testing.assert_allclose(l4.cpu().numpy()[:3], a_ref[:3], rtol=3e-5, atol=5e-6)
    np.testing.assert_allclose(dlp.cpu().numpy(), a_ref[3], rtol=2e-5, atol=1e-9)
