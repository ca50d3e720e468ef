"""Randomised shapes through the tile mixture kernel (forward, backward, fused PPO) against the NumPy oracle: ragged last
tiles, rings that wrap (more tiles per block than planes), every template instance, A from 1 to 44.  One-off confidence
run on the GPU box (the fixed shapes of tests/test_gpu_mixture.py are the regression suite)."""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import pfpn_oracle as orc
from pfpn_amd import kernels
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
n_cases = int(os.environ.get("FUZZ_CASES", "60"))
bad = 0
for case in range(n_cases):
    N = int(rng.choice([8, 16, 35, 50]))
    A = int(rng.integers(1, 45))
    if (A * N) % 4:
        A += 4 - (A % 4) if N % 4 else 0
        while (A * N) % 4: A += 1
    if A > 44: A = 44 if (44 * N) % 4 == 0 else 40
    B = int(rng.choice([1, 2, 3, 5, 17, 64, 255, 1023, 1025, 2049, 4097, 6003]))
    tanh_mode = bool(rng.integers(0, 2)) and N > 3
    loc, logstd = orc.init_particles(A, N, tanh_mode)
    loc = (loc + rng.normal(0, 0.05, loc.shape)).astype(np.float32)
    logstd = (logstd + rng.normal(0, 0.1, logstd.shape)).astype(np.float32)
    logits = rng.normal(0, 1, (B, A, N)).astype(np.float32)
    x = rng.uniform(-1.1, 1.1, (B, A)).astype(np.float32)
    g = rng.normal(0, 1, B).astype(np.float32)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (logits, x, loc, logstd, g)]
    ref_lp, ref_p = orc.mixture_log_prob(logits, x, loc, logstd, tanh_mode)
    r_dl, r_dloc, r_dls, r_dx = orc.mixture_log_prob_bwd(logits, x, loc, logstd, g, tanh_mode)
    try:
        lp, p = kernels.logprob_fwd(t[0], t[1], t[2], t[3], tanh_mode, want_pmix=True)
        o_dl, o_dloc, o_dls, o_dx = kernels.logprob_bwd(t[0], t[1], t[2], t[3], t[4], tanh_mode, want_dx=True)
        torch.cuda.synchronize()
        np.testing.assert_allclose(lp.cpu().numpy(), ref_lp, rtol=1e-5, atol=1e-5)
        sc = np.abs(r_dl).max() + 1e-30
        np.testing.assert_allclose(o_dl.cpu().numpy(), r_dl, rtol=1e-4, atol=3e-6 * sc)
        np.testing.assert_allclose(o_dx.cpu().numpy(), r_dx, rtol=1e-4, atol=1e-5 * (np.abs(r_dx).max() + 1e-30))
        for got, ref in ((o_dloc, r_dloc), (o_dls, r_dls)):
            np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * (np.abs(ref).max() + 1e-30))
        if not tanh_mode and B > 1:
            lp_old = (ref_lp + rng.normal(0, 0.15, B)).astype(np.float32)
            adv = rng.normal(0, 1, B).astype(np.float32); v = rng.normal(0, 1, B).astype(np.float32); vo = rng.normal(0, 1, B).astype(np.float32)
            loss, pl, vl, dlogp, dv = orc.ppo_loss(ref_lp, lp_old, adv, v, vo, 0.2, 0.5, normalize_adv=True)
            q_dl, q_dloc, q_dls, _ = orc.mixture_log_prob_bwd(logits, x, loc, logstd, dlogp)
            d = [torch.from_numpy(a).to(dev) for a in (lp_old, adv, v, vo)]
            st = kernels.adv_stats(d[1], normalize=True)
            dbias = torch.full((A * N,), float("nan"), device=dev)
            out = kernels.ppo_mixture_fwd_bwd(t[0], t[1], t[2], t[3], d[0], d[1], d[2], d[3], st, 0.2, 0.5, dbias=dbias)
            torch.cuda.synchronize()
            sc = np.abs(q_dl).max() + 1e-30
            # rows whose ratio sits on the clip boundary may take the other branch: compare where the oracle's gradient
            # agrees in sign of activity
            got = out["dlogits"].cpu().numpy().reshape(B, A, N)
            rowbad = np.abs(got - q_dl).reshape(B, -1).max(1) > (1e-4 * np.abs(q_dl).reshape(B, -1).max(1) + 3e-6 * sc)
            assert rowbad.sum() <= max(1, B // 500), "ppo dlogits: %d rows off" % rowbad.sum()
            np.testing.assert_allclose(out["logp"].cpu().numpy(), ref_lp, rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(out["dvalue"].cpu().numpy(), dv, rtol=1e-5, atol=1e-9)
            if rowbad.sum() == 0:
                r_db = q_dl.reshape(B, A * N).astype(np.float64).sum(0)
                np.testing.assert_allclose(dbias.cpu().numpy(), r_db, rtol=2e-4, atol=2e-5 * (np.abs(r_db).max() + 1e-30))
        print("ok   B=%5d A=%2d N=%2d tanh=%d" % (B, A, N, tanh_mode))
    except AssertionError as e:
        bad += 1
        print("FAIL B=%5d A=%2d N=%2d tanh=%d: %s" % (B, A, N, tanh_mode, str(e).strip().splitlines()[0][:120]))
print("%d / %d cases failed" % (bad, n_cases))
sys.exit(1 if bad else 0)
