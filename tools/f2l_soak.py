"""Soak of the one-launch PPO pass's exchange protocol: N launches on the 8192 x 128 x 256 shape (and a ragged small one), every
launch's logp / dloc / dlogstd / loss3 / a strided sample of dlogits compared BITWISE on the device with the first launch's, the
error flag read at the end.  A race in the tagged exchange (a stale or torn partial accepted) would show as a mismatch.
  python tools/f2l_soak.py [launches]      (PFPN_WIDE_F2L_SAFE=1 / 2: the memory-side exchange)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
for B, A, N in ((8192, 128, 256), (1031, 12, 256)):
    g = torch.Generator(device=dev).manual_seed(B)
    logits, dlogits = kernels.stream_pair(B, A * N, dev)
    logits.normal_(generator=g)
    x = torch.rand(B, A, device=dev, generator=g) * 2 - 1
    loc = (torch.linspace(-1, 1, N, device=dev)[None, :].expand(A, N) + 0.003 * torch.randn(A, N, device=dev, generator=g)).contiguous()
    logstd = torch.full((A, N), float(np.log(2.0 / (N - 1))), device=dev)
    lp_old = (kernels.logprob_fwd(logits, x, loc, logstd) + 0.15 * torch.randn(B, device=dev, generator=g)).contiguous()
    adv, v, vo = (torch.randn(B, device=dev, generator=g) for _ in range(3))
    stats = kernels.adv_stats(adv)
    out = {"logp": torch.empty(B, device=dev), "dlogits": dlogits, "dloc": torch.empty(A, N, device=dev),
           "dlogstd": torch.empty(A, N, device=dev), "dvalue": torch.empty(B, device=dev), "loss3": torch.empty(3, device=dev),
           "dlogp": torch.empty(B, device=dev)}
    rows = torch.arange(0, B, max(1, B // 64), device=dev)
    _, chunks, err = kernels.ppo_wide_fused(logits, x, loc, logstd, lp_old, adv, v, vo, stats, out=out)
    ref = {k: out[k].clone() for k in ("logp", "dloc", "dlogstd", "loss3", "dlogp")}
    ref_rows = out["dlogits"][rows].clone()
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n):
        kernels.ppo_wide_fused(logits, x, loc, logstd, lp_old, adv, v, vo, stats, out=out)
        for k in ref:
            bad += (out[k] != ref[k]).sum()
        bad += (out["dlogits"][rows] != ref_rows).sum()
    torch.cuda.synchronize()
    print("%d x %d x %d: %d launches in %.1f s, %d chunks, mismatching values %d, wait-expired flag %d, mode %s"
          % (B, A, N, n, time.time() - t0, chunks, int(bad.item()), int(err.item()), os.environ.get("PFPN_WIDE_F2L_SAFE", "0")), flush=True)
