"""The one-launch PPO pass on a grid TWICE the device's capacity with chunks that take every C-th block (test knobs): half of
every chunk's partners cannot be resident -- the bounded waits must expire (flag set, gradients NaN), the launch must END,
and the next normal launch must be clean."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda", 0)
B, A, N = 2048, 128, 256
g = torch.Generator(device=dev).manual_seed(1)
logits = torch.randn(B, A * N, device=dev, generator=g)
x = torch.rand(B, A, device=dev, generator=g) * 2 - 1
loc = torch.linspace(-1, 1, N, device=dev).repeat(A, 1).contiguous()
logstd = torch.full((A, N), float(np.log(2.0 / (N - 1))), device=dev)
lp0 = kernels.logprob_fwd(logits.view(B, A, N), x, loc, logstd)
adv = torch.randn(B, device=dev, generator=g); v = torch.zeros(B, device=dev)
stats = kernels.adv_stats(adv)
out, chunks, err = kernels.ppo_wide_fused(logits, x, loc, logstd, lp0, adv, v, v, stats)
torch.cuda.synchronize()
print("normal: chunks", chunks, "flag", int(err.item()), "finite dloc", bool(torch.isfinite(out["dloc"]).all()))
os.environ["PFPN_WIDE_F2L_OVERSUBSCRIBE"] = "1"
os.environ["PFPN_WIDE_F2L_SAFE"] = "3"
t0 = time.time()
out, chunks, err = kernels.ppo_wide_fused(logits, x, loc, logstd, lp0, adv, v, v, stats)
torch.cuda.synchronize()
print("oversubscribed: chunks", chunks, "flag", int(err.item()), "finite dloc", bool(torch.isfinite(out["dloc"]).all()),
      "nan entries", int(torch.isnan(out["dloc"]).sum().item()), "seconds %.2f" % (time.time() - t0))
os.environ.pop("PFPN_WIDE_F2L_OVERSUBSCRIBE"); os.environ.pop("PFPN_WIDE_F2L_SAFE")
out, chunks, err = kernels.ppo_wide_fused(logits, x, loc, logstd, lp0, adv, v, v, stats)
torch.cuda.synchronize()
print("normal again: chunks", chunks, "flag", int(err.item()), "finite dloc", bool(torch.isfinite(out["dloc"]).all()))
