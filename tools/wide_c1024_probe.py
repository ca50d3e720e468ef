"""The wide mixture kernels on one GPU's shard of BASELINE.json config 5 (1024 x 128 x 256: 128 MiB of logits) under the
row-chunk count of the environment (PFPN_WIDE_C): forward and backward replayed from a hipGraph."""
import os, sys
import numpy as np, torch
R_ = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tools"))
from pfpn_amd import kernels
from sac_actor_probe import time_in_graph  # noqa
dev = torch.device("cuda", 0)
B, A, N = int(os.environ.get("PB", 1024)), 128, 256
g = torch.Generator(device=dev); g.manual_seed(0)
logits = torch.randn(B, A * N, device=dev, generator=g)
x = torch.rand(B, A, device=dev, generator=g) * 2 - 1
loc = torch.linspace(-1, 1, N, device=dev).repeat(A, 1).contiguous()
logstd = torch.full((A, N), float(np.log(2.0 / 255)), device=dev)
dlogp = torch.randn(B, device=dev, generator=g)
out = {"dlogits": torch.empty(B, A * N, device=dev), "dloc": torch.empty(A, N, device=dev), "dlogstd": torch.empty(A, N, device=dev)}
lp = torch.empty(B, device=dev)
db = torch.empty(A * N, device=dev)
tf = time_in_graph(lambda: kernels.logprob_fwd(logits, x, loc, logstd, out=lp), per_graph=10)
tb = time_in_graph(lambda: kernels.logprob_bwd(logits, x, loc, logstd, dlogp, out=out, dbias=db), per_graph=10)
print("PFPN_WIDE_C=%s rows %d: forward %.1f us, backward (+ dbias) %.1f us" % (os.environ.get("PFPN_WIDE_C", "default"), B, tf, tb), flush=True)
