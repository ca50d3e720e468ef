"""HBM-roofline probe of the mixture kernels on the stress shape (BASELINE.json config 5: 128 action dims x
256 particles, batch 8192 -> 1 GiB of logits)."""
import os, sys, json, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
B, A, N = int(os.environ.get("SB", 8192)), int(os.environ.get("SA", 128)), int(os.environ.get("SN", 256))
g = torch.Generator(device=dev); g.manual_seed(0)
logits, dlogits = kernels.stream_pair(B, A * N, dev)          # placed as the network places them
logits.normal_(generator=g)
bw_out = {"dlogits": dlogits, "dloc": torch.empty(A, N, device=dev), "dlogstd": torch.empty(A, N, device=dev)}
x = torch.rand(B, A, device=dev, generator=g) * 2.2 - 1.1
loc = torch.linspace(-1, 1, N, device=dev).repeat(A, 1).contiguous()
logstd = torch.full((A, N), float(np.log(2 / (N - 1))), device=dev)
dl = torch.randn(B, device=dev, generator=g)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
L, X, V, P = 4 * B * A * N, 4 * B * A, 4 * B, 8 * A * N
t_f = timeit(lambda: kernels.logprob_fwd(logits, x, loc, logstd))
t_b = timeit(lambda: kernels.logprob_bwd(logits, x, loc, logstd, dl, out=bw_out))
# the PPO composition at this shape: forward -> loss head -> backward into the caller's buffers (3L)
lp_old = kernels.logprob_fwd(logits, x, loc, logstd).clone()
stats = kernels.adv_stats(dl)
zv = torch.zeros(B, device=dev)
p_out = dict(bw_out, logp=torch.empty(B, device=dev), dvalue=torch.empty(B, device=dev), loss3=torch.empty(3, device=dev))
t_p = timeit(lambda: kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, dl, zv, zv, stats, out=p_out))
if os.environ.get("STRESS_COPY", "1") != "0":
    cp = torch.empty_like(logits)
    t_c = timeit(lambda: cp.copy_(logits))
else:
    t_c = float("nan")
out = {"shape": [B, A, N],
       "fwd": {"us": t_f, "bytes": L + X + V + P, "GBps": (L + X + V + P) / t_f / 1e3, "frac_of_8TBps": (L + X + V + P) / t_f / 1e3 / 8000},
       "bwd": {"us": t_b, "bytes": 2 * L + X + 2 * V + 2 * P, "GBps": (2 * L + X + 2 * V + 2 * P) / t_b / 1e3, "frac_of_8TBps": (2 * L + X + 2 * V + 2 * P) / t_b / 1e3 / 8000},
       "ppo": {"us": t_p, "bytes": 3 * L + 2 * X + 8 * V + 3 * P, "frac_of_8TBps": (3 * L + 2 * X + 8 * V + 3 * P) / t_p / 1e3 / 8000},
       "torch_copy": {"us": t_c, "GBps": 2 * L / t_c / 1e3}}
print(json.dumps(out))
