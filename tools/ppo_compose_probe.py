"""Where does the composed PPO pass (forward -> loss head -> backward) of the 1 GiB stress shape spend its time?"""
import os, sys, json, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
B, A, N = 8192, 128, 256
g = torch.Generator(device=dev); g.manual_seed(0)
logits, dlogits = kernels.stream_pair(B, A * N, dev)
logits.normal_(generator=g)
x = torch.rand(B, A, device=dev, generator=g) * 2.2 - 1.1
loc = torch.linspace(-1, 1, N, device=dev).repeat(A, 1).contiguous()
logstd = torch.full((A, N), float(np.log(2 / (N - 1))), device=dev)
adv = torch.randn(B, device=dev, generator=g)
out = {"dlogits": dlogits, "dloc": torch.empty(A, N, device=dev), "dlogstd": torch.empty(A, N, device=dev),
       "logp": torch.empty(B, device=dev), "dvalue": torch.empty(B, device=dev), "loss3": torch.empty(3, device=dev)}
lp_old = kernels.logprob_fwd(logits, x, loc, logstd).clone()
stats = kernels.adv_stats(adv); zv = torch.zeros(B, device=dev)
dlogp = torch.randn(B, device=dev, generator=g)
def fwd(): kernels.logprob_fwd(logits, x, loc, logstd, out=out["logp"])
def loss(): return kernels.ppo_loss(out["logp"], lp_old, adv, zv, zv, stats, dvalue=out["dvalue"], loss3=out["loss3"])[0]
def bwd(dl=dlogp): kernels.logprob_bwd(logits, x, loc, logstd, dl, out=out)
def timeit(f, n=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / n, 1)
res = {}
for r in range(2):
    for name, f in (("fwd", fwd), ("bwd", bwd), ("loss", loss), ("fwd+bwd", lambda: (fwd(), bwd())),
                    ("fwd+loss+bwd", lambda: (fwd(), bwd(loss()))),
                    ("compose", lambda: kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, zv, zv, stats, out=out))):
        res.setdefault(name, []).append(timeit(f))
print(json.dumps(res))
