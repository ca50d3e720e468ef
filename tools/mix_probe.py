import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, A, N = int(os.environ.get("PB", 4096)), int(os.environ.get("PA", 36)), 35
g = torch.Generator(device=dev); g.manual_seed(0)
logits = torch.randn(B, A*N, device=dev, generator=g)
x = (torch.rand(B, A, device=dev, generator=g)*2.2-1.1)
loc = torch.linspace(-1,1,N,device=dev).repeat(A,1).contiguous()
logstd = torch.full((A,N), float(np.log(2/34)), device=dev)
lp_old = kernels.logprob_fwd(logits, x, loc, logstd)
adv = torch.randn(B, device=dev, generator=g); v = torch.randn(B, device=dev, generator=g); vo = torch.randn(B, device=dev, generator=g)
st = kernels.adv_stats(adv)
out = kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, v, vo, st)
dl = torch.randn(B, device=dev)
def timeit(f, n=20, reps=10):
    """device time per call: n calls captured in one hipGraph (no host launch overhead), replayed."""
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        f()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/(n*reps)
t_ppo = timeit(lambda: kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, v, vo, st, out=out))
t_fwd = timeit(lambda: kernels.logprob_fwd(logits, x, loc, logstd))
t_bwd = timeit(lambda: kernels.logprob_bwd(logits, x, loc, logstd, dl))
print("ROWS=%s PER_CU=%s  ppo %.1f us   fwd %.1f us   bwd %.1f us" % (os.environ.get("PFPN_MIX_ROWS"), os.environ.get("PFPN_MIX_PER_CU"), t_ppo, t_fwd, t_bwd))
