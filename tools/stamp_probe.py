"""In-kernel phase stamps (s_memtime) of block 0 of k_mix_tile<35,PPO> at 4096 x 36 x 35.
Needs the diagnostic build: rm -rf pfpn_amd/build && PFPN_BUILD_STAMPS=1 python -m pfpn_amd.build"""
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
buf = torch.zeros(16*64, dtype=torch.int64, device=dev)
for _ in range(3): kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, v, vo, st, out=out)
torch.cuda.synchronize()
assert lib.pfpn_debug_stamps(buf.data_ptr()) == 0
kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, v, vo, st, out=out)
torch.cuda.synchronize()
lib.pfpn_debug_stamps(None)
s = buf.cpu().numpy().reshape(16, 64)
NC = (A + 3) // 4
t0 = s[:14, 0][s[:14, 0] > 0].min()
def d(w, k): return int(s[w, k] - t0) if s[w, k] else -1
print("cycles since block start (s_memtime)")
print("compute waves: start | bank written | past barrier | bank in regs | fwd(0) done | loop end | slabs out")
for w in range(NC):
    print("   w%-2d %6d %6d %6d %6d %6d %6d %6d" % (w, d(w, 0), d(w, 43), d(w, 42), d(w, 1), d(w, 2), d(w, 40), d(w, 41)))
for it in range(4):
    print("tile %d: top | fwd(k+1) done | row gradient seen | bwd done" % it)
    for w in range(NC):
        b = 3 + it * 6
        print("   w%-2d %6d %6d %6d %6d" % (w, d(w, b), d(w, b + 1), d(w, b + 3), d(w, b + 4)))
print("loader waves: start | past barrier | tiles 1.. landed")
for w in (NC, NC + 1):
    print("   w%-2d %6d %6d   " % (w, d(w, 0), d(w, 42)) + " ".join("%6d" % d(w, 4 + k * 6) for k in range(1, 4)))
w = NC + 2
print("row wave: start | row result of tile k published")
print("   w%-2d %6d   " % (w, d(w, 0)) + " ".join("%6d" % d(w, 5 + k * 6) for k in range(4)))
print("storer waves: start | tile k stored (the block's last tile is stored by the compute waves)")
for w in (NC + 3, NC + 4):
    print("   w%-2d %6d   " % (w, d(w, 0)) + " ".join("%6d" % d(w, 8 + k * 6) for k in range(4)))
