import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, A, N = 4096, 36, 35
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
names = {0:"start",1:"bank done",2:"barrier0"}
for it in range(4):
    for k,nm in enumerate(["top","fwd done","wait done","row done","bwd done","staged next"]):
        names[3+it*6+k] = "t%d %s" % (it, nm)
names[40]="loop end"; names[41]="slab done"; names[42]="past barrier"; names[43]="bank written"; names[44]="bank complete"
t0 = s[:9,0].min()
for w in (0, 4, 8):
    print("wave", w)
    prev = None
    for k in sorted(names):
        if s[w,k] == 0: continue
        dt = (s[w,k]-t0)/100.0   # s_memtime: 100 MHz ticks? print raw delta
        print("   %-14s %8d ticks  (+%d)" % (names[k], s[w,k]-t0, (s[w,k]-prev) if prev else 0))
        prev = s[w,k]

print("\nper-wave timeline (ticks since block start): top | fwd done | wait done | bwd done | staged next")
for it in range(4):
    print("tile", it)
    for w in range(9):
        base = 3 + it * 6
        print("   w%d  top %6d  fwd %6d  wait %6d  row %6d  bwd %6d  staged %6d" % (
            w, s[w, base] - t0, s[w, base + 1] - t0, s[w, base + 2] - t0, s[w, base + 3] - t0, s[w, base + 4] - t0,
            (s[w, base + 5] - t0) if s[w, base + 5] else -1))

print("loader waves: tile k issued / landed (ticks)")
for w in (9, 10):
    print("   w%d " % w + "  ".join("t%d issued %6d landed %6d" % (k, s[w, 3 + k * 6] - t0, s[w, 4 + k * 6] - t0) for k in range(4)))

print("tail: " + "  ".join("w%d loop end %d all arrived %d staged %d slabs out %d done %d" % (w, s[w,40]-t0, s[w,45]-t0, s[w,46]-t0, s[w,47]-t0, s[w,41]-t0) for w in (0,8,9)))
print("start-up: " + "  ".join("w%d barrier %d written %d complete %d regs %d" % (w, s[w,42]-t0, s[w,43]-t0, s[w,44]-t0, s[w,1]-t0) for w in (0,4,8)))
print("starts: " + "  ".join("w%d s0 %d s1 %d s2 %d" % (w, s[w,0]-t0, s[w,1]-t0, (s[w,2]-t0) if s[w,2] else -1) for w in (0,4,8,9,10)))
