"""Probe: first-layer GEMMs merged across the two trunks + K padded, strided-batch bmm operands."""
import os, sys, torch
torch.backends.cuda.matmul.allow_tf32 = False
if "--tune" in sys.argv:
    os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
    torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(30)
dev = torch.device("cuda", 0)
B = 4096

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for K in (197, 200, 208, 224, 256):
    x = torch.randn(B, K, device=dev); w2 = torch.randn(2, K, 1024, device=dev); w1 = torch.randn(K, 2048, device=dev)
    o2 = torch.empty(2, B, 1024, device=dev); o1 = torch.empty(B, 2048, device=dev)
    d2 = torch.randn(2, B, 1024, device=dev); d1 = torch.randn(B, 2048, device=dev)
    g2 = torch.empty(2, K, 1024, device=dev); g1 = torch.empty(K, 2048, device=dev)
    def f2():
        torch.mm(x, w2[0], out=o2[0]); torch.mm(x, w2[1], out=o2[1])
    def f1(): torch.mm(x, w1, out=o1)
    xt = x.t()
    def b2():
        torch.mm(xt, d2[0], out=g2[0]); torch.mm(xt, d2[1], out=g2[1])
    def b1(): torch.mm(xt, d1, out=g1)
    print("K=%d fwd 2xmm %.1f us  merged %.1f us | dW 2xmm %.1f  merged %.1f" % (K, t(f2), t(f1), t(b2), t(b1)), flush=True)

# strided-batch operands for layer 2
a_c = torch.randn(2, B, 1024, device=dev)
a_s = torch.randn(B, 2, 1024, device=dev).transpose(0, 1)          # [2,B,1024] strides (1024, 2048, 1)
w = torch.randn(2, 1024, 512, device=dev); o = torch.empty(2, B, 512, device=dev)
print("fc2 fwd bmm contiguous %.1f us  strided-A %.1f us" % (t(lambda: torch.bmm(a_c, w, out=o)), t(lambda: torch.bmm(a_s, w, out=o))))
d = torch.randn(2, B, 512, device=dev)
da_c = torch.empty(2, B, 1024, device=dev); da_s = torch.empty(B, 2, 1024, device=dev).transpose(0, 1)
wt = w.transpose(1, 2)
print("fc2 dX bmm contiguous out %.1f us  strided out %.1f us" % (t(lambda: torch.bmm(d, wt, out=da_c)), t(lambda: torch.bmm(d, wt, out=da_s))))
ref = torch.bmm(d, wt); torch.bmm(d, wt, out=da_s); print("strided out correct:", torch.equal(ref, da_s), da_s.stride())
gw = torch.empty(2, 1024, 512, device=dev)
print("fc2 dW bmm contiguous %.1f us  strided-A %.1f us" % (t(lambda: torch.bmm(a_c.transpose(1, 2), d, out=gw)), t(lambda: torch.bmm(a_s.transpose(1, 2), d, out=gw))))
# policy head with bias epilogue
h = torch.randn(B, 512, device=dev); wp = torch.randn(512, 1260, device=dev); bp = torch.randn(1260, device=dev); lo = torch.empty(B, 1260, device=dev)
print("policy mm %.1f us  addmm(bias) %.1f us" % (t(lambda: torch.mm(h, wp, out=lo)), t(lambda: torch.addmm(bp, h, wp, out=lo))))
for Np in (1260, 1280):
    wp = torch.randn(512, Np, device=dev); lo = torch.empty(B, Np, device=dev); dl = torch.randn(B, Np, device=dev)
    gwp = torch.empty(512, Np, device=dev); dh = torch.empty(B, 512, device=dev)
    print("N=%d policy fwd %.1f  dW %.1f  dh %.1f" % (Np, t(lambda: torch.mm(h, wp, out=lo)), t(lambda: torch.mm(h.t(), dl, out=gwp)), t(lambda: torch.mm(dl, wp.t(), out=dh))))
