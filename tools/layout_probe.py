import torch, sys
torch.backends.cuda.matmul.allow_tf32 = False
torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(100); torch.cuda.tunable.set_max_tuning_iterations(50)
torch.cuda.tunable.set_filename("/tmp/lp.csv")
dev = torch.device("cuda", 0)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H, N = 4096, 512, 1260
h = torch.randn(B, H, device=dev); dl = torch.randn(B, N, device=dev); b = torch.randn(N, device=dev)
W = torch.randn(H, N, device=dev); Wt = W.t().contiguous()
lo = torch.empty(B, N, device=dev); gW = torch.empty(H, N, device=dev); gWt = torch.empty(N, H, device=dev); dh = torch.empty(B, H, device=dev)
a = (t(lambda: torch.addmm(b, h, W, out=lo)), t(lambda: torch.mm(h.t(), dl, out=gW)), t(lambda: torch.mm(dl, W.t(), out=dh)))
c = (t(lambda: torch.addmm(b, h, Wt.t(), out=lo)), t(lambda: torch.mm(dl.t(), h, out=gWt)), t(lambda: torch.mm(dl, Wt, out=dh)))
print("W [512,1260]:  fwd %.1f dW %.1f dX %.1f  sum %.1f" % (a + (sum(a),)))
print("W^T [1260,512]: fwd %.1f dW %.1f dX %.1f  sum %.1f" % (c + (sum(c),)))
# layer 1 (batched) alternatives: weights [G,1024,512] vs transposed [G,512,1024]
G = 2
x = torch.randn(B, G * 1024, device=dev).view(B, G, 1024).transpose(0, 1)     # strided like act0
W1 = torch.randn(G, 1024, 512, device=dev); W1t = W1.transpose(1, 2).contiguous()
y = torch.empty(G, B, 512, device=dev); dy = torch.randn(G, B, 512, device=dev)
gw = torch.empty(G, 1024, 512, device=dev); gwt = torch.empty(G, 512, 1024, device=dev)
dxb = torch.empty(B, G * 1024, device=dev).view(B, G, 1024).transpose(0, 1)
a = (t(lambda: torch.bmm(x, W1, out=y)), t(lambda: torch.bmm(x.transpose(1, 2), dy, out=gw)), t(lambda: torch.bmm(dy, W1.transpose(1, 2), out=dxb)))
c = (t(lambda: torch.bmm(x, W1t.transpose(1, 2), out=y)), t(lambda: torch.bmm(dy.transpose(1, 2), x, out=gwt)), t(lambda: torch.bmm(dy, W1t, out=dxb)))
print("L1 W [G,1024,512]:  fwd %.1f dW %.1f dX %.1f  sum %.1f" % (a + (sum(a),)))
print("L1 W^T [G,512,1024]: fwd %.1f dW %.1f dX %.1f  sum %.1f" % (c + (sum(c),)))
# layer 0 merged: W0 [197, 2048] vs [2048, 197]
xs = torch.randn(B, 197, device=dev); W0 = torch.randn(197, 2048, device=dev); W0t = W0.t().contiguous()
y0 = torch.empty(B, 2048, device=dev); d0 = torch.randn(B, 2048, device=dev); g0 = torch.empty(197, 2048, device=dev); g0t = torch.empty(2048, 197, device=dev)
a = (t(lambda: torch.mm(xs, W0, out=y0)), t(lambda: torch.mm(xs.t(), d0, out=g0)))
c = (t(lambda: torch.mm(xs, W0t.t(), out=y0)), t(lambda: torch.mm(d0.t(), xs, out=g0t)))
print("L0 W [197,2048]: fwd %.1f dW %.1f sum %.1f | W^T [2048,197]: fwd %.1f dW %.1f sum %.1f" % (a + (sum(a),) + c + (sum(c),)))
