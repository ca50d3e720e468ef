import os, sys, torch
dev = torch.device("cuda:0")
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
M=4096
for K in (197, 200, 208, 224, 256):
    x = torch.randn(M, K, device=dev); w = torch.randn(K, 1024, device=dev); out = torch.empty(M, 1024, device=dev)
    dy = torch.randn(M, 1024, device=dev); dw = torch.empty(K, 1024, device=dev)
    t1 = bench(lambda: torch.mm(x, w, out=out)); t2 = bench(lambda: torch.mm(x.t(), dy, out=dw))
    # strided view: K logical 197 inside padded buffer
    print("K=%d  fwd %.1f us (%.0f TF eff@197)  dW %.1f us (%.0f TF eff@197)" % (K, t1, 2*M*197*1024/t1/1e6, t2, 2*M*197*1024/t2/1e6))
# fused twin: N=2048
for K in (197, 208):
    x = torch.randn(M, K, device=dev); w = torch.randn(K, 2048, device=dev); out = torch.empty(M, 2048, device=dev)
    dy = torch.randn(M, 2048, device=dev); dw = torch.empty(K, 2048, device=dev)
    t1 = bench(lambda: torch.mm(x, w, out=out)); t2 = bench(lambda: torch.mm(x.t(), dy, out=dw))
    print("N=2048 K=%d  fwd %.1f us  dW %.1f us" % (K, t1, t2))
# head with padded N
for N in (1260, 1280):
    h = torch.randn(M, 512, device=dev); w = torch.randn(512, N, device=dev); out = torch.empty(M, N, device=dev)
    d = torch.randn(M, N, device=dev); dw = torch.empty(512, N, device=dev); dh = torch.empty(M, 512, device=dev)
    print("head N=%d fwd %.1f  dW %.1f  dX %.1f us" % (N, bench(lambda: torch.mm(h, w, out=out)), bench(lambda: torch.mm(h.t(), d, out=dw)), bench(lambda: torch.mm(d, w.t(), out=dh))))
