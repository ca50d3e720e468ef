import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
dev = torch.device("cuda:0")
shapes = {  # name: (M, K, N, transA, transB, batch)
 "L0 fwd x[4096,197]@W[197,1024]": (4096,197,1024,False,False,1),
 "L1 fwd bmm[2][4096,1024]@[1024,512]": (4096,1024,512,False,False,2),
 "head fwd [4096,512]@[512,1260]": (4096,512,1260,False,False,1),
 "head dW [512,4096]@[4096,1260]": (512,4096,1260,True,False,1),
 "head dX [4096,1260]@[1260,512]": (4096,1260,512,False,True,1),
 "L1 dW bmm[2][1024,4096]@[4096,512]": (1024,4096,512,True,False,2),
 "L1 dX bmm[2][4096,512]@[512,1024]": (4096,512,1024,False,True,2),
 "L0 dW [197,4096]@[4096,1024]": (197,4096,1024,True,False,1),
}
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
for name,(M,K,N,ta,tb,bt) in shapes.items():
    a = torch.randn((bt,K,M) if ta else (bt,M,K), device=dev)
    b = torch.randn((bt,N,K) if tb else (bt,K,N), device=dev)
    A = a.transpose(1,2) if ta else a
    Bm = b.transpose(1,2) if tb else b
    out = torch.empty(bt,M,N,device=dev)
    if bt==1:
        f = lambda: torch.mm(A[0],Bm[0],out=out[0])
    else:
        f = lambda: torch.bmm(A,Bm,out=out)
    us = bench(f)
    fl = 2.0*M*K*N*bt
    print("%-40s %8.1f us  %6.1f TF" % (name, us, fl/us/1e6))
