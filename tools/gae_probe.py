import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
def timeit(f, n=20, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)
for E, T in ((8, 512), (64, 64), (1, 4096), (1, 512), (512, 8)):
    r = torch.randn(E * T, device=dev); v = torch.randn(E * T + E, device=dev)
    off = torch.arange(0, E * T + 1, T, device=dev, dtype=torch.int64)
    adv = kernels.gae(r, v, off, 0.95, 0.95 * 0.95)
    t = timeit(lambda: kernels.gae(r, v, off, 0.95, 0.95 * 0.95))
    print("k_gae  %3d sequences x %4d steps (%5d transitions): %6.2f us" % (E, T, E * T, t))
for E, T in ((16, 64), (64, 64), (8, 512)):
    n = E * T
    lt, lb, v, r = (torch.randn(n, device=dev) * 0.1 for _ in range(4))
    nt = torch.ones(n, device=dev)
    kernels.vtrace(lt, lb, v, r, nt, E, 0.95)
    t = timeit(lambda: kernels.vtrace(lt, lb, v, r, nt, E, 0.95))
    print("k_vtrace %3d sequences x %4d steps (%5d transitions): %6.2f us" % (E, T, n, t))
