import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
def bench(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
R = int(os.environ.get("ROWS", 4096))
shapes = (("L0", 2, 200, 1024), ("L1", 2, 1024, 512), ("head", 1, 512, 1260))
if os.environ.get("ONLY_L1"): shapes = shapes[1:2]
for name, G, K, N in shapes:
    x = torch.randn(R, K, device=dev) if name == "L0" else torch.randn(G, R, K, device=dev)
    w = torch.randn(G, K, N, device=dev); b = torch.randn(G, N, device=dev)
    y = torch.empty(G, R, N, device=dev); dy = torch.randn(G, R, N, device=dev)
    hin = torch.rand(G, R, K, device=dev) * 6; dx = torch.empty(G, R, K, device=dev)
    dw = torch.empty(G, K, N, device=dev); db = torch.empty(G, N, device=dev)
    fl = 2.0 * G * R * K * N
    t1 = bench(lambda: kernels.dense_fwd(x, w, b, y, True))
    t2 = bench(lambda: kernels.dense_dgrad(dy, w, hin, dx))
    t3 = bench(lambda: kernels.dense_wgrad(x, dy, dw, db))
    xe = x if x.dim() == 3 else x.unsqueeze(0).expand(G, R, K).contiguous()
    r1 = bench(lambda: torch.bmm(xe, w, out=y)); r2 = bench(lambda: torch.bmm(dy, w.transpose(1, 2), out=dx)); r3 = bench(lambda: torch.bmm(xe.transpose(1, 2), dy, out=dw))
    print("%-5s fwd %.1f us %.0f TF (torch %.1f) | dgrad %.1f us %.0f TF (torch %.1f) | wgrad %.1f us %.0f TF (torch %.1f)" % (name, t1, fl/t1/1e6, r1, t2, fl/t2/1e6, r2, t3, fl/t3/1e6, r3))
