import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0"); B, H = 4096, 512
y = torch.randn(2 * B, H, device=dev); b = torch.randn(2, H, device=dev); w = torch.randn(H, device=dev); vb = torch.randn(1, device=dev); v = torch.empty(B, device=dev)
def t(f, n=20, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream()); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)
print("bias_act_value %.2f us | bias_act %.2f us" % (t(lambda: kernels.bias_act_value(y, b, B, 1, w, vb, v)), t(lambda: kernels.bias_act(y, b, True, B))))
