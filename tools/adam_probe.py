"""clip + Adam launch alone on the DPPO parameter buffer (prenorm form: partial sums given)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0"); n = 2104520
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4)); v = v.abs()
off = (ctypes.c_longlong * 1)(0); ln = (ctypes.c_longlong * 1)(n); lr = (ctypes.c_float * 1)(1e-4)
bp = torch.tensor([[0.9, 0.999, 0.9, 0.999]], device=dev); ns = torch.zeros(2, device=dev); part = torch.rand(330, device=dev)
def f(): kernels.clip_adam(p, g, m, v, off, ln, lr, bp, 1.0, ns, prenorm=(part, 330))
for _ in range(3): f()
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream()); gr = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    f(); torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(20): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): gr.replay()
e1.record(); torch.cuda.synchronize()
print("clip+Adam (prenorm) %.2f us per launch" % (e0.elapsed_time(e1) * 1e3 / 200))
