"""Which job of the learner step's deferred reduction launch sets its duration?  Times pfpn_reduce_jobs on the step's job
shapes, all together and with one job kind left out at a time."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
B, AN, S = 4096, 1260, 197
f = lambda *s: torch.randn(*s, device=dev)
ws_mix = f(256 * (3 * AN + 4)).view(torch.uint8); ws0 = f(256 * 2048).view(torch.uint8); ws1 = f(2 * 256 * 512).view(torch.uint8)
wsv = f(256 * 512).view(torch.uint8); dv = f(B); wsm = torch.randn(256 * 2 * S, device=dev, dtype=torch.float64).view(torch.uint8)
o = lambda n: torch.empty(n, device=dev)
step = torch.tensor([5.0], device=dev); mean, std = f(S), f(S).abs() + 1
jobs = {
    "mix (3 x 1260 cols of 256 slabs, stride 3784)": [kernels.job_colsum(ws_mix, k * AN, 256, 3 * AN + 4, AN, o(AN)) for k in range(3)],
    "loss": [kernels.job_ppo_loss(ws_mix, 3 * AN, 256, 3 * AN + 4, 1 / B, 0.5, o(3))],
    "L0 bias (2048 cols of 256 chunks)": [kernels.job_colsum(ws0, 0, 256, 2048, 2048, o(2048))],
    "L1 bias (2 x 512 cols)": [kernels.job_colsum(ws1, g * 256 * 512, 256, 512, 512, o(512)) for g in range(2)],
    "value head dw + db": [kernels.job_colsum(wsv, 0, 256, 512, 512, o(512)), kernels.job_vecsum(dv, o(1))],
    "norm EMA (fp64)": [kernels.job_norm_ema(wsm, B, S, step, mean, std, o(S), o(S))],
}
def t(js, n=200):
    for _ in range(5): kernels.reduce_jobs(js)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        kernels.reduce_jobs(js); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): kernels.reduce_jobs(js)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 200
allj = [j for v in jobs.values() for j in v]
print("all %d jobs: %.2f us" % (len(allj), t(allj)))
for k in jobs:
    rest = [j for kk, v in jobs.items() if kk != k for j in v]
    print("  without %-50s %.2f us   | alone %.2f us" % (k, t(rest), t(jobs[k])))
