"""Does the relative placement of the two 1-GiB streams of the stress shape's backward (logits read, dlogits written)
matter?  One pool, logits at its start, dlogits at 1 GiB + skew for a list of skews; the backward kernel and the build's
own streaming copy timed for each, two rounds interleaved.  Usage (GPU box): python tools/skew_probe.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
B, A, N = 8192, 128, 256
n = B * A * N
GiB = 1 << 30
pool = torch.empty((2 * GiB + (256 << 20)) // 4, dtype=torch.float32, device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
logits = pool[:n].view(B, A * N)
logits.normal_(generator=g)
x = torch.rand(B, A, device=dev, generator=g) * 2.2 - 1.1
loc = torch.linspace(-1, 1, N, device=dev).repeat(A, 1).contiguous()
logstd = torch.full((A, N), float(np.log(2 / (N - 1))), device=dev)
dl = torch.randn(B, device=dev, generator=g)
out_small = {"dloc": torch.empty(A, N, device=dev), "dlogstd": torch.empty(A, N, device=dev)}


def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


skews = [0] + [1 << k for k in range(8, 28)] + [4096 + 512, 4096 + 2048, 3 * 4096, 5 * 4096, 4096 + (1 << 20), 4096 + (16 << 20),
                                                 (16 << 10) + 4096, (16 << 20) + 4096]
if len(sys.argv) > 1:
    skews = [int(a, 0) for a in sys.argv[1:]]
print("pool base %#x" % pool.data_ptr())
for rnd in range(1):
    for sk in skews:
        d = pool[(GiB + sk) // 4:(GiB + sk) // 4 + n]
        o = dict(out_small, dlogits=d.view(B, A * N))
        tb = timeit(lambda: kernels.logprob_bwd(logits, x, loc, logstd, dl, out=o))
        tc = timeit(lambda: kernels.stream_copy(d, pool[:n], 1, True, 32768))
        print("round %d skew %9d B: bwd %.1f us   own copy %.1f us" % (rnd, sk, tb, tc), flush=True)
