"""Does the forward of the stress shape (one 1-GiB read stream) care where its logits buffer starts?  One pool, logits at a
list of offsets; C from PFPN_WIDE_C as usual.  Usage (GPU box): python tools/skew_probe_fwd.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
B, A, N = 8192, 128, 256
n = B * A * N
pool = torch.empty(((1 << 30) + (600 << 20)) // 4, dtype=torch.float32, device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
pool[: n + (512 << 20) // 4].normal_(generator=g)
x = torch.rand(B, A, device=dev, generator=g) * 2.2 - 1.1
loc = torch.linspace(-1, 1, N, device=dev).repeat(A, 1).contiguous()
logstd = torch.full((A, N), float(np.log(2 / (N - 1))), device=dev)


def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print("pool base %#x (mod 1 GiB: %d MiB)" % (pool.data_ptr(), (pool.data_ptr() % (1 << 30)) >> 20))
for rnd in range(2):
    for off in [0, 4096, 64 << 10, 1 << 20, 2 << 20, 6 << 20, 16 << 20, 34 << 20, 64 << 20, 128 << 20, 256 << 20, 384 << 20, 512 << 20]:
        lg = pool[off // 4: off // 4 + n].view(B, A * N)
        print("round %d offset %9d B: fwd %.1f us" % (rnd, off, timeit(lambda: kernels.logprob_fwd(lg, x, loc, logstd))), flush=True)
