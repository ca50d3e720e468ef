"""The box's read+write rate for a 1 GiB float4 stream (the byte count of the stress shape's logits): the build's own copy
kernel over unroll depth x grid size x cache policy, torch's copy_ and hipMemcpy D2D beside it.  The best entry is what
bench.py's roofline_stress.device_copy uses (pfpn_stream_copy).  Usage: python tools/copy_probe.py"""
import os, sys
import torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from pfpn_amd import kernels as K
dev = torch.device("cuda", 0)
n = 8192 * 128 * 256
src = torch.randn(n, device=dev); dst = torch.empty_like(src)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


best = None
for nt in (0, 1):
    for u in (1, 2, 4, 8):
        for blocks in (1024, 2048, 4096, 8192, 16384, 32768):
            us = t(lambda: K.stream_copy(dst, src, u, nt, blocks))
            tb = 2 * n * 4 / us / 1e6
            if best is None or tb > best[0]:
                best = (tb, u, nt, blocks)
            print("unroll %d nt %d blocks %5d: %7.1f us  %.2f TB/s" % (u, nt, blocks, us, tb), flush=True)
us = t(lambda: dst.copy_(src))
print("torch copy_: %.1f us  %.2f TB/s" % (us, 2 * n * 4 / us / 1e6))
print("best own copy: %.2f TB/s = %.3f of 8 TB/s (unroll %d, nt %d, %d blocks)" % (best[0], best[0] / 8, best[1], best[2], best[3]))
assert torch.equal(dst, src)
