"""Stress shape (8192 x 128 x 256): the wide backward writing dlogits (a) into the paired buffer kernels.stream_pair places,
(b) into an independent allocation, (c) IN PLACE over the logits it has just read (a wave reads a segment whole before it
writes it back, so aliasing is safe).  Alternating, 3 rounds.  PFPN_WIDE_C=<C> per process for the chunk count."""
import os, sys, json, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
B, A, N = 8192, 128, 256
g = torch.Generator(device=dev); g.manual_seed(0)
logits, paired = kernels.stream_pair(B, A * N, dev)
logits.normal_(generator=g)
keep = logits.clone()
indep = torch.empty_like(logits)
x = torch.rand(B, A, device=dev, generator=g) * 2.2 - 1.1
loc = torch.linspace(-1, 1, N, device=dev).repeat(A, 1).contiguous()
logstd = torch.full((A, N), float(np.log(2 / (N - 1))), device=dev)
dl = torch.randn(B, device=dev, generator=g)
dloc, dls = torch.empty(A, N, device=dev), torch.empty(A, N, device=dev)
def timeit(dst, n=10):
    f = lambda: kernels.logprob_bwd(logits, x, loc, logstd, dl, out={"dlogits": dst, "dloc": dloc, "dlogstd": dls})
    if dst is logits:
        logits.copy_(keep)
    f()
    if dst is logits:
        logits.copy_(keep)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()          # (in place: later repeats read gradients as logits -- same traffic, finite values)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
# correctness of the in-place form: same bits as the out-of-place result
ref = kernels.logprob_bwd(logits, x, loc, logstd, dl, out={"dlogits": paired, "dloc": dloc.clone(), "dlogstd": dls.clone()})[0].clone()
kernels.logprob_bwd(logits, x, loc, logstd, dl, out={"dlogits": logits, "dloc": dloc, "dlogstd": dls})
same = bool(torch.equal(ref.view(-1), logits.view(-1)))
logits.copy_(keep)
nb = 2 * 4 * B * A * N + 4 * B * A + 8 * B + 16 * A * N
res = {"C": os.environ.get("PFPN_WIDE_C", "default"), "in_place_bits_equal": same}
for r in range(3):
    for name, dst in (("paired", paired), ("independent", indep), ("in_place", logits)):
        u = timeit(dst)
        res.setdefault(name, []).append(round(u, 1))
res["frac_best"] = {k: round(nb / min(v) / 1e3 / 8000, 4) for k, v in res.items() if isinstance(v, list)}
print(json.dumps(res))
