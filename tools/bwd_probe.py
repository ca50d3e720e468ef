"""the wide backward alone on the stress shape (8192 x 128 x 256), 3 x 10 launches: us per launch"""
import os, sys, json, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
B, A, N = 8192, 128, 256
g = torch.Generator(device=dev); g.manual_seed(0)
logits, dlogits = kernels.stream_pair(B, A * N, dev)
logits.normal_(generator=g)
out = {"dlogits": dlogits, "dloc": torch.empty(A, N, device=dev), "dlogstd": torch.empty(A, N, device=dev)}
x = torch.rand(B, A, device=dev, generator=g) * 2.2 - 1.1
loc = torch.linspace(-1, 1, N, device=dev).repeat(A, 1).contiguous()
logstd = torch.full((A, N), float(np.log(2 / (N - 1))), device=dev)
dl = torch.randn(B, device=dev, generator=g)
f = lambda: kernels.logprob_bwd(logits, x, loc, logstd, dl, out=out)
f(); f(); torch.cuda.synchronize()
res = []
for r in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    res.append(round(e0.elapsed_time(e1) * 100, 1))
nb = 2 * 4 * B * A * N + 4 * B * A + 8 * B + 16 * A * N
print(json.dumps({"bwd_us": res, "frac_best": round(nb / min(res) / 1e3 / 8000, 4), "checksum": float(out["dloc"].double().sum())}))
