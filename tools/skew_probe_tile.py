"""Same question as skew_probe.py for the headline's fused mixture kernel (4096 x 36 x 35: 20.6 MB in, 20.6 MB out): does the
distance between logits and dlogits move k_mix_tile<35,PPO>?  Replayed from a hipGraph.  Usage: python tools/skew_probe_tile.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
B, A, N = 4096, 36, 35
n = B * A * N
pool = torch.empty((256 << 20) // 4, dtype=torch.float32, device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
logits = pool[:n].view(B, A * N); logits.normal_(generator=g)
x = torch.rand(B, A, device=dev, generator=g) * 2.2 - 1.1
loc = torch.linspace(-1, 1, N, device=dev).repeat(A, 1).contiguous()
logstd = torch.full((A, N), float(np.log(2 / (N - 1))), device=dev)
adv = torch.randn(B, device=dev, generator=g)
lp_old = kernels.logprob_fwd(logits, x, loc, logstd).clone()
v = torch.zeros(B, device=dev)
stats = kernels.adv_stats(adv)
base = (n * 4 + 4095) // 4096 * 4096
for sk in [0, 256, 1024, 4096, 8192, 16 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20, 16 << 20, 64 << 20]:
    off = (base + sk) // 4
    out = {"logp": torch.empty(B, device=dev), "dlogits": pool[off:off + n].view(B, A * N), "dloc": torch.empty(A, N, device=dev),
           "dlogstd": torch.empty(A, N, device=dev), "dvalue": torch.empty(B, device=dev), "loss3": torch.empty(3, device=dev)}
    f = lambda: kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, v, v, stats, out=out)
    f(); f(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(50): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print("skew %9d B: %.2f us per launch (kernel + finalize)" % (sk, e0.elapsed_time(e1) * 1e3 / 200), flush=True)
