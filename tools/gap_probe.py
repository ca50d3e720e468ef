"""Where do the ~5.7 us gaps on both sides of the mixture kernel in the step timeline come from?  Sequences of kernels
captured in hipGraphs (20 repetitions each), timed with events: pair time minus the single-kernel times = boundary cost."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels, tuning
dev = torch.device("cuda:0")
if os.environ.get("GAP_TUNED", "1") == "1":
    try:
        tuning.enable_tuned_gemms(tune_missing=False); print("tuned GEMM selection on")
    except Exception as e:
        print("tuning.enable failed:", e)
B, A, N, H = 4096, 36, 35, 512
g = torch.Generator(device=dev); g.manual_seed(0)
ha = torch.randn(B, H, device=dev, generator=g)
W = torch.randn(H, A * N, device=dev, generator=g) * 0.05
bias = torch.zeros(A * N, device=dev)
logits = torch.empty(B, A * N, device=dev)
x = torch.rand(B, A, device=dev, generator=g) * 2.2 - 1.1
loc = torch.linspace(-1, 1, N, device=dev).repeat(A, 1).contiguous()
logstd = torch.full((A, N), float(np.log(2 / 34)), device=dev)
torch.addmm(bias, ha, W, out=logits)
lp_old = kernels.logprob_fwd(logits, x, loc, logstd)
adv = torch.randn(B, device=dev, generator=g); v = torch.randn(B, device=dev, generator=g); vo = torch.randn(B, device=dev, generator=g)
st = kernels.adv_stats(adv)
out = kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, v, vo, st, finalize=False)
dW = torch.empty(H, A * N, device=dev); dh = torch.empty(B, H, device=dev)
big = torch.randn(B, 1024, device=dev, generator=g); big2 = torch.empty_like(big)

def f_addmm(): torch.addmm(bias, ha, W, out=logits)
def f_mix(): kernels.ppo_mixture_fwd_bwd(logits, x, loc, logstd, lp_old, adv, v, vo, st, out=out, finalize=False)
def f_dw(): torch.mm(ha.t(), out["dlogits"], out=dW)
def f_dx(): torch.mm(out["dlogits"], W.t(), out=dh)
def f_ew(): torch.mul(big, 2.0, out=big2)
def f_fwd(): kernels.logprob_fwd(logits, x, loc, logstd)

def timeit(fs, n=20, reps=10):
    def run():
        for f in fs: f()
    for _ in range(3): run()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        run(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(n): run()
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)

single = {}
for name, f in (("addmm", f_addmm), ("mix", f_mix), ("dW", f_dw), ("dX", f_dx), ("ew", f_ew), ("fwd", f_fwd)):
    single[name] = timeit([f])
    print("single %-6s %7.2f us" % (name, single[name]))
for names, fs in ((("addmm", "mix"), [f_addmm, f_mix]), (("mix", "dW"), [f_mix, f_dw]), (("mix", "dX"), [f_mix, f_dx]),
                  (("addmm", "mix", "dW", "dX"), [f_addmm, f_mix, f_dw, f_dx]), (("ew", "mix"), [f_ew, f_mix]),
                  (("addmm", "ew"), [f_addmm, f_ew]), (("addmm", "fwd"), [f_addmm, f_fwd]), (("addmm", "dW"), [f_addmm, f_dw]),
                  (("ew", "dW"), [f_ew, f_dw])):
    t = timeit(fs)
    print("seq %-22s %7.2f us   sum of singles %7.2f   extra %+6.2f" % ("+".join(names), t, sum(single[n] for n in names),
                                                                      t - sum(single[n] for n in names)))
