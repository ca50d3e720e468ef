"""Round 6: the three fc_policy GEMMs of the 4096-row learner step (forward addmm, dX, dW; head width A*N = 1260) issued as TWO
GEMMs over column ranges [0, c) and [c, 1260) of the same buffers (no layout change: column views with the row stride 1260; the
second dX product accumulates with beta = 1) -- does a power-of-two-friendly first part beat the library's one 1260-wide GEMM
(0.66-0.71 of the fp32 MFMA peak in profiles/r06_step_timeline.txt, against 0.89-0.93 for the 512 / 1024-wide layers)?
TunableOp tunes every shape.  Usage (GPU box): python tools/head_split_probe.py [rows]"""
import sys, time, torch
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(100)
torch.cuda.tunable.set_max_tuning_iterations(50)
torch.cuda.tunable.set_filename("/tmp/head_split_probe.csv")
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H, N = 512, 1260


def timed(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n // 20):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (n // 20 * 20) * 1e6)
    return best


h = torch.randn(B, H, device=dev)
w = torch.randn(H, N, device=dev) * 0.01
bias = torch.randn(N, device=dev) * 0.01
lo = torch.empty(B, N, device=dev)
d = torch.randn(B, N, device=dev)
gw = torch.empty(H, N, device=dev)
dh = torch.empty(B, H, device=dev)
ref_lo = torch.addmm(bias, h, w)
ref_dh = torch.mm(d, w.t())
ref_gw = torch.mm(h.t(), d)


def fwd(cuts):
    def f():
        for a, b in cuts:
            torch.addmm(bias[a:b], h, w[:, a:b], out=lo[:, a:b])
    return f


def dx(cuts):
    def f():
        for i, (a, b) in enumerate(cuts):
            if i == 0:
                torch.mm(d[:, a:b], w[:, a:b].t(), out=dh)
            else:
                torch.addmm(dh, d[:, a:b], w[:, a:b].t(), out=dh)
    return f


def dw(cuts):
    def f():
        for a, b in cuts:
            torch.mm(h.t(), d[:, a:b], out=gw[:, a:b])
    return f


for cs in ((), (1024,), (768,), (512,), (640,), (1152,), (896,), (256, 1024 + 256), (512, 1024)):
    edges = [0] + list(cs) + [N]
    cuts = [(edges[i], edges[i + 1]) for i in range(len(edges) - 1) if edges[i + 1] > edges[i]]
    lo.zero_(); dh.zero_(); gw.zero_()
    tf, tx, tw = timed(fwd(cuts)), timed(dx(cuts)), timed(dw(cuts))
    fwd(cuts)(); dx(cuts)(); dw(cuts)()
    torch.cuda.synchronize()
    err = [float((a - b).abs().max() / b.abs().max()) for a, b in ((lo, ref_lo), (dh, ref_dh), (gw, ref_gw))]
    print("rows %d cuts %-22s fwd %6.1f  dX %6.1f  dW %6.1f  sum %6.1f us   max err / max|ref| %.1e %.1e %.1e"
          % (B, cuts, tf, tx, tw, tf + tx + tw, *err), flush=True)
