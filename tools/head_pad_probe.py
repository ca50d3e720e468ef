"""Does padding the categorical head's width (A*N = 1260 columns) buy faster GEMMs?  Times the three fc_policy GEMMs of the
4096-row learner step (forward addmm, dX, dW) under TunableOp at the real width and at padded widths, each both with a
compact operand (ld = width) and as a 1260-column view of a padded buffer (ld = padded width, width unchanged).
Usage (GPU box): python tools/head_pad_probe.py [rows]"""
import sys, time, torch
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(150)
torch.cuda.tunable.set_max_tuning_iterations(60)
torch.cuda.tunable.set_filename("/tmp/head_pad_probe.csv")
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = 512


def timed(fn, n=200):
    for _ in range(5):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // 20):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n // 20 * 20) * 1e6


def case(N, ld):
    """N logical columns inside buffers of row stride ld"""
    h = torch.randn(B, H, device=dev)
    wbuf = torch.randn(H, ld, device=dev) * 0.01
    bias = torch.zeros(ld, device=dev)
    lbuf = torch.empty(B, ld, device=dev)
    dl = torch.randn(B, ld, device=dev)
    gwb = torch.empty(H, ld, device=dev)
    dh = torch.empty(B, H, device=dev)
    w, b, lo, d, gw = wbuf[:, :N], bias[:N], lbuf[:, :N], dl[:, :N], gwb[:, :N]
    f = timed(lambda: torch.addmm(b, h, w, out=lo))
    dx = timed(lambda: torch.mm(d, w.t(), out=dh))
    dw = timed(lambda: torch.mm(h.t(), d, out=gw))
    print("rows %d  N %4d ld %4d : fwd %6.1f  dX %6.1f  dW %6.1f  sum %6.1f us" % (B, N, ld, f, dx, dw, f + dx + dw), flush=True)


for N, ld in ((1260, 1260), (1280, 1280), (1260, 1280), (1264, 1264), (1344, 1344), (1536, 1536), (1024, 1024)):
    case(N, ld)
