"""Layer-0 GEMMs of the 4096-row step with the 197-wide state padded to 200 / 208 / 256 columns (zero columns / zero weight
rows): forward x [B,S] @ W [S, 2048] and weight gradient x^T [S,B] @ d [B, 2048].  TunableOp on.  us per GEMM."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd.tuning import enable_tuned_gemms
enable_tuned_gemms(tune_missing=True)
dev = torch.device("cuda")
B, H = 4096, 2048
def timed(fn, reps=20, replays=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)
for rnd in range(2):
    for S in (197, 200, 208, 224, 256):
        x = torch.randn(B, S, device=dev); w = torch.randn(S, H, device=dev) * 0.01; d = torch.randn(B, H, device=dev)
        y = torch.empty(B, H, device=dev); gw = torch.empty(S, H, device=dev)
        tf = timed(lambda: torch.mm(x, w, out=y))
        tw = timed(lambda: torch.mm(x.t(), d, out=gw))
        # the padded buffer read through a 197-column view (row stride S): what a padded xn with the logical width would cost
        xv = x[:, :197]; wv = w[:197]
        yv = torch.empty(B, H, device=dev); gv = torch.empty(197, H, device=dev)
        tfv = timed(lambda: torch.mm(xv, wv, out=yv)) if S != 197 else float("nan")
        twv = timed(lambda: torch.mm(xv.t(), d, out=gv)) if S != 197 else float("nan")
        print("round %d  S=%3d: fwd %.1f us  dW %.1f us | 197-wide views of the padded rows: fwd %.1f  dW %.1f" % (rnd, S, tf, tw, tfv, twv))
