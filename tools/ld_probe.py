"""Does the leading dimension of the normalised state (S = 197 floats = 788 B: rows not 16-B aligned) cost the layer-0
GEMMs?  Times x @ W0 ([B,197] x [197,2048]) and x^T @ dY ([197,B] x [B,2048]) with x stored at ld = 197 / 200 / 208 / 256,
each with TunableOp's pick for that layout."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(30); torch.cuda.tunable.set_max_tuning_iterations(20)
torch.cuda.tunable.set_filename("/tmp/ld_probe_tunable.csv")
dev = torch.device("cuda:0")
B, S, H = int(os.environ.get("LB", 4096)), 197, 2048
g = torch.Generator(device=dev); g.manual_seed(0)
W = torch.randn(S, H, device=dev, generator=g) * 0.05
dY = torch.randn(B, H, device=dev, generator=g)
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for ld in (197, 200, 208, 224, 256):
    buf = torch.zeros(B, ld, device=dev)
    x = buf[:, :S]
    x.copy_(torch.randn(B, S, device=dev, generator=g))
    y = torch.empty(B, H, device=dev); dW = torch.empty(S, H, device=dev)
    t_f = timeit(lambda: torch.mm(x, W, out=y))
    t_w = timeit(lambda: torch.mm(x.t(), dY, out=dW))
    fl = 2.0 * B * S * H
    print("ld=%3d  fwd %6.2f us (%5.1f TF)   dW %6.2f us (%5.1f TF)" % (ld, t_f, fl / t_f / 1e6, t_w, fl / t_w / 1e6))
# padded K as well: x [B,208] zero-padded against W [208,2048] zero rows
for kp in (200, 208, 224, 256):
    xb = torch.zeros(B, kp, device=dev); xb[:, :S].copy_(torch.randn(B, S, device=dev, generator=g))
    Wp = torch.zeros(kp, H, device=dev); Wp[:S].copy_(W)
    y = torch.empty(B, H, device=dev); dWp = torch.empty(kp, H, device=dev)
    t_f = timeit(lambda: torch.mm(xb, Wp, out=y))
    t_w = timeit(lambda: torch.mm(xb.t(), dY, out=dWp))
    fl = 2.0 * B * S * H
    print("K padded to %3d  fwd %6.2f us (%5.1f useful TF)   dW %6.2f us (%5.1f useful TF)" % (kp, t_f, fl / t_f / 1e6, t_w, fl / t_w / 1e6))
