"""The twin critics' layer-1 GEMMs of the SAC step (replay batch 1024: 2048-row pass) as the step issues them (bmm over the
two groups, weight gradient from the second half of the rows) against per-group mm calls, shipped GEMM selection on."""
import os, sys
import torch
R_ = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tools"))
from pfpn_amd import tuning
tuning.enable_tuned_gemms(tune_missing=False)
from sac_actor_probe import time_in_graph  # noqa
dev = torch.device("cuda", 0)
for B in (1024, 256):
    act = torch.randn(2, 2 * B, 1024, device=dev)
    d = torch.randn(2, 2 * B, 512, device=dev)
    w = torch.randn(2, 1024, 512, device=dev)
    gw = torch.empty(2, 1024, 512, device=dev)
    dact = torch.empty(2, 2 * B, 1024, device=dev)
    h = torch.empty(2, 2 * B, 512, device=dev)
    t_dw_bmm = time_in_graph(lambda: torch.bmm(act[:, B:].transpose(1, 2), d[:, B:], out=gw))

    def dw_mm():
        for g in range(2):
            torch.mm(act[g, B:].t(), d[g, B:], out=gw[g])
    t_dw_mm = time_in_graph(dw_mm)
    t_dx_bmm = time_in_graph(lambda: torch.bmm(d, w.transpose(1, 2), out=dact))

    def dx_mm():
        for g in range(2):
            torch.mm(d[g], w[g].t(), out=dact[g])
    t_dx_mm = time_in_graph(dx_mm)
    t_f_bmm = time_in_graph(lambda: torch.bmm(act, w, out=h))

    def f_mm():
        for g in range(2):
            torch.mm(act[g], w[g], out=h[g])
    t_f_mm = time_in_graph(f_mm)
    print("B %4d  dW: bmm %.1f us, 2 x mm %.1f us | dX: bmm %.1f, 2 x mm %.1f | forward: bmm %.1f, 2 x mm %.1f"
          % (B, t_dw_bmm, t_dw_mm, t_dx_bmm, t_dx_mm, t_f_bmm, t_f_mm), flush=True)
# the policy head: bias inside the GEMM (torch.addmm -> a hipBLASLt bias-epilogue kernel) against a plain mm
for R in (2048, 512, 4096):
    hh = torch.randn(R, 512, device=dev)
    wp = torch.randn(512, 1260, device=dev)
    bp = torch.randn(1260, device=dev)
    out = torch.empty(R, 1260, device=dev)
    t_add = time_in_graph(lambda: torch.addmm(bp, hh, wp, out=out))
    t_mm = time_in_graph(lambda: torch.mm(hh, wp, out=out))
    print("head %4d rows: addmm %.1f us, mm %.1f us" % (R, t_add, t_mm), flush=True)
