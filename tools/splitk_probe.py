"""The two skinny weight-gradient GEMMs of the 4096-row DPPO step (head: 512 x 1260 from K = 4096 rows, 104 TFLOP/s in the
step; layer 0: 197 x 2048, 100 TFLOP/s) as ONE GEMM against a row-split batch of S partial products (bmm over S chunks of
the rows; the S partials would be added by the step's reduction launch).  TunableOp tunes every shape it meets here."""
import os, sys
import torch
R_ = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tools"))
from pfpn_amd import tuning
tuning.enable_tuned_gemms(tune_missing=True, max_duration_ms=100, max_iterations=50)
from sac_actor_probe import time_in_graph  # noqa
dev = torch.device("cuda", 0)
B = 4096
for name, K_in, N_out in (("head dW", 512, 1260), ("layer-0 dW", 197, 2048), ("head dW, padded 1280", 512, 1280)):
    a = torch.randn(B, K_in, device=dev)
    d = torch.randn(B, N_out, device=dev)
    out = torch.empty(K_in, N_out, device=dev)
    res = ["%s: mm %.1f us" % (name, time_in_graph(lambda: torch.mm(a.t(), d, out=out)))]
    for S in (2, 4, 8):
        outs = torch.empty(S, K_in, N_out, device=dev)
        a3, d3 = a.view(S, B // S, K_in), d.view(S, B // S, N_out)
        t = time_in_graph(lambda: torch.bmm(a3.transpose(1, 2), d3, out=outs))
        res.append("S=%d bmm %.1f us (+ %.1f MB of partials)" % (S, t, S * K_in * N_out * 4 / 1e6))
    print(" | ".join(res), flush=True)
