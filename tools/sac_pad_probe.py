"""SAC-PFPN close-out: the twin critics' layer-0 GEMMs see a 233-column input (197 state + 36 action
columns; reference networks/actor_critic/sac.py:107-121) -- K = 233 is no multiple of 8.  What would padding the Q input
to 240 columns buy?  The step's layer-0 critic GEMMs at both replay batch sizes, K = 233 against K = 240, each tuned by
TunableOp first (both selections fresh, so that the comparison is not the shipped file against an untuned shape):
forward live critics over 2B rows, forward target critics over B rows, weight gradient over the B rows of the batch half.
Kill criterion of the review: under 2 % of the step (0.373 ms at 1024, 0.2024 ms at 256) -> stop."""
import os
import sys

import torch

R_ = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, "tools"))
from pfpn_amd import tuning  # noqa: E402

tuning.enable_tuned_gemms(tune_missing=True, max_duration_ms=60, max_iterations=40)
from sac_actor_probe import time_in_graph  # noqa: E402

dev = torch.device("cuda", 0)
for B, step_us in ((1024, 373.0), (256, 202.4)):
    total = {}
    for K in (233, 240):
        x2 = torch.randn(2 * B, K, device=dev)
        x1 = torch.randn(B, K, device=dev)
        w = torch.randn(K, 2048, device=dev) * 0.05
        d = torch.randn(B, 2048, device=dev)
        h2, h1, gw = torch.empty(2 * B, 2048, device=dev), torch.empty(B, 2048, device=dev), torch.empty(K, 2048, device=dev)
        for _ in range(3):                                      # TunableOp picks its kernels here
            torch.mm(x2, w, out=h2); torch.mm(x1, w, out=h1); torch.mm(x1.t(), d, out=gw)
        torch.cuda.synchronize()
        t_live = time_in_graph(lambda: torch.mm(x2, w, out=h2))
        t_tgt = time_in_graph(lambda: torch.mm(x1, w, out=h1))
        t_dw = time_in_graph(lambda: torch.mm(x1.t(), d, out=gw))
        total[K] = t_live + t_tgt + t_dw
        print("B %4d K %3d: forward live (2B rows) %.2f us, forward target %.2f us, dW %.2f us, sum %.2f us"
              % (B, K, t_live, t_tgt, t_dw, total[K]), flush=True)
    gain = total[233] - total[240]
    print("B %4d: padding 233 -> 240 saves %.2f us of a %.1f-us step = %.2f %%" % (B, gain, step_us, 100 * gain / step_us), flush=True)
