"""One dense launch of the small-row family repeated (for `rocprofv3 --pmc`): the layer-1 forward of the DPPO trunks
(2 x [rows x 1024 x 512], bias + ReLU6 + critic-head partials).  Usage: python tools/rowdense_pmc.py rows cfg"""
import os, sys
import torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from pfpn_amd import kernels as K
dev = torch.device("cuda", 0)
B, cfg = int(sys.argv[1]), int(sys.argv[2])
H0, H1 = 1024, 512
r = lambda *s: torch.randn(*s, device=dev)
act0 = r(B, 2 * H0).clamp(0, 6); a0 = act0.view(B, 2, H0).transpose(0, 1)
w1, b1, o1 = r(2, H0, H1) * 0.05, r(2, H1), torch.empty(2, B, H1, device=dev)
jobs = [K.dense_job_nn(a0, w1, o1, bias=b1, relu6=True, cfg=cfg)]
for _ in range(60):
    K.dense_jobs(jobs)
torch.cuda.synchronize()
