"""The layer-0 epilogue passes of the 4096-row DPPO step ([4096, 2048]: BiasAdd + Relu6 + mask bits; Relu6Grad from the bits +
bias-gradient partials) alone, replayed back to back (cache-warm) and each behind a GEMM that has just written its input."""
import os, sys
import torch
R_ = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tools"))
from pfpn_amd import kernels, tuning
tuning.enable_tuned_gemms(tune_missing=False)
from sac_actor_probe import time_in_graph  # noqa
dev = torch.device("cuda", 0)
B, H = 4096, 2048
y = torch.randn(B, H, device=dev)
bias = torch.randn(H, device=dev)
mask = kernels.relu6_mask(B, H, dev)
ws = kernels.learner_workspace(B, H, dev)
x = torch.randn(B, 197, device=dev)
w = torch.randn(197, H, device=dev)
t_f = time_in_graph(lambda: kernels.bias_act_mask(y, bias, mask))
t_b = time_in_graph(lambda: kernels.act_bwd_partial(None, y, 0, ws, mask=mask))
t_g = time_in_graph(lambda: torch.mm(x, w, out=y))


def gemm_then_f():
    torch.mm(x, w, out=y)
    kernels.bias_act_mask(y, bias, mask)


def gemm_then_b():
    torch.mm(x, w, out=y)
    kernels.act_bwd_partial(None, y, 0, ws, mask=mask)
t_gf, t_gb = time_in_graph(gemm_then_f), time_in_graph(gemm_then_b)
print("bias_act_mask [4096 x 2048]: alone %.2f us, behind the GEMM that wrote its input %.2f us | act_bwd_partial_mask: alone %.2f, "
      "behind a GEMM %.2f | (the GEMM: %.2f us); 67 MB each way = %.2f TB/s alone" % (t_f, t_gf - t_g, t_b, t_gb - t_g, t_g, 67.1 / t_f))
