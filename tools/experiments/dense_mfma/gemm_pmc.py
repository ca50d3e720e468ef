import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfpn_amd import kernels
dev = torch.device("cuda:0")
G, R, K, N = 2, 4096, 1024, 512
x = torch.randn(G, R, K, device=dev); w = torch.randn(G, K, N, device=dev); b = torch.randn(G, N, device=dev)
y = torch.empty(G, R, N, device=dev)
for _ in range(5): kernels.dense_fwd(x, w, b, y, True)
for _ in range(3): torch.bmm(x, w, out=y)
torch.cuda.synchronize()
