import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from test_gpu_dppo_step import make_net
from pfpn_amd.graphs import GraphedRollout
dev = torch.device("cuda", 0)
net = make_net(dev, seed=0)
rng = np.random.default_rng(0)
E = int(os.environ.get("RE", 8))
st = rng.normal(0, 1, (E, 197)).astype(np.float32)
gr = GraphedRollout(net, E)
for _ in range(30): gr(st)[0].cpu()
