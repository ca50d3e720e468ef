"""Learner step rate at the reference's own minibatch size (32 rows, deepmimic_dppo_base.py) through net.train():
eager launches vs the self-captured hipGraph."""
import os, sys, time
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_dppo_step import make_net, make_batch
from pfpn_amd.models import SyncReplicasApplier
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for B in (32, 256, 4096):
    for mode in ("0", "1"):
        os.environ["PFPN_GRAPH_TRAIN"] = mode
        net = make_net(dev, seed=1, resample_interval=10 ** 9)
        opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
        b = make_batch(rng, B); b["log_prob"] = np.full(B, -40.0, np.float32)
        args = (b["state"], b["action"], b["value"], b["log_prob"], b["advantage"])
        for _ in range(10): net.train(None, opt, [], *args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): net.train(None, opt, [], *args)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 100
        print("net.train(), %4d-row minibatch, %s: %.3f ms/step -> %.0f steps/s" % (B, "hipGraph" if mode == "1" else "eager   ", t * 1e3, 1 / t))
