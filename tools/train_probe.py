"""Learner step rate at the reference's own minibatch size (32 rows, deepmimic_dppo_base.py) from HOST numpy batches:
per-minibatch net.train() (eager / self-captured hipGraph) vs one staged training round (`train_round`: the rollout
goes to the device once, minibatches are gathered there, losses come back once)."""
import os, sys, time
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_dppo_step import make_net, make_batch
from pfpn_amd.models import SyncReplicasApplier
from pfpn_amd.models.distributed_model import minibatch_plan
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
names = ["state", "action", "value", "log_prob", "advantage"]
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
# the reference schedule: one worker's 512-step unroll, 16 minibatches of 32 (deepmimic_dppo_base.py:6-12)
os.environ["PFPN_GRAPH_TRAIN"] = "1"
for n, bs, rnd in ((512, 32, "1"), (512, 32, "0"), (4096, 256, "1"), (4096, 256, "0")):
    net = make_net(dev, seed=1, resample_interval=10 ** 9)
    net.ROUND_STEPS_PER_REPLAY = 8 if rnd == "1" else 10 ** 9      # 8 minibatch steps per graph replay / one-step replays only
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    b = make_batch(rng, n); b["log_prob"] = np.full(n, -40.0, np.float32)
    cols = {k: b[k] for k in names}
    r = np.random.RandomState(0)
    for _ in range(3): net.train_round(opt, [], names, cols, minibatch_plan(n, bs, 1, r))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R_ = 20
    for _ in range(R_): net.train_round(opt, [], names, cols, minibatch_plan(n, bs, 1, r))
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / (R_ * (n // bs))
    print("train_round(), %d rows in %d-row minibatches, %s steps per replay: %.3f ms/step -> %.0f steps/s, %.0f transitions/s"
          % (n, bs, "8" if rnd == "1" else "1", t * 1e3, 1 / t, bs / t))
