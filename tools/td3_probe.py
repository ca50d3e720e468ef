"""TD3-PFPN learner step rate through net.train() (host batches): eager launches vs the self-captured hipGraphs (one per kind
of step: policy update / policy-delay)."""
import os, sys, time
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_td3_step import make_net
from pfpn_amd.models import SyncReplicasApplier
from pfpn_amd import tuning
tuning.enable_tuned_gemms(tune_missing=False)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
F = np.float32
for B in (256, 1024):
    for mode in ("0", "1"):
        os.environ["PFPN_GRAPH_TRAIN"] = mode
        net = make_net(dev)
        opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4}, {"names": net.actor_param_names(), "lr": 1e-4}],
                                  norm_clip=1.0, world_size=1)
        data = [torch.from_numpy(a).to(dev) for a in (rng.normal(0, 1, (B, 197)).astype(F), np.tanh(rng.normal(0, 1, (B, 36))).astype(F),
                                                      rng.random(B).astype(F), np.ones(B, F), rng.normal(0, 1, (B, 197)).astype(F))]
        for _ in range(14): net.train(None, opt, [], *data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): net.train(None, opt, [], *data)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 100
        print("TD3-PFPN net.train(), replay batch %4d, %s: %.3f ms/step -> %.0f steps/s" % (B, "hipGraphs" if mode == "1" else "eager    ", t * 1e3, 1 / t), flush=True)
