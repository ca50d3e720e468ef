"""SAC-PFPN learner step rate (BASELINE.json config 3: replay batch 1024) -- reported in DESIGN.md, not the bench metric."""
import os, sys, time
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_sac_step import make_net
from pfpn_amd.models import SyncReplicasApplier
from pfpn_amd import tuning
tuning.enable_tuned_gemms(tune_missing=False)
dev = torch.device("cuda", 0)
net = make_net(dev)
groups = [{"names": net.critic_param_names(), "lr": 1e-4}, {"names": net.actor_param_names(), "lr": 1e-4}]
opt = SyncReplicasApplier(net, groups, norm_clip=1.0, world_size=1)
rng = np.random.default_rng(0)
for B in (256, 1024, 4096):
    b = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(np.float32),
                                               rng.uniform(-0.9, 0.9, (B, 36)).astype(np.float32),
                                               rng.normal(0.5, 0.3, B).astype(np.float32),
                                               (rng.uniform(size=B) > 0.02).astype(np.float32),
                                               (rng.normal(0, 1, (B, 197)) * 2).astype(np.float32))]
    def step():
        return net.train(None, opt, [], *b)
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for _ in range(n): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    print("SAC-PFPN learner step, replay batch %4d: %.3f ms/step -> %.0f steps/s" % (B, t * 1e3, 1 / t))

from pfpn_amd.graphs import GraphedLearnerStep
for B in (256, 1024, 4096):
  for both in ("0", "1", "0", "1"):
    net = make_net(dev)
    net.batch_actor = net.batch_critic = both == "1"   # the actor once over [s; s'], the live critic once over [(s, a_pi); (s, a_hist)]
    opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4}, {"names": net.actor_param_names(), "lr": 1e-4}],
                              norm_clip=1.0, world_size=1)
    rng = np.random.default_rng(1)
    b = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(np.float32),
                                               rng.uniform(-0.9, 0.9, (B, 36)).astype(np.float32),
                                               rng.normal(0.5, 0.3, B).astype(np.float32),
                                               (rng.uniform(size=B) > 0.02).astype(np.float32),
                                               (rng.normal(0, 1, (B, 197)) * 2).astype(np.float32))]
    step = GraphedLearnerStep(net, opt, *b, noise=net.make_train_noise(B))
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 200
    print("SAC-PFPN learner step as one hipGraph, replay batch %4d, actor / live critic %s: %.3f ms/step -> %.0f steps/s (loss %s)" % (
        B, "one pass over 2B rows" if both == "1" else "one pass per input", t * 1e3, 1 / t, step.loss3.cpu().numpy()))
