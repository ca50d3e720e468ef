"""BASELINE.json config 5 as a learner step on one GPU's 1024-row shard (128 action dims x 256 particles, head 512 -> 32768)
as a hipGraph of 8 steps per replay, a few replays: the program to put under `rocprofv3 --kernel-trace`
(`tools/step_timeline.py <trace> k_mix_wide`)."""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import bench
from pfpn_amd.networks import ParticleFilteringClipPPONetwork
from pfpn_amd.models import SyncReplicasApplier
from pfpn_amd import graphs, tuning
tuning.enable_tuned_gemms(tune_missing=False)
dev = torch.device("cuda", 0)
cA, cN, cB, S = 128, 256, 1024, bench.S
net = ParticleFilteringClipPPONetwork(trainable=True, state_shape=[S], action_shape=[cA], action_lower_bound=-np.ones(cA),
                                      action_upper_bound=np.ones(cA), device=dev, seed=bench.SEED,
                                      **dict(bench.NET_OPTS, particles=cN))
net.init()
opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
rng = np.random.default_rng(5)
st = torch.from_numpy((rng.normal(0, 1, (cB, S)) * 2 + 0.3).astype(np.float32)).to(dev)
ac = torch.from_numpy(rng.uniform(-1, 1, (cB, cA)).astype(np.float32)).to(dev)
adv = torch.from_numpy(rng.normal(0, 1, cB).astype(np.float32)).to(dev)
with torch.no_grad():
    p, _, hc = net._forward(st)
    lp = p.log_prob(ac).clone()
    v = net._value(hc).clone()
args = (st, ac, v, lp, adv)
for _ in range(3):
    net.train_device(opt, *args)
K = 8
step = graphs.GraphedLearnerStep(net, opt, *[t.clone() for t in args], steps_per_replay=K)
import time
for _ in range(3):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(6):
    step()
torch.cuda.synchronize()
print("config-5 learner step, 1024 rows: %.4f ms, loss %s" % ((time.perf_counter() - t0) / (6 * K) * 1e3, step.loss3.cpu().numpy()))
