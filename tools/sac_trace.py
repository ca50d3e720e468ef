"""The SAC-PFPN learner step (replay batch 1024) as a hipGraph, a few dozen replays: the program to put under
`rocprofv3 --kernel-trace` for a kernel-by-kernel timeline (`tools/step_timeline.py <trace> k_act_bwd_partial_value`: since
round 4 the noise draw and the next step's prologue ride in the optimiser launch and the loss head in the first launch of the
critics' backward pass, eight steps per replay; that launch marks a step)."""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_sac_step import make_net
from pfpn_amd.models import SyncReplicasApplier
from pfpn_amd.graphs import GraphedLearnerStep
from pfpn_amd import tuning
tuning.enable_tuned_gemms(tune_missing=False)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
net = make_net(dev)
opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4}, {"names": net.actor_param_names(), "lr": 1e-4}],
                          norm_clip=1.0, world_size=1)
b = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(np.float32),
                                           rng.uniform(-0.9, 0.9, (B, 36)).astype(np.float32),
                                           rng.normal(0.5, 0.3, B).astype(np.float32),
                                           (rng.uniform(size=B) > 0.02).astype(np.float32),
                                           (rng.normal(0, 1, (B, 197)) * 2).astype(np.float32))]
step = GraphedLearnerStep(net, opt, *b, noise=net.make_train_noise(B), steps_per_replay=8)
for _ in range(12):
    step()
torch.cuda.synchronize()
print("done", step.loss3.cpu().numpy())
