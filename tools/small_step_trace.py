"""The DPPO learner step at a small minibatch (default 32 rows) as a hipGraph, a few dozen replays: the program to put under
`rocprofv3 --kernel-trace` for a kernel-by-kernel timeline (`tools/step_timeline.py <trace>`)."""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_dppo_step import make_net, make_batch
from pfpn_amd.models import SyncReplicasApplier
from pfpn_amd.graphs import GraphedLearnerStep
from pfpn_amd import tuning
tuning.enable_tuned_gemms(tune_missing=False)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(0)
net = make_net(dev, seed=1, resample_interval=10 ** 9)
opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
b = make_batch(rng, B); b["log_prob"] = np.full(B, -40.0, np.float32)
args = [torch.from_numpy(b[k]).to(dev) for k in ("state", "action", "value", "log_prob", "advantage")]
g = GraphedLearnerStep(net, opt, *args, steps_per_replay=8)
for _ in range(12):
    g()
torch.cuda.synchronize()
print("done")
