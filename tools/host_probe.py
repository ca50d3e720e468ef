"""Host enqueue time of one eager learner step (no hipGraph): must stay below the GPU time of the step, or the
data-parallel path (eager around the RCCL all-reduce) becomes host-bound.  Prints enqueue us/step and GPU us/step."""
import os, sys, time
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_dppo_step import make_net, make_batch
from pfpn_amd.models import SyncReplicasApplier
dev = torch.device("cuda", 0)
net = make_net(dev)
opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
rng = np.random.default_rng(0)
B = 4096
b = make_batch(rng, B); b["log_prob"] = np.full(B, -40.0, np.float32)
args = [torch.from_numpy(b[k]).to(dev) for k in ("state", "action", "value", "log_prob", "advantage")]
for _ in range(20):
    net.train_device(opt, *args)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    net.train_device(opt, *args)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("eager step: host enqueue %.1f us/step, wall %.1f us/step" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
# host-only cost: enqueue while the GPU is far behind is what the line above shows; for the pure host cost run tiny batches
if os.environ.get("PFPN_HOST_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(100):
        net.train_device(opt, *args)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
