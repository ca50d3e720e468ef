"""The data-parallel learner step of ONE rank at `rows` rows (default 512 = a rank of the 8-GPU strong-scaling run):
per-replica clip, bucket all-reduce through RCCL (world size 1), Adam with 1/W, as two hipGraphs.  Run under
`rocprofv3 --kernel-trace` and feed the trace to tools/step_timeline.py.  Usage: python tools/shard_probe.py [rows] [fused]"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fused = len(sys.argv) > 2 and sys.argv[2] == "fused"          # the single-replica step instead (no exchange)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29617")
from pfpn_amd.tuning import enable_tuned_gemms
enable_tuned_gemms(tune_missing=True)
dist.init_process_group("nccl", rank=0, world_size=1)
from test_gpu_dppo_step import make_net, make_batch
from pfpn_amd.models import SyncReplicasApplier
from pfpn_amd import graphs
dev = torch.device("cuda", 0)
net = make_net(dev, seed=1, resample_interval=10 ** 9)
opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1, force_exchange=not fused)
b = make_batch(np.random.default_rng(0), rows); b["log_prob"] = np.full(rows, -40.0, np.float32)
args = [torch.from_numpy(b[k]).to(dev) for k in ("state", "action", "value", "log_prob", "advantage")]
step = graphs.GraphedLearnerStep(net, opt, *args)
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 200
print("%s step, %d rows: %.1f us/step" % ("fused single-replica" if fused else "data-parallel (exchange)", rows, t * 1e6))
dist.destroy_process_group()
