"""Small-row learner steps: the hand-written dense family (csrc/rowdense_kernels.hip, one launch per layer and direction)
against PyTorch-ROCm GEMMs + epilogue kernels, same box, same process.  Single-replica step as a one-step hipGraph.
Usage: python tools/rowdense_probe.py [rows ...]      (PROBE_EXCHANGE=1: the data-parallel step over a world-1 RCCL group)"""
import os, sys, time
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
rows_list = [int(a) for a in sys.argv[1:]] or [32, 256, 512, 1024]
exchange = os.environ.get("PROBE_EXCHANGE", "0") == "1"
from pfpn_amd.tuning import enable_tuned_gemms
enable_tuned_gemms(tune_missing=True)
if exchange:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29618")
    dist.init_process_group("nccl", rank=0, world_size=1)
from test_gpu_dppo_step import make_net, make_batch
from pfpn_amd.models import SyncReplicasApplier
from pfpn_amd.networks.ops import DenseStack
from pfpn_amd import graphs
dev = torch.device("cuda", 0)
for rows in rows_list:
    res = {}
    for mode, rmax in (("library", 0), ("rowdense", 1 << 20)):
        DenseStack.ROW_MAX = rmax
        net = make_net(dev, seed=1, resample_interval=10 ** 9)
        opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1, force_exchange=exchange)
        b = make_batch(np.random.default_rng(0), rows); b["log_prob"] = np.full(rows, -40.0, np.float32)
        args = [torch.from_numpy(b[k]).to(dev) for k in ("state", "action", "value", "log_prob", "advantage")]
        K = 1 if exchange else int(os.environ.get("PROBE_STEPS_PER_REPLAY", "8"))      # (a worker's training round replays
        step = graphs.GraphedLearnerStep(net, opt, *args, steps_per_replay=K)           #  eight minibatch steps per graph)
        for _ in range(20): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300 // K): step()
        torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / (300 // K * K) * 1e6
        del net, opt, step
    print("rows %5d: library %.1f us/step, rowdense %.1f us/step" % (rows, res["library"], res["rowdense"]), flush=True)
