"""Offline GEMM selection for the learner step: run the step under PyTorch TunableOp with a generous budget, write the
chosen hipBLASLt / rocBLAS solutions to a CSV, then time the step with exactly that file loaded and tuning OFF.
Usage (GPU box): python tools/tune_gemms.py <out.csv> [iterations] [base.csv] ; the result is committed as
pfpn_amd/tuned/tunableop_gfx950.csv and loaded by bench.py (pfpn_amd/tuning.py).  With `base.csv` the selections already in
that file are kept and only the shapes it lacks are tuned (round 2: the 512 / 2048-row shards of the strong-scaling run and
the 128 x 256 head of BASELINE.json config 5)."""
import os, sys, time
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
out = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(200)
torch.cuda.tunable.set_max_tuning_iterations(iters)
torch.cuda.tunable.set_filename(out)
if len(sys.argv) > 3:
    assert torch.cuda.tunable.read_file(sys.argv[3]), "base selection rejected (validators differ?)"
from test_gpu_dppo_step import make_net, make_batch
from test_gpu_sac_step import make_net as make_sac
from pfpn_amd.models import SyncReplicasApplier
from pfpn_amd import graphs
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)


def dppo(B):
    net = make_net(dev)
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    b = make_batch(rng, B); b["log_prob"] = np.full(B, -40.0, np.float32)
    args = [torch.from_numpy(b[k]).to(dev) for k in ("state", "action", "value", "log_prob", "advantage")]
    return net, opt, args


net, opt, args = dppo(4096)
for _ in range(3):
    net.train_device(opt, *args)
for B in (32, 64, 256, 512, 1024, 2048):           # minibatch sizes of the shipped settings / SAC replay batches / DP shards
    n2, o2, a2 = dppo(B)
    for _ in range(2):
        n2.train_device(o2, *a2)
# BASELINE.json config 5 (128 action dims x 256 particles): one GPU's 1024-row shard
from pfpn_amd.networks import ParticleFilteringClipPPONetwork
import test_gpu_dppo_step as tds
c5 = tds.make_net(dev, A=128, particles=256)
c5o = SyncReplicasApplier(c5, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
c5b = make_batch(rng, 1024, A=128); c5b["log_prob"] = np.full(1024, -300.0, np.float32)
c5a = [torch.from_numpy(c5b[k]).to(dev) for k in ("state", "action", "value", "log_prob", "advantage")]
for _ in range(2):
    c5.train_device(c5o, *c5a)
del c5, c5o, c5a
# rollout inference (one state per call, and a pool of environments per call)
for E in (1, 8, 64):
    st = torch.randn(E, 197, device=dev)
    for _ in range(2):
        net.run_batch(st)
# SAC-PFPN learner step at the reference's replay batch (256) and BASELINE.json config 3 (1024)
for B in (256, 1024):
    sac = make_sac(dev)
    so = SyncReplicasApplier(sac, [{"names": sac.critic_param_names(), "lr": 1e-4}, {"names": sac.actor_param_names(), "lr": 1e-4}],
                             norm_clip=1.0, world_size=1)
    sb = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(np.float32),
                                                rng.uniform(-0.9, 0.9, (B, 36)).astype(np.float32),
                                                rng.normal(0.5, 0.3, B).astype(np.float32),
                                                (rng.uniform(size=B) > 0.02).astype(np.float32),
                                                (rng.normal(0, 1, (B, 197)) * 2).astype(np.float32))]
    for _ in range(2):
        sac.train_device(so, *sb)
    sac.run_batch(torch.randn(1, 197, device=dev))
torch.cuda.synchronize()
torch.cuda.tunable.tuning_enable(False)
step = graphs.GraphedLearnerStep(net, opt, *args)
for _ in range(20):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print("tuned file %s: %d entries; learner step %.1f us" % (out, len(torch.cuda.tunable.get_results()), (time.perf_counter() - t0) / 300 * 1e6))
