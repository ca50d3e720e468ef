"""Rollout inference cost per environment step: the reference's one-state `run()` vs `run_batch(E)`."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from test_gpu_dppo_step import make_net
from pfpn_amd.env_pool import BatchedPPORollout, EnvPool
from pfpn_amd.envs import SyntheticDeepMimicEnv

dev = torch.device("cuda", 0)
net = make_net(dev, seed=0)
rng = np.random.default_rng(0)
s1 = rng.normal(0, 1, 197).astype(np.float32)
for _ in range(20): net.run(None, s1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): net.run(None, s1)
t = (time.perf_counter() - t0) / 200
print("run() one state           : %7.1f us per call  -> %8.0f env steps/s" % (t * 1e6, 1 / t))
for E in (1, 8, 32, 128, 512):
    st = rng.normal(0, 1, (E, 197)).astype(np.float32)
    for _ in range(10): net.run_batch(st)[0].cpu()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): net.run_batch(st)[0].cpu()          # the action must reach the host every step
    t = (time.perf_counter() - t0) / 100
    print("run_batch(E=%3d)          : %7.1f us per call  -> %8.0f env steps/s" % (E, t * 1e6, E / t))
from pfpn_amd.graphs import GraphedRollout
for E in (1, 8, 128, 512):
    st = rng.normal(0, 1, (E, 197)).astype(np.float32)
    gr = GraphedRollout(net, E)
    for _ in range(10): gr(st)[0].cpu()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): gr(st)[0].cpu()
    t = (time.perf_counter() - t0) / 100
    print("GraphedRollout(E=%3d)     : %7.1f us per call  -> %8.0f env steps/s" % (E, t * 1e6, E / t))
for E, P in ((8, 4), (32, 8)):
    pool = EnvPool([(lambda i=i: SyntheticDeepMimicEnv(seed=i)) for i in range(E)], processes=P)
    ro = BatchedPPORollout(net, pool, 64)
    ro.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ro.collect()
    t = time.perf_counter() - t0
    print("EnvPool E=%2d (%d procs) + batched policy, 64-step unroll incl. GAE: %.1f ms -> %.0f env steps/s" % (E, P, t * 1e3, E * 64 / t))
    pool.close()

# overlapped rollout: one half's environments step on the host while the GPU runs the other half's policy call
# (BatchedPPORollout(overlap=True)), against the sequential pool, at several synthetic step costs (PyBullet DeepMimic ~1 ms)
for cost in (0.0, 100e-6, 300e-6, 1e-3):
    for E, P in ((16, 8),):
        res = {}
        for overlap in (False, True):
            pool = EnvPool([(lambda i=i: SyntheticDeepMimicEnv(seed=i, step_cost=cost)) for i in range(E)], processes=P)
            ro = BatchedPPORollout(net, pool, 64, overlap=overlap)
            ro.collect()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ro.collect(); ro.collect()
            res[overlap] = E * 128 / (time.perf_counter() - t0)
            pool.close()
        print("step cost %4.0f us, E=%d, %d procs (%d envs each): sequential %.0f env steps/s, overlapped %.0f (%.2fx)"
              % (cost * 1e6, E, P, E // P, res[False], res[True], res[True] / res[False]), flush=True)
