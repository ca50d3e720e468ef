"""SAC-PFPN learner step (8 steps per replay, noise drawn inside the graph) with one of the step's launch-saving forms off / on
(argv[1]: fuse_actor (default) | ride_loss | pair_critics), alternating on ONE box, replay batches 256 / 1024."""
import os, sys, time
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_sac_step import make_net
from pfpn_amd.models import SyncReplicasApplier
from pfpn_amd import tuning, graphs
tuning.enable_tuned_gemms(tune_missing=False)
dev = torch.device("cuda", 0)
ATTR = sys.argv[1] if len(sys.argv) > 1 else "fuse_actor"
for B in (256, 1024):
    for rep in range(3):
        for fuse in (False, True):
            net = make_net(dev)
            setattr(net, ATTR, fuse)
            opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                            {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1)
            rng = np.random.default_rng(1)
            b = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(np.float32),
                                                       rng.uniform(-0.9, 0.9, (B, 36)).astype(np.float32),
                                                       rng.normal(0.5, 0.3, B).astype(np.float32),
                                                       (rng.uniform(size=B) > 0.02).astype(np.float32),
                                                       (rng.normal(0, 1, (B, 197)) * 2).astype(np.float32))]
            step = graphs.GraphedLearnerStep(net, opt, *b, noise=net.make_train_noise(B), steps_per_replay=8)
            for _ in range(6): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): step()
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / (40 * 8)
            print("B %4d %s %-5s %.4f ms/step loss %s" % (B, ATTR, fuse, t * 1e3, step.loss3.cpu().numpy()), flush=True)
            del step, net, opt
