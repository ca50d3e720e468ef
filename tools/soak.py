import os, sys, time
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from pfpn_amd import main
t0 = time.time()
st = main.parse(["--env", "DeepMimicWalk-v0", "--seed", "1", "--setting", "settings.deepmimic.deepmimic_dppo_pfpn", "--particles", "35", "--train"])
st.worker_opts = dict(st.worker_opts, unroll_length=128)
st.network_opts = dict(st.network_opts, resample_interval=37)
st.max_iterations = 800
model = main.build(st)
worker = model.start({"worker": []})
net = worker.global_net
ok = all(np.isfinite(v.detach().cpu().numpy()).all() for v in net.variables().values())
print("dppo steps", worker.global_step, "finite", ok, "graphs", len(net._train_graphs), len(getattr(net, "_round_graphs", {})), "mem MB", torch.cuda.memory_allocated() >> 20, "t %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
st = main.parse(["--env", "DeepMimicKick-v0", "--seed", "2", "--setting", "settings.deepmimic.deepmimic_sac_pfpn", "--particles", "35", "--train"])
st.worker_opts = dict(st.worker_opts, batch_size=256, buffer_capacity=5000)
st.network_opts = dict(st.network_opts, resample_interval=53)
st.max_iterations = 600
model = main.build(st)
worker = model.start({"worker": []})
net = worker.global_net
ok = all(np.isfinite(v.detach().cpu().numpy()).all() for v in net.variables().values())
print("sac steps", worker.global_step, "finite", ok, "graphs", len(net._train_graphs), "mem MB", torch.cuda.memory_allocated() >> 20, "t %.1f s" % (time.time() - t0), flush=True)
