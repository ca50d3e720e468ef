"""GPU: the reference-shaped launcher runs rollout -> GAE -> minibatch learner steps end to end on the
synthetic DeepMimic-shaped environment (PyBullet is absent in the image), for DPPO and SAC."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_dppo_launcher_end_to_end(dev):
    from pfpn_amd import main
    st = main.parse(["--env", "DeepMimicWalk-v0", "--seed", "34114", "--setting",
                     "settings.deepmimic.deepmimic_dppo_pfpn", "--particles", "35", "--train"])
    st.worker_opts = dict(st.worker_opts, unroll_length=64)
    st.max_iterations = 6
    model = main.build(st)
    worker = model.start({"worker": []})
    net = worker.global_net
    assert worker.global_step >= 6 and net.global_step == worker.global_step
    assert net.sum_active.sum().item() > 0          # rollouts accumulated activation statistics
    assert all(np.isfinite(v.detach().cpu().numpy()).all() for v in net.variables().values())
    # north_star alias: worker.step(batch)
    rng = np.random.default_rng(0)
    B = 32
    batch = {"state": rng.normal(0, 1, (B, 197)).astype(np.float32), "action": rng.uniform(-1, 1, (B, 36)).astype(np.float32),
             "value": np.zeros(B, np.float32), "log_prob": np.full(B, -40, np.float32),
             "advantage": rng.normal(0, 1, B).astype(np.float32)}
    worker.request_stop = False
    loss, res = worker.step(batch)
    assert np.isfinite(loss[0]) and res == [worker.global_step]


def test_sac_launcher_end_to_end(dev):
    from pfpn_amd import main
    st = main.parse(["--env", "DeepMimicKick-v0", "--seed", "33406", "--setting",
                     "settings.deepmimic.deepmimic_sac_pfpn", "--particles", "35", "--train"])
    st.worker_opts = dict(st.worker_opts, batch_size=32, buffer_capacity=2000)
    st.max_iterations = 5
    model = main.build(st)
    worker = model.start({"worker": []})
    net = worker.global_net
    assert worker.global_step >= 5
    assert net._target_initialised
    # target Q moved towards the eval Q by tau per step but is not equal to it
    ev, tg = net.variables(), net.target_variables()
    d = (ev["critic/q1/fc1/weight"] - tg["critic/q1/fc1/weight"]).abs().max().item()
    assert 0 < d < 1e-2
    assert all(np.isfinite(v.detach().cpu().numpy()).all() for v in ev.values())
