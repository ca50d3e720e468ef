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


@pytest.mark.parametrize("family", ["a3c", "impala"])
def test_a2c_and_vtrace_launchers_end_to_end(dev, family):
    """the A3C and IMPALA settings (synchronous replicas here): rollout on the synthetic environment -> A2C targets /
    V-trace -> learner steps with the entropy term, finite parameters, the step counters in line."""
    from pfpn_amd import main
    st = main.parse(["--env", "DeepMimicWalk-v0", "--seed", "3", "--setting",
                     "settings.deepmimic.deepmimic_%s_pfpn" % family, "--particles", "35", "--train"])
    st.max_iterations = 4
    if family == "impala":
        st.max_samples = 4 * 4 * 64
    model = main.build(st)
    worker = model.start({"worker": []})
    net = worker.global_net
    assert worker.global_step >= 4 and net.global_step == worker.global_step
    assert all(np.isfinite(v.detach().cpu().numpy()).all() for v in net.variables().values())
    assert all(np.isfinite(g.detach().cpu().numpy()).all() for g in net.gradients().values())


def test_batched_rollout_feeds_learner(dev):
    """E environments in host processes, one batched policy call per step, PPO buffers, one step."""
    import numpy as np
    import torch
    from pfpn_amd.env_pool import BatchedPPORollout, EnvPool
    from pfpn_amd.envs import SyntheticDeepMimicEnv
    from pfpn_amd.models import SyncReplicasApplier
    from test_gpu_dppo_step import make_net
    net = make_net(dev, seed=3)
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    pool = EnvPool([(lambda i=i: SyntheticDeepMimicEnv(episode_length=50, seed=i)) for i in range(8)], processes=2)
    try:
        ro = BatchedPPORollout(net, pool, unroll_length=64)
        buf = ro.collect()
        assert buf["state"].shape == (512, 197) and torch.isfinite(buf["advantage"]).all()
        # log_prob stored at sampling time == log-prob of that action under the (unchanged) policy
        policy, _, _ = net._forward(buf["state"])
        lp = policy.log_prob(buf["action"])
        assert torch.allclose(lp, buf["log_prob"], rtol=1e-4, atol=1e-3)
        assert net.sum_active.sum().item() > 0                      # rollout updated the activation statistics
        loss, _ = net.train(None, opt, [], buf["state"], buf["action"], buf["value"], buf["log_prob"], buf["advantage"])
        assert np.isfinite(loss[0]) and net.global_step == 1
    finally:
        pool.close()


def test_graphed_rollout_matches_eager(dev):
    """One hipGraph replay per environment step gives what run_batch gives on the same states and draws, keeps
    the activation statistics going, and survives the learner using other batch sizes in between."""
    import numpy as np
    import torch
    from pfpn_amd.graphs import GraphedRollout
    from pfpn_amd.models import SyncReplicasApplier
    from test_gpu_dppo_step import make_net, make_batch
    net = make_net(dev, seed=4)
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    E = 16
    gr = GraphedRollout(net, E)
    rng = np.random.default_rng(0)
    for it in range(3):
        st = rng.normal(0, 1, (E, 197)).astype(np.float32)
        before = net.sum_active.clone()
        a, lp, v = [t.clone() for t in gr(st)]
        noise = {k: t.clone() for k, t in gr.noise.items()}
        after_graph = net.sum_active.clone()
        net.sum_active.copy_(before)
        a2, lp2, v2 = net.run_batch(st, noise=noise)
        assert torch.equal(a, a2) and torch.equal(lp, lp2) and torch.equal(v, v2)
        assert torch.equal(after_graph, net.sum_active) and not torch.equal(before, after_graph)
        # a learner step with another batch size in between must not disturb the captured buffers
        b = make_batch(rng, 64)
        b["log_prob"] = np.full(64, -40.0, np.float32)
        net.train(None, opt, [], b["state"], b["action"], b["value"], b["log_prob"], b["advantage"])
        net.run(None, st[0])


@pytest.mark.parametrize("n,rounds", [(200, 2), (256, 3)])
def test_staged_training_round_matches_stepwise_training(dev, n, rounds):
    """`train_round` (rollout staged once, device-side minibatch gathers into the captured step, one loss read) gives
    the bits of per-minibatch `train()` calls on the same plan, across the switch from eager steps to hipGraph replays,
    a ragged last minibatch (n = 200), resample ticks, and -- with uniform minibatches (n = 256) -- replays of EIGHT
    steps per graph whose gathers run inside the graph, falling back to one-step replays around a tick."""
    import numpy as np
    import torch
    from pfpn_amd.models import SyncReplicasApplier
    from pfpn_amd.models.distributed_model import minibatch_plan
    from test_gpu_dppo_step import make_net, make_batch
    names = ["state", "action", "value", "log_prob", "advantage"]
    rng = np.random.default_rng(5)
    nets = [make_net(dev, seed=11, resample_interval=13) for _ in range(2)]
    opts = [SyncReplicasApplier(n_, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1) for n_ in nets]
    per_round = 2 * ((n + 31) // 32)                        # two epochs of 32-row minibatches (n = 200: 6 + one of 8)
    for rnd in range(rounds):
        b = make_batch(rng, n)
        b["log_prob"] = rng.normal(-38, 0.3, n).astype(np.float32)
        plan = minibatch_plan(n, 32, 2, np.random.RandomState(rnd))
        cols = {k: b[k] for k in names}
        out_round = nets[0].train_round(opts[0], ["global_step"], names, cols, plan)
        out_steps = [nets[1].train(None, opts[1], ["global_step"], *[cols[k][rows] for k in names]) for rows in plan]
        assert len(out_round) == len(plan) == per_round
        for (l0, e0), (l1, e1) in zip(out_round, out_steps):
            assert l0 == l1 and e0 == e1
        assert torch.equal(nets[0].store.flat, nets[1].store.flat) and torch.equal(nets[0].store.v, nets[1].store.v)
        assert torch.equal(nets[0].state_normalizer.mean, nets[1].state_normalizer.mean)
    assert nets[0].global_step == rounds * per_round
    assert nets[0].train_flag == nets[1].train_flag == rounds * per_round % 13          # steps since the last tick (every 13)
    assert len(nets[0]._train_graphs) >= 1                  # the round really ran on captured steps
    if n == 256:
        (rg,) = nets[0]._round_graphs.values()              # rounds 2 and 3 replayed eight steps at a time
        assert rg.K == 8 and rg.replays >= 2


def test_gather_columns_kernel(dev):
    import torch
    from pfpn_amd import kernels
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    n, B = 777, 100
    srcs = [torch.randn(n, 197, device=dev, generator=g), torch.randn(n, 36, device=dev, generator=g),
            torch.randn(n, device=dev, generator=g), torch.randn(n, device=dev, generator=g)]
    rows = torch.randint(0, n, (B,), device=dev, generator=g)
    dsts = [torch.empty((B,) + tuple(s.shape[1:]), device=dev) for s in srcs]
    kernels.gather_columns(srcs, dsts, rows)
    for s, d in zip(srcs, dsts):
        assert torch.equal(d, s[rows])


def test_overlapped_rollout_on_the_gpu_matches_the_sequential_one(dev):
    """EnvPool halves + two hipGraph policy replays per time step (each half's environments step on the host while the
    GPU runs the other half) against the sequential pool with the same seeds: the step's noise is drawn once for all E
    environments in both, so the drawn particles agree and the buffers agree to the rounding of a GEMM over E/2 instead
    of E rows."""
    import numpy as np
    import torch
    from pfpn_amd.env_pool import BatchedPPORollout, EnvPool
    from pfpn_amd.envs import SyntheticDeepMimicEnv
    from test_gpu_dppo_step import make_net
    E, T = 8, 24
    fns = [(lambda i=i: SyntheticDeepMimicEnv(episode_length=17, seed=50 + i)) for i in range(E)]
    bufs, ks = [], []
    for overlap in (False, True):
        pool = EnvPool(fns, processes=4)
        try:
            net = make_net(dev, seed=21)
            ro = BatchedPPORollout(net, pool, T, overlap=overlap)
            bufs.append(ro.collect())
            ks.append((net.max_active.clone(), net.sum_active.clone()))
        finally:
            pool.close()
    a, b = bufs
    for k in ("state", "action", "value", "log_prob", "advantage"):
        assert a[k].shape == b[k].shape
        torch.testing.assert_close(a[k], b[k], rtol=2e-4, atol=2e-4, msg=k)
    # the activation statistics saw the same rows (in two halves per step): same maxima, sums to summation order
    torch.testing.assert_close(ks[0][0], ks[1][0], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(ks[0][1], ks[1][1], rtol=1e-4, atol=1e-6)
