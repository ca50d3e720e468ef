"""GPU parity of the TD3-PFPN learner step (reference networks/actor_critic/ddpg.py) vs the NumPy oracle:
deterministic particle action with its custom gradient, twin critics, the full target network with its own actor and
normaliser statistics, the policy-delay gate and the delayed soft update."""
import numpy as np
import pytest
import torch

from _parity import close_scaled

pytestmark = pytest.mark.gpu
F = np.float32

NET_OPTS = dict(common_net_shape=[], critic_net_shape=[1024, 512], actor_net_shape=[1024, 512],
                value_loss_coef=0.5, gamma=0.95, normalize_state=True, clip_state=5.0,
                normalize_value=False, clip_value=None, normalize_advantage=False, clip_advantage=None,
                particles=35, resample=-1, resample_interval=12000)


def make_net(dev, trainable=True, S=197, A=36, seed=7, **kw):
    from pfpn_amd.networks import ParticleFilteringDDPGNetwork
    net = ParticleFilteringDDPGNetwork(trainable=trainable, state_shape=[S], action_shape=[A],
                                       action_lower_bound=-np.ones(A), action_upper_bound=np.ones(A),
                                       device=dev, seed=seed, **dict(NET_OPTS, **kw))
    net.init()
    return net


@pytest.mark.parametrize("B", [16, 256, 1024])         # 1024 = the replay batch of BASELINE.json config 3
def test_td3_step_matches_oracle(oracle, dev, B):
    from pfpn_amd.models import SyncReplicasApplier
    S, A, N = 197, 36, 35
    net = make_net(dev)
    for op in net.init_ops:
        op()
    opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                    {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1)
    st = oracle.TD3State(np.random.default_rng(0), S, A, N)
    for k, v in net.variables().items():
        st.P[k] = v.detach().cpu().numpy().reshape(st.P[k].shape).copy()
    st.PT = {k: v.copy() for k, v in st.P.items()}
    # the fixed-width tanh grid (act_noise -> init_sigma, fixed_sigma) of the network == the oracle's
    o_loc, o_ls = oracle.init_particles(A, N, tanh_mode=True, init_sigma=0.1)
    np.testing.assert_array_equal(st.P["actor/samples"], o_loc)
    np.testing.assert_array_equal(st.P["actor/samples_std"], o_ls)
    rng = np.random.default_rng(B)
    for it in range(4):                                   # steps 0 and 2 carry the policy gradient and the soft update
        batch = {"state": (rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F),
                 "action": np.tanh(rng.normal(0, 1, (B, A))).astype(F),
                 "reward": rng.random(B).astype(F),
                 "not_terminal": (rng.random(B) < 0.9).astype(F),
                 "state_": (rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F)}
        noise_t = rng.normal(0, 1, (B, A)).astype(F)
        sync = st.global_step % 2 == 0
        # keep every pre-activation that has a backward off the ReLU6 kinks (Relu6Grad is a step function: a value the
        # GPU's GEMM rounds to the other side moves a whole row's contribution, and at 1024 rows x ~6000 units one does)
        for _ in range(20):
            _, _, aux0 = oracle.td3_grads(st.P, st.PT, (st.norm_mean, st.norm_std), (st.norm_mean_t, st.norm_std_t), batch,
                                          noise_t, A, N, sync)
            x = oracle.state_normalize(batch["state"], st.norm_mean, st.norm_std, 5.0)
            xa = np.concatenate([x, aux0["a_opt"]], -1).astype(F)
            xh = np.concatenate([x, batch["action"]], -1).astype(F)
            bad = np.zeros(B, bool)
            for prefix, h in (("actor", x), ("critic/q", xa), ("critic/q", xh), ("critic/twin_q", xh)):
                for layer in ("fc1", "fc2"):
                    pre = h @ st.P["%s/%s/weight" % (prefix, layer)] + st.P["%s/%s/bias" % (prefix, layer)]
                    bad |= (np.minimum(np.abs(pre), np.abs(pre - 6.0)) < 2e-6).any(axis=1)
                    h = np.minimum(np.maximum(pre, 0), 6).astype(F)
            # ... and every argmax over a dimension's particle logits off a tie (fp32 logits from another GEMM order differ
            # by ~1e-7: a second-best within that of the best picks another particle, an O(0.1) jump of the action)
            def tied(Pn, xin):
                lg, _ = oracle.actor_forward(Pn, xin, A, N)
                top = np.sort(lg.reshape(B, A, N), axis=-1)
                return ((top[..., -1] - top[..., -2]) < 2e-6).any(axis=1)
            bad |= tied(st.P, x)
            x2 = oracle.state_normalize(batch["state_"], st.norm_mean_t, st.norm_std_t, 5.0)
            bad2 = tied(st.PT, x2)
            if not bad.any() and not bad2.any():
                break
            batch["state"][bad] = (rng.normal(0, 1, (int(bad.sum()), S)) * 2 + 0.3).astype(F)
            batch["state_"][bad2] = (rng.normal(0, 1, (int(bad2.sum()), S)) * 2 + 0.3).astype(F)
        assert not bad.any() and not bad2.any()
        (l_ref, pl_ref, vl_ref), G_ref, aux = oracle.td3_grads(
            st.P, st.PT, (st.norm_mean, st.norm_std), (st.norm_mean_t, st.norm_std_t), batch, noise_t, A, N, sync)
        oracle.td3_learner_step(st, batch, noise_t)
        args = [net._to_dev(batch[k]) for k in ("state", "action", "reward", "not_terminal", "state_")]
        loss3 = net.compute_gradients(*args, noise={"target": torch.from_numpy(noise_t).to(dev)}).cpu().numpy()
        assert net._sync_now == sync
        assert np.array_equal(net.last["k"].cpu().numpy(), aux["k"]), "argmax particle indices must be bit-exact"
        close_scaled(net.last["a_opt"].cpu().numpy(), aux["a_opt"], 1e-6, "deterministic action")
        close_scaled(net.last["a2"].cpu().numpy(), aux["a2"], 1e-5, "smoothed target action")
        close_scaled(loss3, [l_ref, pl_ref, vl_ref], 2e-5, "losses", floor=1.0)
        G = net.gradients()
        for name, g_ref in G_ref.items():
            g_ref = np.asarray(g_ref, F)
            close_scaled(G[name].cpu().numpy(), g_ref, 2e-4, "grad " + name, floor=1e-30,
                         flips=max(8, g_ref.size // 500))          # ReLU6 mask bits that round the other way
            if not sync and name.startswith("actor"):
                assert not G[name].any()
        opt.apply(net)
        for name, v in net.variables().items():
            close_scaled(v.cpu().numpy(), st.P[name], 3e-6, name, floor=1.0, flips=max(8, st.P[name].size // 500))
        for name, v in net.target_variables().items():
            close_scaled(v.cpu().numpy(), st.PT[name], 3e-7, "target " + name, floor=1.0, flips=max(8, v.numel() // 500))
        close_scaled(net.state_normalizer.mean.cpu().numpy(), st.norm_mean, 1e-5, "norm mean", floor=1.0)
        close_scaled(net.target_norm[0].cpu().numpy(), st.norm_mean_t, 1e-5, "target norm mean", floor=1.0)
        close_scaled(net.target_norm[1].cpu().numpy(), st.norm_std_t, 1e-5, "target norm std", floor=1.0)
    assert net.global_step == 4
    assert not torch.equal(net.target_store.flat, net.store.flat[:net.target_store.flat.numel()])


def test_td3_api_surface(dev):
    """run() -> [action] inside the action box, exploring when trainable and deterministic when not; train() with host
    arrays returns the reference's loss tuple; unbuilt options raise."""
    from pfpn_amd.models import SyncReplicasApplier
    net = make_net(dev)
    opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                    {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1)
    rng = np.random.default_rng(0)
    s = rng.normal(0, 1, 197).astype(F)
    a1, a2 = net.run(None, s)[0], net.run(None, s)[0]
    assert a1.shape == (36,) and np.all(np.abs(a1) <= 1) and not np.array_equal(a1, a2)       # exploration noise
    B = 64
    loss, extra = net.train(None, opt, ["global_step"], rng.normal(0, 1, (B, 197)).astype(F),
                            np.tanh(rng.normal(0, 1, (B, 36))).astype(F), rng.random(B).astype(F),
                            np.ones(B, F), rng.normal(0, 1, (B, 197)).astype(F))
    assert extra == [1] and loss[1] is None and all(np.isfinite([loss[0], loss[2], loss[3]]))
    ev = make_net(dev, trainable=False)
    e1, e2 = ev.run(None, s)[0], ev.run(None, s)[0]
    assert np.array_equal(e1, e2) and np.all(np.abs(e1) < 1)
    with pytest.raises(NotImplementedError):
        make_net(dev, ou_noise=True)
    with pytest.raises(NotImplementedError):
        make_net(dev, twin_q=False)


@pytest.mark.parametrize("policy_step", [True, False])
def test_td3_step_as_graph_matches_eager(dev, policy_step):
    """Each of the two kinds of TD3 learner step (the policy is updated / the policy-delay step) captured into a hipGraph
    and replayed == the eager step on the same draws: loss, parameters, Adam slots and target network bit for bit."""
    from pfpn_amd.graphs import GraphedLearnerStep
    from pfpn_amd.models import SyncReplicasApplier
    B = 64
    rng = np.random.default_rng(3)
    nets, opts = [], []
    for _ in range(2):
        net = make_net(dev)
        for op in net.init_ops:
            op()
        net.global_step = 0 if policy_step else 1          # policy_delay 2: even steps update the policy
        nets.append(net)
        opts.append(SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                              {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1))
    assert nets[0]._graph_phase() == (policy_step,)
    batch = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(F),
                                                   np.tanh(rng.normal(0, 1, (B, 36))).astype(F),
                                                   rng.random(B).astype(F), (rng.random(B) < 0.9).astype(F),
                                                   (rng.normal(0, 1, (B, 197)) * 2).astype(F))]
    g = GraphedLearnerStep(nets[0], opts[0], *[t.clone() for t in batch], noise=nets[0].make_train_noise(B))
    for it in range(3):
        nets[0]._sync_now = policy_step                     # what train() sets before a replay
        l_g = g().clone()
        noise = {k: t.clone() for k, t in g.noise.items()}
        l_e = nets[1].compute_gradients(*batch, noise, sync=policy_step).clone()
        opts[1].apply(nets[1])
        assert torch.equal(l_g, l_e), it
        assert torch.equal(nets[0].store.flat, nets[1].store.flat) and torch.equal(nets[0].store.m, nets[1].store.m)
        assert torch.equal(nets[0].target_store.flat, nets[1].target_store.flat)
    assert nets[0].global_step == nets[1].global_step == (3 if policy_step else 4)


def test_td3_train_captures_one_graph_per_kind_of_step(dev):
    """net.train() alternates policy-update and policy-delay steps; after the eager warm-up calls each kind replays its own
    graph, the delayed soft update still fires on policy steps only, losses stay finite."""
    from pfpn_amd.models import SyncReplicasApplier
    net = make_net(dev)
    opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                    {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1)
    rng = np.random.default_rng(1)
    B = 32
    data = (rng.normal(0, 1, (B, 197)).astype(F), np.tanh(rng.normal(0, 1, (B, 36))).astype(F), rng.random(B).astype(F),
            np.ones(B, F), rng.normal(0, 1, (B, 197)).astype(F))
    tgt = []
    for it in range(16):
        before = net.target_store.flat.clone()
        loss, extra = net.train(None, opt, ["global_step"], *data)
        assert extra == [it + 1] and all(np.isfinite([loss[0], loss[2], loss[3]]))
        tgt.append(not torch.equal(before, net.target_store.flat))
    assert len(net._train_graphs) == 2 and {k[2] for k in net._train_graphs} == {True, False}
    assert tgt == [it % 2 == 0 for it in range(16)]         # the target moves on policy steps (even global_step) only


def test_td3_checkpoint_keeps_the_target_normaliser(dev, tmp_path):
    """the target network's OWN normaliser statistics (soft-updated with tau) survive save -> restore: the Q-target of the
    first step after a resume is the one the uninterrupted run computes."""
    from pfpn_amd import checkpoint as ck
    from pfpn_amd.models import SyncReplicasApplier
    S, A = 197, 36
    rng = np.random.default_rng(3)

    def build(seed):
        net = make_net(dev, seed=seed)
        for op in net.init_ops:
            op()
        opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                        {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1)
        return net, opt

    def batch(B=64):
        return [(rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F), np.tanh(rng.normal(0, 1, (B, A))).astype(F),
                rng.random(B).astype(F), (rng.random(B) < 0.9).astype(F), (rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F)]

    net, opt = build(7)
    for _ in range(4):                      # the live statistics move every step, the target's every other step (tau)
        net.train(None, opt, [], *batch())
    assert not torch.equal(net.target_norm[0], net.state_normalizer.mean)
    assert float(net.target_norm[0].abs().max()) > 0
    prefix = str(tmp_path / "td3.ckpt-4")
    entries = ck.save(prefix, net, opt)
    assert "global_net/target_net/state_normalizer/mean" in entries
    net2, opt2 = build(99)
    ck.restore(prefix, net2, opt2)
    assert torch.equal(net2.target_norm[0], net.target_norm[0]) and torch.equal(net2.target_norm[1], net.target_norm[1])
    assert torch.equal(net2.target_store.flat, net.target_store.flat)
    b = batch()
    noise = {"target": torch.randn(64, A, device=dev)}
    l1 = net.compute_gradients(*[net._to_dev(x) for x in b], noise={k: v.clone() for k, v in noise.items()})
    l2 = net2.compute_gradients(*[net2._to_dev(x) for x in b], noise={k: v.clone() for k, v in noise.items()})
    assert torch.equal(l1, l2) and torch.equal(net.store.grad, net2.store.grad)
    # a checkpoint written without the target statistics starts the target normaliser from the live ones
    rd_entries = {k: v for k, v in ck.network_tensors(net, opt).items() if "target_net/state_normalizer" not in k}
    ck.write_checkpoint(str(tmp_path / "old.ckpt-4"), rd_entries)
    net3, opt3 = build(5)
    ck.restore(str(tmp_path / "old.ckpt-4"), net3, opt3)
    assert torch.equal(net3.target_norm[0], net3.state_normalizer.mean)
    assert torch.equal(net3.target_norm[1], net3.state_normalizer.std)
