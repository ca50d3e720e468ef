"""GPU parity of the SAC-PFPN learner step (config 3 of BASELINE.json) vs the NumPy oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F = np.float32
from _parity import assert_close_mostly  # noqa: E402


NET_OPTS = dict(common_net_shape=[], critic_net_shape=[1024, 512], actor_net_shape=[1024, 512],
                value_loss_coef=0.5, gamma=0.95, normalize_state=True, clip_state=5.0,
                normalize_value=False, clip_value=None, normalize_advantage=False, clip_advantage=None,
                particles=35, resample=-1, resample_interval=12000)


def make_net(dev, trainable=True, S=197, A=36, seed=7):
    from pfpn_amd.networks import ParticleFilteringSACNetwork
    net = ParticleFilteringSACNetwork(trainable=trainable, state_shape=[S], action_shape=[A],
                                      action_lower_bound=-np.ones(A), action_upper_bound=np.ones(A),
                                      device=dev, seed=seed, **NET_OPTS)
    net.init()
    return net


@pytest.mark.parametrize("B,one_buffer", [(16, True), (16, False), (256, True), (1024, True), (1024, False),
                                          (40, True),      # rows not on 16-row chunks: one actor pass, two critic passes
                                          (6, True)])      # rows not a multiple of 4: the undeferred step, one pass per input
def test_sac_step_matches_oracle(oracle, dev, B, one_buffer):
    """B = 1024 is BASELINE.json config 3 (replay batch).  `one_buffer`: the four noise tensors as the step draws them
    (one [4, B, A*N] buffer: the actor runs ONCE over [s; s']); else four separate tensors (one actor pass per state)."""
    from pfpn_amd.models import SyncReplicasApplier
    S, A, N = 197, 36, 35
    net = make_net(dev)
    for op in net.init_ops:
        op()
    opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                    {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1)
    st = oracle.SACState(np.random.default_rng(0), S, A, N)
    for k, v in net.variables().items():
        st.P[k] = v.detach().cpu().numpy().reshape(st.P[k].shape).copy()
    for k, v in net.target_variables().items():
        st.PT[k] = v.detach().cpu().numpy().reshape(st.PT[k].shape).copy()
    # tanh-mode particle grid of the network == oracle grid
    o_loc, o_ls = oracle.init_particles(A, N, tanh_mode=True)
    np.testing.assert_array_equal(st.P["actor/samples"], o_loc)
    np.testing.assert_array_equal(st.P["actor/samples_std"], o_ls)
    rng = np.random.default_rng(B)
    for it in range(2):
        batch = {"state": (rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F),
                 "action": np.tanh(rng.normal(0, 1, (B, A))).astype(F),
                 "reward": rng.random(B).astype(F),
                 "not_terminal": (rng.random(B) < 0.9).astype(F),
                 "state_": (rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F)}
        noise = {k: (-np.log(-np.log(rng.uniform(1e-10, 1, (B, A, N)))) if k.startswith("g")
                     else rng.normal(0, 1, (B, A, N))).astype(F) for k in ("gumbel", "eps", "gumbel_", "eps_")}
        (l_ref, pl_ref, vl_ref), G_ref, aux = oracle.sac_grads(st.P, st.PT, st.norm_mean, st.norm_std, batch, noise,
                                                               A, N)
        oracle.sac_learner_step(st, batch, noise)
        dn = {k: torch.from_numpy(v.reshape(B, A * N)).to(dev) for k, v in noise.items()}
        if one_buffer:
            buf = torch.stack([dn[k] for k in ("gumbel", "gumbel_", "eps", "eps_")])
            dn = {"gumbel": buf[0], "gumbel_": buf[1], "eps": buf[2], "eps_": buf[3], "_buf": buf}
        args = [net._to_dev(batch[k]) for k in ("state", "action", "reward", "not_terminal", "state_")]
        loss3 = net.compute_gradients(*args, noise=dn).cpu().numpy()
        assert (net.actor_trunk._bufs.get((B, 3)) is not None) == (one_buffer and B % 4 == 0)        # which path ran
        assert ((2 * B, 0) in net.q_stack._bufs) == (B % 16 == 0)
        assert np.array_equal(net.last["k"].cpu().numpy(), aux["k"]), "rsample particle indices must be bit-exact"
        np.testing.assert_allclose(net.last["logp"].cpu().numpy(), aux["logp"], rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(loss3, [l_ref, pl_ref, vl_ref], rtol=2e-4, atol=2e-5)
        G = net.gradients()
        for name, g_ref in G_ref.items():
            g_ref = np.asarray(g_ref, F)
            got = G[name].cpu().numpy().reshape(g_ref.shape)
            sc = np.abs(g_ref).max() + 1e-30
            assert_close_mostly(got, g_ref, rtol=3e-3, atol=3e-4 * sc, err_msg=name)
        opt.apply(net)
        for name, v in net.variables().items():
            ref = st.P[name]
            assert_close_mostly(v.cpu().numpy().reshape(ref.shape), ref, rtol=1e-4, atol=3e-6, err_msg=name)
        for name, v in net.target_variables().items():
            ref = st.PT[name]
            assert_close_mostly(v.cpu().numpy().reshape(ref.shape), ref, rtol=1e-5, atol=2e-7, err_msg="target " + name)
        np.testing.assert_allclose(net.state_normalizer.mean.cpu().numpy(), st.norm_mean, rtol=1e-5, atol=1e-6)
    assert net.global_step == 2 and net.train_flag == 2


def test_sac_api_surface(dev):
    """run() -> [action] in (-1, 1); train() with host arrays returns the reference's loss tuple."""
    from pfpn_amd.models import SyncReplicasApplier
    net = make_net(dev)
    opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                    {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1)
    rng = np.random.default_rng(0)
    out = net.run(None, rng.normal(0, 1, 197).astype(F))
    assert len(out) == 1 and out[0].shape == (36,) and np.all(np.abs(out[0]) < 1)
    B = 64
    loss, extra = net.train(None, opt, ["global_step"], rng.normal(0, 1, (B, 197)).astype(F),
                            np.tanh(rng.normal(0, 1, (B, 36))).astype(F), rng.random(B).astype(F),
                            np.ones(B, F), rng.normal(0, 1, (B, 197)).astype(F))
    assert extra == [1] and loss[1] is None and all(np.isfinite([loss[0], loss[2], loss[3]]))
    # deterministic evaluator: tanh(loc[argmax])
    ev = make_net(dev, trainable=False)
    a1 = ev.run(None, np.zeros(197, F))[0]
    a2 = ev.run(None, np.zeros(197, F))[0]
    assert np.array_equal(a1, a2) and np.all(np.abs(a1) < 1)


def test_sac_step_as_graph_matches_eager(dev):
    """The captured SAC learner step (draws refilled outside the graph) == the eager step on the same draws."""
    from pfpn_amd.graphs import GraphedLearnerStep
    from pfpn_amd.models import SyncReplicasApplier
    rng = np.random.default_rng(3)
    B = 64
    nets, opts = [], []
    for _ in range(2):
        net = make_net(dev)
        nets.append(net)
        opts.append(SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                              {"names": net.actor_param_names(), "lr": 1e-4}],
                                        norm_clip=1.0, world_size=1))
    assert torch.equal(nets[0].store.flat, nets[1].store.flat)
    batch = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(F),
                                                   rng.uniform(-0.9, 0.9, (B, 36)).astype(F),
                                                   rng.normal(0.5, 0.3, B).astype(F),
                                                   (rng.uniform(size=B) > 0.05).astype(F),
                                                   (rng.normal(0, 1, (B, 197)) * 2).astype(F))]
    g = GraphedLearnerStep(nets[0], opts[0], *[t.clone() for t in batch], noise=nets[0].make_train_noise(B))
    assert g.noise_in_graph                  # every replay's optimiser launch draws the next step's noise (no launch between)
    for it in range(3):
        g.ensure_noise()                      # (a no-op after the first replay: the previous optimiser launch drew it)
        noise = {k: t.clone() for k, t in g.noise.items()}
        l_g = g().clone()
        l_e = nets[1].compute_gradients(*batch, noise).clone()
        opts[1].apply(nets[1])
        assert torch.equal(l_g, l_e), it
        assert torch.equal(nets[0].store.flat, nets[1].store.flat) and torch.equal(nets[0].store.v, nets[1].store.v)
        assert torch.equal(nets[0].target_store.flat, nets[1].target_store.flat)
        assert nets[0].global_step == nets[1].global_step == it + 1


def test_sac_multi_step_replay_draws_its_noise_inside_the_graph(dev):
    """K SAC learner steps in ONE hipGraph: step k's optimiser launch carries step k+1's prologue (s and s' normalised) AND
    draws step k+1's four noise tensors (Philox keyed by the step's device-resident draw counter, which the loss head
    advances), so nothing is launched between the steps or between replays.  Equal, bit for bit, to K eager steps that draw
    their noise with the same generator before each step."""
    from pfpn_amd.graphs import GraphedLearnerStep
    from pfpn_amd.models import SyncReplicasApplier
    rng = np.random.default_rng(5)
    B, K = 64, 3
    nets, opts = [], []
    for _ in range(2):
        net = make_net(dev)
        nets.append(net)
        opts.append(SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                              {"names": net.actor_param_names(), "lr": 1e-4}],
                                        norm_clip=1.0, world_size=1))
    sets = [[torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(F),
                                                   rng.uniform(-0.9, 0.9, (B, 36)).astype(F),
                                                   rng.normal(0.5, 0.3, B).astype(F),
                                                   (rng.uniform(size=B) > 0.05).astype(F),
                                                   (rng.normal(0, 1, (B, 197)) * 2).astype(F))] for _ in range(K)]
    g = GraphedLearnerStep(nets[0], opts[0], steps_per_replay=K, input_sets=[[t.clone() for t in s_] for s_ in sets],
                           noise=nets[0].make_train_noise(B), debug_nodes=True)
    assert g.K == K and g.noise_in_graph
    one = GraphedLearnerStep(nets[1], opts[1], *[t.clone() for t in sets[0]], noise=nets[1].make_train_noise(B),
                             debug_nodes=True)
    nk, n1 = g.kernel_nodes(), one.kernel_nodes()
    if nk and n1:
        assert nk <= K * n1 - (K - 1), (nk, n1)             # the K - 1 prologues inside the replay launch nothing
    for rep in range(2):
        g()
        for s_ in sets:
            noise = nets[1].make_train_noise(B)
            nets[1].compute_gradients(*s_, noise)
            opts[1].apply(nets[1])
        torch.cuda.synchronize()
        assert torch.equal(nets[0].store.flat, nets[1].store.flat), rep
        assert torch.equal(nets[0].target_store.flat, nets[1].target_store.flat)
        assert torch.equal(nets[0].state_normalizer.std, nets[1].state_normalizer.std)
        assert torch.equal(nets[0].noise_epoch, nets[1].noise_epoch) and int(nets[0].noise_epoch.item()) == K * (rep + 1)
        assert nets[0].global_step == nets[1].global_step == K * (rep + 1)


def test_sac_multi_step_capture_without_the_noise_rider_is_refused(dev, tmp_path):
    """Round 5: the draw for step k + 1 rides in step k's optimiser launch ONLY on the fused one-replica path with
    the norm hook.  Where the optimiser drops the rider (here: `fused_norm=False`, no hook) a K-step replay would run K steps
    on ONE set of Gumbel / normal draws: the stepper counts the takers and refuses; a one-step stepper (noise refilled between
    replays) is fine.  Also: `noise_epoch` exists from construction and a restored checkpoint sets it to the restored step."""
    from pfpn_amd import checkpoint as ck
    from pfpn_amd.graphs import GraphedLearnerStep
    from pfpn_amd.models import SyncReplicasApplier
    rng = np.random.default_rng(6)
    B, K = 64, 2
    net = make_net(dev)
    assert int(net.noise_epoch.item()) == 0
    opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                    {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1,
                              fused_norm=False)
    sets = [[torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(F),
                                                   rng.uniform(-0.9, 0.9, (B, 36)).astype(F),
                                                   rng.normal(0.5, 0.3, B).astype(F),
                                                   (rng.uniform(size=B) > 0.05).astype(F),
                                                   (rng.normal(0, 1, (B, 197)) * 2).astype(F))] for _ in range(K)]
    flat0 = net.store.flat.clone()
    with pytest.raises(RuntimeError, match="drew its noise inside the graph in 0 of its steps"):
        GraphedLearnerStep(net, opt, steps_per_replay=K, input_sets=sets, noise=net.make_train_noise(B))
    assert torch.equal(net.store.flat, flat0) and net.global_step == 0       # the refused capture left the state alone
    one = GraphedLearnerStep(net, opt, *sets[0], noise=net.make_train_noise(B))
    assert not one.noise_in_graph
    one(); one()
    assert net.global_step == 2 and int(net.noise_epoch.item()) == 2
    prefix = str(tmp_path / "model.ckpt-2")
    ck.save(prefix, net, opt)
    net2 = make_net(dev)
    opt2 = SyncReplicasApplier(net2, [{"names": net2.critic_param_names(), "lr": 1e-4},
                                      {"names": net2.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1)
    net2.make_train_noise(B)                                  # a draw BEFORE the restore must not pin the counter at 0
    ck.restore(prefix, net2, opt2)
    assert net2.global_step == 2 and int(net2.noise_epoch.item()) == 2


@pytest.mark.parametrize("B", [64, 1024])
def test_sac_deferred_reduce_step_matches_immediate_step(dev, monkeypatch, B):
    """SAC step with both Q heads folded into the trunk epilogues and ONE deferred reduction launch == the step that
    launches every head / bias-gradient reduction / EMA on its own (same draws): same bits."""
    from pfpn_amd.models import SyncReplicasApplier
    rng = np.random.default_rng(B)
    nets, opts = [], []
    for i in range(2):
        net = make_net(dev)
        # compared bit for bit: the same grouping of the norm's partial sums (fused_norm=False) and the same GEMM row counts
        # (the deferred step would run the actor and the live critic once over 2B rows: own test below)
        net.batch_actor = net.batch_critic = False
        net.defer_reduce = i == 0
        nets.append(net)
        opts.append(SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                              {"names": net.actor_param_names(), "lr": 1e-4}],
                                        norm_clip=1.0, world_size=1, fused_norm=False))
    for it in range(3):
        batch = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(F),
                                                       rng.uniform(-0.9, 0.9, (B, 36)).astype(F),
                                                       rng.normal(0.5, 0.3, B).astype(F),
                                                       (rng.uniform(size=B) > 0.05).astype(F),
                                                       (rng.normal(0, 1, (B, 197)) * 2).astype(F))]
        noise = nets[0].make_train_noise(B)
        losses = []
        for i, (net, opt) in enumerate(zip(nets, opts)):
            losses.append(net.compute_gradients(*batch, {k: t.clone() for k, t in noise.items()}).clone())
            assert net._defer_ok[B] is (i == 0)
            g = net.store.grad.clone()
            opt.apply(net)
            losses.append(g)
        assert torch.equal(losses[0], losses[2]), it
        assert torch.equal(losses[1], losses[3]), it
        assert torch.equal(nets[0].store.flat, nets[1].store.flat)
        assert torch.equal(nets[0].last["q_pi"], nets[1].last["q_pi"]) and torch.equal(nets[0].q_tgt, nets[1].q_tgt)
        assert torch.equal(nets[0].state_normalizer.std, nets[1].state_normalizer.std)


@pytest.mark.parametrize("B", [64, 1024])
def test_sac_one_pass_over_both_batches_matches_two_passes(dev, monkeypatch, B):
    """The step that runs the actor once over [s; s'] and the live critic once over [(s, a_pi); (s, action_hist)] (2B-row
    passes; the critic's parameter gradients from the second half of the rows only) against the step with one pass per
    input: same particle indices, log-probs / Q values / losses / every gradient / parameters to fp32 GEMM rounding (a
    2B-row GEMM may sum in another order), identical normaliser statistics."""
    from pfpn_amd.models import SyncReplicasApplier
    from _parity import assert_close_mostly
    rng = np.random.default_rng(B + 7)
    nets, opts = [], []
    for i in range(2):
        net = make_net(dev)
        net.batch_actor = net.batch_critic = i == 0
        nets.append(net)
        opts.append(SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                              {"names": net.actor_param_names(), "lr": 1e-4}],
                                        norm_clip=1.0, world_size=1))
    for it in range(2):
        batch = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(F),
                                                       rng.uniform(-0.9, 0.9, (B, 36)).astype(F),
                                                       rng.normal(0.5, 0.3, B).astype(F),
                                                       (rng.uniform(size=B) > 0.05).astype(F),
                                                       (rng.normal(0, 1, (B, 197)) * 2).astype(F))]
        noise = nets[0].make_train_noise(B)
        l3 = [net.compute_gradients(*batch, {k: t.clone() for k, t in noise.items()}).clone() for net in nets]
        assert (nets[0].actor_trunk._bufs.get((B, 3)) is not None) and nets[1].actor_trunk._bufs.get((B, 3)) is None
        assert (2 * B, 0) in nets[0].q_stack._bufs and (2 * B, 0) not in nets[1].q_stack._bufs
        assert torch.equal(nets[0].last["k"], nets[1].last["k"])
        torch.testing.assert_close(nets[0].last["logp"], nets[1].last["logp"], rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(nets[0].last["q_pi"], nets[1].last["q_pi"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(l3[0], l3[1], rtol=1e-5, atol=1e-5)
        g0, g1 = nets[0].gradients(), nets[1].gradients()
        for name in g1:
            a, b = g0[name].cpu().numpy(), g1[name].cpu().numpy()
            assert_close_mostly(a, b, rtol=1e-3, atol=1e-5 * (np.abs(b).max() + 1e-30), err_msg=name)
        for net, opt in zip(nets, opts):
            opt.apply(net)
        torch.testing.assert_close(nets[0].store.flat, nets[1].store.flat, rtol=1e-4, atol=2e-6)
        assert torch.equal(nets[0].state_normalizer.std, nets[1].state_normalizer.std)


@pytest.mark.parametrize("B", [64, 1024])
def test_sac_loss_head_rides_in_the_critics_backward_launch(dev, B):
    """`net.ride_loss`: the loss head has no launch of its own -- the first launch of the critics' backward pass forms the
    dQ of its rows itself.  Against the step with the stand-alone loss head: dQ, dlogp and every gradient that descends from
    them bit for bit; the three losses and d log_alpha (block sums over 256 instead of 1024 threads) to rounding; the draw
    counter advanced alike."""
    from pfpn_amd.models import SyncReplicasApplier
    rng = np.random.default_rng(B + 3)
    nets = []
    for i in range(2):
        net = make_net(dev)
        net.ride_loss = i == 0
        nets.append(net)
    for it in range(2):
        batch = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(F),
                                                       rng.uniform(-0.9, 0.9, (B, 36)).astype(F),
                                                       rng.normal(0.5, 0.3, B).astype(F),
                                                       (rng.uniform(size=B) > 0.05).astype(F),
                                                       (rng.normal(0, 1, (B, 197)) * 2).astype(F))]
        noise = nets[0].make_train_noise(B)
        l3 = [net.compute_gradients(*batch, {k: t.clone() for k, t in noise.items()}).clone() for net in nets]
        assert torch.equal(nets[0].dq_both, nets[1].dq_both)
        assert torch.equal(nets[0].sac_out["dlogp"], nets[1].sac_out["dlogp"])
        torch.testing.assert_close(l3[0], l3[1], rtol=2e-6, atol=1e-6)
        assert int(nets[0].noise_epoch) == int(nets[1].noise_epoch) == it + 1
        g0, g1 = nets[0].gradients(), nets[1].gradients()
        for name in g1:
            if "log_alpha" in name:
                torch.testing.assert_close(g0[name], g1[name], rtol=2e-6, atol=1e-7)
            else:
                assert torch.equal(g0[name], g1[name]), name


@pytest.mark.parametrize("B", [64, 1024])
def test_sac_live_and_target_critics_share_their_epilogue_launches(dev, B):
    """`net.pair_critics`: the live critics' pass over [(s, a_pi); (s, a_hist)] and the target critics' pass over (s', a')
    run layer by layer with ONE epilogue launch for both -- same Q values, losses, gradients and parameters, bit for bit."""
    from pfpn_amd.models import SyncReplicasApplier
    rng = np.random.default_rng(B + 11)
    nets, opts = [], []
    for i in range(2):
        net = make_net(dev)
        net.pair_critics = i == 0
        nets.append(net)
        opts.append(SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                              {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1))
    for it in range(3):
        batch = [torch.from_numpy(a).to(dev) for a in ((rng.normal(0, 1, (B, 197)) * 2).astype(F),
                                                       rng.uniform(-0.9, 0.9, (B, 36)).astype(F),
                                                       rng.normal(0.5, 0.3, B).astype(F),
                                                       (rng.uniform(size=B) > 0.05).astype(F),
                                                       (rng.normal(0, 1, (B, 197)) * 2).astype(F))]
        noise = nets[0].make_train_noise(B)
        l3 = [net.compute_gradients(*batch, {k: t.clone() for k, t in noise.items()}).clone() for net in nets]
        assert torch.equal(nets[0].q_both, nets[1].q_both) and torch.equal(nets[0].q_tgt, nets[1].q_tgt)
        assert torch.equal(l3[0], l3[1])
        assert torch.equal(nets[0].store.grad, nets[1].store.grad)
        for net, opt in zip(nets, opts):
            opt.apply(net)
        assert torch.equal(nets[0].store.flat, nets[1].store.flat)
        assert torch.equal(nets[0].target_store.flat, nets[1].target_store.flat)


def test_gumbel_transform_matches_tensor_ops(dev):
    from pfpn_amd import kernels
    u = torch.rand(1000, 1260, device=dev)
    u[0, 0] = 0.0                                      # clamped to FLT_MIN like TFP's minval=tiny
    ref = -torch.log(-torch.log(u.clamp(min=float(np.finfo(np.float32).tiny))))
    g = kernels.gumbel_from_uniform_(u.clone())
    assert torch.isfinite(g).all()
    torch.testing.assert_close(g, ref, rtol=2e-6, atol=2e-6)


def test_sac_noise_transform_distributions(dev):
    """rows 0,1 Gumbel(0,1) (mean = Euler's constant, var = pi^2/6); rows 2,3 independent N(0,1) (Box-Muller pair)."""
    from pfpn_amd import kernels
    g = torch.Generator(device=dev); g.manual_seed(5)
    u = torch.rand(4, 2048, 1260, device=dev, generator=g)
    u[0, 0, 0] = 0.0; u[2, 0, 0] = 0.0
    ref_g = -torch.log(-torch.log(u[1].clamp(min=float(np.finfo(np.float32).tiny))))
    z = kernels.sac_noise_from_uniform_(u)
    assert torch.isfinite(z).all()
    torch.testing.assert_close(z[1], ref_g, rtol=2e-6, atol=2e-6)
    for r in (0, 1):
        assert abs(z[r].mean().item() - 0.5772157) < 5e-3 and abs(z[r].var().item() - np.pi ** 2 / 6) < 2e-2
    for r in (2, 3):
        assert abs(z[r].mean().item()) < 3e-3 and abs(z[r].var().item() - 1.0) < 5e-3
        assert abs((z[r] ** 4).mean().item() - 3.0) < 5e-2                      # kurtosis of a normal
    assert abs((z[2] * z[3]).mean().item()) < 3e-3                                # the pair is uncorrelated


def test_sac_noise_fill_philox(dev):
    """pfpn_sac_noise_fill: one launch draws the step's four noise tensors from Philox4x32-10 blocks.  Known answer of the
    generator (Random123's kat vector: counter 0, key 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8), distributions, streams."""
    from pfpn_amd import kernels
    n = 2048 * 1260
    z = kernels.sac_noise_fill_(torch.empty(4, n, device=dev), 0, 0)
    assert torch.isfinite(z).all()
    r = [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    u = [np.float32((x >> 8) * 2.0 ** -24) for x in r]
    first = z[:, 0].cpu().numpy()
    np.testing.assert_allclose(first[0], -np.log(-np.log(u[0])), rtol=1e-5)
    np.testing.assert_allclose(first[1], -np.log(-np.log(u[1])), rtol=1e-5)
    rr = np.sqrt(-2.0 * np.log(np.float64(u[2])))
    np.testing.assert_allclose(first[2], rr * np.cos(2 * np.pi * np.float64(u[3])), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(first[3], rr * np.sin(2 * np.pi * np.float64(u[3])), rtol=2e-5, atol=1e-6)
    for row in (0, 1):
        assert abs(z[row].mean().item() - 0.5772157) < 5e-3 and abs(z[row].var().item() - np.pi ** 2 / 6) < 2e-2
    for row in (2, 3):
        assert abs(z[row].mean().item()) < 3e-3 and abs(z[row].var().item() - 1.0) < 5e-3
        assert abs((z[row] ** 4).mean().item() - 3.0) < 5e-2
    assert abs((z[2] * z[3]).mean().item()) < 3e-3 and abs((z[0] * z[1]).mean().item() - 0.5772157 ** 2) < 5e-3
    # same (seed, offset) -> same draws; another offset or seed -> another stream
    assert torch.equal(z, kernels.sac_noise_fill_(torch.empty(4, n, device=dev), 0, 0))
    for seed, off in ((0, 1), (1, 0), (1 << 40, 0), (0, 1 << 33)):
        z2 = kernels.sac_noise_fill_(torch.empty(4, n, device=dev), seed, off)
        assert not torch.equal(z2, z) and abs((z2[2] * z[2]).mean().item()) < 3e-3
    # the offset read from device memory when the launch runs (the learner step's draw counter) == the same offset by value
    ep = torch.tensor([1 << 33], dtype=torch.int64, device=dev)
    assert torch.equal(kernels.sac_noise_fill_(torch.empty(4, n, device=dev), 0, 12345, epoch=ep),
                       kernels.sac_noise_fill_(torch.empty(4, n, device=dev), 0, 1 << 33))
