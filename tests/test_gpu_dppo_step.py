"""GPU parity of the whole DPPO learner step (reference-shaped API) vs the NumPy oracle."""
import numpy as np
import pytest
import torch

from _parity import assert_close_mostly, away_from_clip, close_scaled, near_kink_rows

pytestmark = pytest.mark.gpu
F = np.float32

NET_OPTS = dict(common_net_shape=[], critic_net_shape=[1024, 512], actor_net_shape=[1024, 512],
                value_loss_coef=0.5, gamma=0.95, normalize_state=True, clip_state=5.0,
                normalize_value=False, clip_value=None, normalize_advantage=True, clip_advantage=None,
                particles=35, resample=-1, resample_interval=368)


def make_net(dev, S=197, A=36, seed=34114, **over):
    from pfpn_amd.networks import ParticleFilteringClipPPONetwork
    opts = dict(NET_OPTS); opts.update(over)
    net = ParticleFilteringClipPPONetwork(trainable=True, state_shape=[S], action_shape=[A],
                                          action_lower_bound=-np.ones(A), action_upper_bound=np.ones(A),
                                          device=dev, seed=seed, **opts)
    net.init()
    return net


def make_batch(rng, B, S=197, A=36):
    return {"state": (rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F),
            "action": rng.uniform(-1.1, 1.1, (B, A)).astype(F),
            "value": rng.normal(0, 0.1, B).astype(F),
            "log_prob": None,
            "advantage": rng.normal(0, 1, B).astype(F)}


def sync_oracle_state(oracle, net, st):
    for k, v in net.variables().items():
        st.P[k] = v.detach().cpu().numpy().reshape(st.P[k].shape).copy()


@pytest.mark.parametrize("B", [32, 256, 4096])          # 4096 = BASELINE.json config 2, the bench's batch
def test_dppo_step_matches_oracle(oracle, dev, B):
    from pfpn_amd.models import SyncReplicasApplier
    net = make_net(dev)
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    st = oracle.DPPOState(np.random.default_rng(0))
    sync_oracle_state(oracle, net, st)
    rng = np.random.default_rng(B)
    n_it = 3 if B < 4096 else 2
    for it in range(n_it):
        batch = make_batch(rng, B)
        for _ in range(50):     # keep every pre-activation off the ReLU6 kinks (see near_kink_rows)
            x = oracle.state_normalize(batch["state"], st.norm_mean, st.norm_std, 5.0)
            bad = near_kink_rows(st.P, x)
            if not bad.any():
                break
            batch["state"][bad] = make_batch(rng, int(bad.sum()))["state"]
        assert not bad.any()
        # old log-prob: the current policy's log-prob of the action, perturbed (ratios around 1)
        logits, _ = oracle.actor_forward(st.P, x, 36, 35)
        lp, _ = oracle.mixture_log_prob(logits, batch["action"], st.P["actor/samples"], st.P["actor/samples_std"])
        batch["log_prob"] = away_from_clip(lp, (lp + rng.normal(0, 0.1, B)).astype(F))
        # oracle gradients of this step (before its update) for a direct gradient check
        (l_ref, pl_ref, vl_ref), G_ref, aux = oracle.dppo_grads(st.P, st.norm_mean, st.norm_std, batch, 36, 35)
        _, (_, c2) = oracle.critic_forward(st.P, aux["x"])
        v_terms = np.abs(c2 * st.P["critic/fc3/weight"][:, 0]).sum(1).max()      # size of the head's dot-product terms
        losses = oracle.dppo_learner_step(st, [batch])
        loss, extra = net.train(None, opt, ["global_step"], batch["state"], batch["action"], batch["value"],
                                batch["log_prob"], batch["advantage"])
        assert extra == [it + 1]
        # north_star: fp32 log-prob / value within 1e-5.  log-prob: per element (|logp| ~ 40); value: 1e-5 of the
        # size of the terms of the critic head's 512-term dot product (with the 0.01-sigma initial weights the terms
        # cancel to a value 20x smaller than their sum, and fp32 cannot hold 1e-5 of THAT); losses: of max(|loss|, 1),
        # the size of the surrogate terms whose signed mean the policy loss is
        np.testing.assert_allclose(net.train_out["logp"].cpu().numpy(), aux["logp"], rtol=1e-5, atol=0)
        close_scaled(net.value_buf.cpu().numpy(), aux["value"], 1e-5, "value", floor=v_terms)
        close_scaled([loss[0], loss[2]], [l_ref, pl_ref], 1e-5, "loss", floor=1.0)
        close_scaled(loss[3], vl_ref, 1e-5, "value loss")
        assert loss[1] is None
        # gradients (before clipping): 2e-4 of the tensor's scale.  Every actor gradient is proportional to the ratio
        # exp(logp - logp_old), and a fp32 log-prob of size ~40 (36 summed terms) is only good to ~1e-4 ABSOLUTE
        # (that is the 2.7e-6 relative measured here, inside north_star's 1e-5): the ratio, hence a whole row of
        # dlogits, carries that 1e-4 relatively.  Measured: 9e-5 here, 2-4e-5 against the reference graph's own
        # gradients (tools/parity_report.py); 1e-6 on the critic side.  No allowance for flipped ReLU6 masks: the
        # batch was drawn off the kinks
        G = net.gradients()
        for name, g_ref in G_ref.items():
            close_scaled(G[name].cpu().numpy(), g_ref, 2e-4, "grad " + name)
        # parameters after the update: within 2 % of one step (lr = 1e-4).  From zero moments the first Adam steps are
        # ~lr * sign(g): entries with |g| <~ 1e-7 (eps / sqrt(1 - beta2)) turn their gradient's relative error into
        # the step, everything else moves by exactly +-lr
        for name, v in net.variables().items():
            close_scaled(v.cpu().numpy(), st.P[name], 2e-6, name, floor=1.0)
        close_scaled(net.state_normalizer.mean.cpu().numpy(), st.norm_mean, 1e-5, "norm mean", floor=1.0)
        close_scaled(net.state_normalizer.std.cpu().numpy(), st.norm_std, 1e-5, "norm std")
    assert net.global_step == n_it and net.train_flag == n_it


def test_resample_tick_and_policy_api(oracle, dev):
    """run()/evaluate()/GAE surface + the resample tick firing at resample_interval."""
    from pfpn_amd.models import SyncReplicasApplier
    net = make_net(dev, resample_interval=4)
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    rng = np.random.default_rng(1)
    s = rng.normal(0, 1, 197).astype(F)
    a, lp, v = net.run(None, s)
    assert a.shape == (36,) and np.isfinite(lp) and np.isfinite(v)
    assert abs(net.evaluate(None, s) - v) < 1e-6
    assert net.sum_active.sum().item() == pytest.approx(2 * 36, rel=1e-4)     # run + evaluate each add softmax rows
    # log_prob returned by run() == log_prob of that action under the policy
    st = torch.from_numpy(s[None]).to(dev)
    policy, _, _ = net._forward(st)
    lp2 = policy.log_prob(torch.from_numpy(a[None]).to(dev))[0].item()
    assert abs(lp2 - lp) < 1e-4 * abs(lp)
    adv = net.generalized_advantage_estimate(list(rng.normal(0, 1, 9).astype(F)), list(rng.normal(0, 1, 10).astype(F)))
    assert len(adv) == 9
    # drive a few particles dead, then 4 train steps trigger the resample and zero the statistics
    for _ in range(40):
        net.run(None, rng.normal(0, 1, 197).astype(F))
    for it in range(4):
        b = make_batch(rng, 32)
        b["log_prob"] = np.full(32, -40.0, F)
        net.train(None, opt, [], b["state"], b["action"], b["value"], b["log_prob"], b["advantage"])
    assert net.train_flag == 0 and net.sum_active.abs().sum().item() == 0 and net.max_active.abs().sum().item() == 0
    dead, src = net.last_resample
    # with near-uniform weights (1/35 >> 0.05/35) nothing is dead: the tick resampled no particle (that an all-alive
    # tick leaves every tensor untouched is tests/test_gpu_learner_utils.py::test_resample_noop_when_all_alive)
    assert dead.sum().item() == 0


def test_deterministic_evaluator(dev):
    from pfpn_amd.networks import ParticleFilteringClipPPONetwork
    A = 36
    net = ParticleFilteringClipPPONetwork(trainable=False, state_shape=[197], action_shape=[A],
                                          action_lower_bound=-np.ones(A), action_upper_bound=np.ones(A),
                                          device=dev, seed=1, **NET_OPTS)
    net.init()
    s = np.random.default_rng(0).normal(0, 1, 197).astype(F)
    out1, out2 = net.run(None, s), net.run(None, s)
    assert len(out1) == 1 and np.array_equal(out1[0], out2[0])
    # deterministic action = loc[a, argmax logits]: a grid point of the (untrained) particle grid
    grid = -1 + 2 * np.arange(35) / 34
    assert np.all(np.abs(out1[0][:, None] - grid[None]).min(1) < 1e-6)


def test_train_captures_itself_and_matches_eager(dev, monkeypatch):
    """train() switches to a hipGraph replay after a few steps at one minibatch size: same bits as the eager path."""
    from pfpn_amd.models import SyncReplicasApplier
    nets = [make_net(dev, seed=21, resample_interval=5) for _ in range(2)]
    opts = [SyncReplicasApplier(n, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1) for n in nets]
    rng = np.random.default_rng(2)
    for it in range(9):
        b = make_batch(rng, 32)
        b["log_prob"] = np.full(32, -40.0, F)
        losses = []
        for i, (net, opt) in enumerate(zip(nets, opts)):
            monkeypatch.setenv("PFPN_GRAPH_TRAIN", "1" if i == 0 else "0")
            losses.append(net.train(None, opt, ["global_step"], b["state"], b["action"], b["value"], b["log_prob"],
                                    b["advantage"]))
        assert losses[0] == losses[1], it
        assert torch.equal(nets[0].store.flat, nets[1].store.flat), it
        assert torch.equal(nets[0].state_normalizer.std, nets[1].state_normalizer.std)
    assert len(nets[0]._train_graphs) == 1 and len(nets[1]._train_graphs) == 0
    assert nets[0].global_step == 9 and nets[0].train_flag == nets[1].train_flag      # resample ticks ran on both


@pytest.mark.parametrize("B", [32, 4096])
def test_deferred_reduce_step_matches_immediate_step(dev, monkeypatch, B):
    """The step that leaves its small reductions (mixture finalize, bias gradients, critic head, normaliser EMA) to
    ONE launch produces the same bits as the step that launches each of them where the reference's graph has them."""
    from pfpn_amd.models import SyncReplicasApplier
    nets, opts = [], []
    monkeypatch.setenv("PFPN_GRAPH_TRAIN", "0")
    for i in range(2):
        nets.append(make_net(dev, seed=5, resample_interval=3))
        nets[-1].defer_reduce = i == 0
        # (fused_norm=False: the norm's partial sums group differently when they ride in the reduction launch: own test below)
        opts.append(SyncReplicasApplier(nets[-1], [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1,
                                        fused_norm=False))
    rng = np.random.default_rng(B)
    for it in range(4):
        b = make_batch(rng, B)
        b["log_prob"] = np.full(B, -40.0, F)
        out = []
        for i, (net, opt) in enumerate(zip(nets, opts)):
            out.append(net.train(None, opt, ["global_step"], b["state"], b["action"], b["value"], b["log_prob"],
                                 b["advantage"]))
        assert nets[0]._defer_ok[(B, True)] is True and nets[1]._defer_ok[(B, True)] is False
        assert out[0] == out[1], it
        assert torch.equal(nets[0].store.grad, nets[1].store.grad), it
        assert torch.equal(nets[0].store.flat, nets[1].store.flat), it
        assert torch.equal(nets[0].value_buf, nets[1].value_buf)
        assert torch.equal(nets[0].state_normalizer.mean, nets[1].state_normalizer.mean)
        assert torch.equal(nets[0].state_normalizer.std, nets[1].state_normalizer.std)


def test_multi_step_graph_replay_matches_single_steps(dev):
    """K learner steps captured in one hipGraph == K one-step replays, bit for bit; a replay is refused when a resample
    tick would fall inside it, and the tick still fires on the right step."""
    from pfpn_amd.graphs import GraphedLearnerStep
    from pfpn_amd.models import SyncReplicasApplier
    B, K = 64, 3
    nets = [make_net(dev, seed=13, resample_interval=7) for _ in range(2)]
    opts = [SyncReplicasApplier(n, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1) for n in nets]
    rng = np.random.default_rng(8)
    sets = []
    for _ in range(K):
        b = make_batch(rng, B); b["log_prob"] = np.full(B, -40.0, F)
        sets.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ("state", "action", "value", "log_prob", "advantage")))
    gk = GraphedLearnerStep(nets[0], opts[0], steps_per_replay=K, input_sets=[tuple(t.clone() for t in s) for s in sets])
    g1 = GraphedLearnerStep(nets[1], opts[1], *[t.clone() for t in sets[0]])
    for rep in range(2):                                   # steps 1-3 and 4-6: no tick inside (interval 7)
        assert gk.can_replay()
        gk()
        for s in sets:
            g1(s)
        assert torch.equal(nets[0].store.flat, nets[1].store.flat), rep
        assert torch.equal(nets[0].state_normalizer.std, nets[1].state_normalizer.std)
        assert nets[0].global_step == nets[1].global_step == 3 * (rep + 1) and nets[0].train_flag == nets[1].train_flag
    assert nets[0].steps_until_tick() == 1 and not gk.can_replay()          # step 7 resamples: not inside a 3-step replay
    with pytest.raises(ValueError):
        gk()
    nets[0].max_active.fill_(1.0); nets[1].max_active.fill_(1.0)
    g1k = GraphedLearnerStep(nets[0], opts[0], *[t.clone() for t in sets[0]])
    g1k(sets[0]); g1(sets[0])
    assert nets[0].train_flag == nets[1].train_flag == 0 and torch.equal(nets[0].store.flat, nets[1].store.flat)
    assert gk.can_replay()


def test_next_prologue_rides_in_the_optimiser_launch(dev, monkeypatch):
    """Inside a K-step capture the optimiser launch of step k carries step k+1's prologue as extra blocks (K - 1 riders, one
    launch less per step); `net.ride_prologue = False` keeps the separate launch.  Same bits either way."""
    from pfpn_amd import kernels
    from pfpn_amd.graphs import GraphedLearnerStep
    from pfpn_amd.models import SyncReplicasApplier
    B, K = 128, 4
    rng = np.random.default_rng(21)
    sets = []
    for _ in range(K):
        b = make_batch(rng, B); b["log_prob"] = np.full(B, -40.0, F)
        sets.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ("state", "action", "value", "log_prob", "advantage")))
    riders, plain = [], []
    real_adam, real_pro = kernels.clip_adam, kernels.state_prologue
    monkeypatch.setattr(kernels, "clip_adam", lambda *a, **kw: (riders.append(kw.get("rider") is not None), real_adam(*a, **kw))[1])
    monkeypatch.setattr(kernels, "state_prologue", lambda *a, **kw: (plain.append(1), real_pro(*a, **kw))[1])
    finals = []
    for flag, want_riders in ((True, K - 1), (False, 0)):
        net = make_net(dev, seed=17, resample_interval=50)
        net.ride_prologue = flag
        opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
        g = GraphedLearnerStep(net, opt, steps_per_replay=K, input_sets=[tuple(t.clone() for t in s_) for s_ in sets],
                               warmup=1)
        # the capture is the last K optimiser calls / the prologue launches after the warm-up's one
        assert sum(riders[-K:]) == want_riders, (flag, riders)
        assert sum(plain[-(K - want_riders):]) == K - want_riders and len(plain) >= K - want_riders
        riders.clear(); plain.clear()
        g(); g()
        torch.cuda.synchronize()
        finals.append((net.store.flat.clone(), net.store.m.clone(), net.state_normalizer.mean.clone(),
                       net.state_normalizer.std.clone(), g.loss3.clone()))
        assert net.global_step == 2 * K
    for a, b in zip(*finals):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B", [64, 4096])
def test_global_norm_from_the_reduction_launch(dev, monkeypatch, B):
    """The reduction launch that also yields the global-norm partial sums (+ the one-launch optimiser) against the separate
    square-sum launch: identical gradients, the same norm to fp32 summation order, the same parameters to that precision,
    identical Adam powers."""
    from pfpn_amd.models import SyncReplicasApplier
    nets, opts = [], []
    monkeypatch.setenv("PFPN_GRAPH_TRAIN", "0")
    for i in range(2):
        nets.append(make_net(dev, seed=17, resample_interval=3))
        opts.append(SyncReplicasApplier(nets[-1], [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1,
                                        fused_norm=i == 0))
    assert nets[0]._norm_hook is not None and nets[1]._norm_hook is None
    rng = np.random.default_rng(B + 1)
    for it in range(4):
        b = make_batch(rng, B)
        b["log_prob"] = np.full(B, -40.0, F)
        out = [net.train(None, opt, ["global_step"], b["state"], b["action"], b["value"], b["log_prob"], b["advantage"])
               for net, opt in zip(nets, opts)]
        assert out[0] == out[1], it
        if it == 0:
            assert torch.equal(nets[0].store.grad, nets[1].store.grad)
        n0, n1 = opts[0].norm_scale.cpu().numpy(), opts[1].norm_scale.cpu().numpy()
        np.testing.assert_allclose(n0, n1, rtol=2e-6)
        ref = nets[0].store.grad.double().pow(2).sum().sqrt().item() if it == 0 else None
        if ref is not None:
            np.testing.assert_allclose(n0[0], ref, rtol=2e-6)
        assert torch.equal(opts[0].beta_pow, opts[1].beta_pow)
        torch.testing.assert_close(nets[0].store.flat, nets[1].store.flat, rtol=1e-5, atol=2e-8)   # Adam amplifies the 1e-7 relative norm difference on near-zero parameters
        assert torch.equal(nets[0].state_normalizer.std, nets[1].state_normalizer.std)


def test_config5_shaped_learner_step_matches_oracle(oracle, dev):
    """BASELINE.json config 5 as a LEARNER STEP (128 action dims x 256 particles: head 512 -> 32768): the PPO network at
    that shape takes the composed path (wide forward -> loss head -> wide backward, the fused kernel covers N <= 64) and
    matches the oracle's step -- log-prob, losses, every gradient, parameters."""
    from pfpn_amd.models import SyncReplicasApplier
    A, N, B = 128, 256, 64
    net = make_net(dev, A=A, particles=N)
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    st = oracle.DPPOState(np.random.default_rng(0), 197, A, N)
    sync_oracle_state(oracle, net, st)
    assert st.P["actor/fc_policy/weight"].shape == (512, A * N)
    rng = np.random.default_rng(5)
    for it in range(2):
        batch = make_batch(rng, B, A=A)
        batch["action"] = rng.uniform(-1.0, 1.0, (B, A)).astype(F)        # inside the grid: sigma = 2/255
        for _ in range(50):
            x = oracle.state_normalize(batch["state"], st.norm_mean, st.norm_std, 5.0)
            bad = near_kink_rows(st.P, x)
            if not bad.any():
                break
            batch["state"][bad] = make_batch(rng, int(bad.sum()), A=A)["state"]
        logits, _ = oracle.actor_forward(st.P, x, A, N)
        lp, _ = oracle.mixture_log_prob(logits, batch["action"], st.P["actor/samples"], st.P["actor/samples_std"])
        batch["log_prob"] = away_from_clip(lp, (lp + rng.normal(0, 0.1, B)).astype(F))
        (l_ref, pl_ref, vl_ref), G_ref, aux = oracle.dppo_grads(st.P, st.norm_mean, st.norm_std, batch, A, N)
        oracle.dppo_learner_step(st, [batch])
        loss, _ = net.train(None, opt, [], batch["state"], batch["action"], batch["value"], batch["log_prob"],
                            batch["advantage"])
        np.testing.assert_allclose(net.train_out["logp"].cpu().numpy(), aux["logp"], rtol=1e-5, atol=0)
        close_scaled([loss[0], loss[2]], [l_ref, pl_ref], 1e-5, "loss", floor=1.0)
        # gradients: every actor gradient carries the ratio exp(logp - logp_old), and a fp32 log-prob summed over 128
        # dims (|logp| ~ 3e2) is good to ~3e-4 absolute: 1e-3 of the tensor's scale (measured 3e-4 on the particle
        # locations, whose terms are amplified by 1/sigma = 128)
        G = net.gradients()
        for name, g_ref in G_ref.items():
            close_scaled(G[name].cpu().numpy(), g_ref, 1e-3, "grad " + name)
        for name, v in net.variables().items():       # within 5 % of one Adam step (lr = 1e-4; see the step test above)
            close_scaled(v.cpu().numpy(), st.P[name], 5e-6, name, floor=1.0)


def test_learner_raises_instead_of_training_on_after_an_expired_one_launch_pass(dev, monkeypatch, tmp_path):
    """the one-launch PPO pass of the wide grids needs the device to itself; a wait that expires makes
    its gradients NaN, and the optimiser applies them.  The expiry leaves a sticky mark in the device's fault word and the
    learner RAISES at the loss read-back of that very step (not a NaN line in the [TRAIN] log); a checkpoint save refuses too.
    Forced as in tests/test_gpu_mixture.py: a grid twice the device's capacity whose chunks take every C-th block."""
    from pfpn_amd import checkpoint, kernels
    from pfpn_amd.models import SyncReplicasApplier
    A, N, B = 128, 256, 2048
    monkeypatch.setenv("PFPN_GRAPH_TRAIN", "0")              # eager steps: the knobs below are read at every launch
    net = make_net(dev, A=A, particles=N)
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    rng = np.random.default_rng(11)
    batch = make_batch(rng, B, A=A)
    batch["action"] = rng.uniform(-1.0, 1.0, (B, A)).astype(F)
    with torch.no_grad():                                    # old log-prob = the initial policy's own: ratios of 1
        policy, _, _ = net._forward(torch.from_numpy(batch["state"]).to(dev))
        batch["log_prob"] = policy.log_prob(torch.from_numpy(batch["action"]).to(dev)).cpu().numpy()
    args = (batch["state"], batch["action"], batch["value"], batch["log_prob"], batch["advantage"])
    kernels.check_faults()
    loss, _ = net.train(None, opt, [], *args)                # a clean step through the one-launch pass
    assert np.isfinite(loss[0]) and bool(torch.isfinite(net.store.flat).all())
    monkeypatch.setenv("PFPN_WIDE_F2L_OVERSUBSCRIBE", "1")
    monkeypatch.setenv("PFPN_WIDE_F2L_SAFE", "3")
    try:
        with pytest.raises(kernels.DeviceFault, match="PFPN_WIDE_FUSED=0"):
            net.train(None, opt, [], *args)
        assert not bool(torch.isfinite(net.store.flat).all())     # what the exception is about: the run IS poisoned
        monkeypatch.delenv("PFPN_WIDE_F2L_OVERSUBSCRIBE")
        monkeypatch.delenv("PFPN_WIDE_F2L_SAFE")
        with pytest.raises(kernels.DeviceFault):                  # sticky: later (clean) launches do not clear it
            net.train(None, opt, [], *args)
        with pytest.raises(kernels.DeviceFault):
            checkpoint.save(str(tmp_path / "model.ckpt-1"), net, opt)
        assert not list(tmp_path.iterdir())                       # nothing was written
    finally:
        kernels.clear_faults()
