"""Flip census on UNCURATED batches (round 4): the learner step's loss has step discontinuities --
Relu6Grad (reference networks/ops.py:105-117), PPO's clipped surrogate (networks/actor_critic/ppo.py:44-54), SAC's
min(Q1, Q2) (sac.py:155-165) -- and two fp32 implementations MUST take different branches on the few units whose argument
sits within summation-order error of a kink.  The other step tests draw their batches away from the kinks; these do not.

Per batch, HIP path against the oracle:
  (i)   the census: every unit / row on which the two took different branches is found (from the HIP path's own
        activation buffers and gradient rows), counted, printed, and bounded;
  (ii)  the oracle re-evaluated with exactly those decisions forced (oracle `force=`) agrees with the HIP gradients at the
        NORMAL ledger bound everywhere, no free entries: a flip moves what it must and nothing else;
  (iii) what has no discontinuity -- log-prob, value, Q -- meets 1e-5 on every row of the uncurated batch.
The bounds and their rationale: DESIGN.md section 2.
"""
import json
import os

import numpy as np
import pytest
import torch

from _parity import close_scaled

pytestmark = pytest.mark.gpu
F = np.float32
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "flip_census.jsonl")


def relu6_on(act):
    return ((act > 0) & (act < 6)).cpu().numpy()


def census(own, other):
    """{branch name: indices where the two implementations decided differently}"""
    out = {}
    for k in own:
        d = np.argwhere(np.asarray(own[k]) != np.asarray(other[k]))
        if len(d):
            out[k] = d.tolist()
    return out


def report(tag, flips, extra=None):
    n = {k: len(v) for k, v in flips.items()}
    rec = dict(test=tag, flipped=n, where={k: v[:16] for k, v in flips.items()}, **(extra or {}))
    print("flip census", json.dumps(rec))
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    return n


def beyond(got, ref, tol):
    got, ref = np.asarray(got, np.float64).reshape(np.shape(ref)), np.asarray(ref, np.float64)
    return int((np.abs(got - ref) > tol * max(np.abs(ref).max(), 1e-30)).sum())


# measured on MI355X (gpurun_out/flip_census.jsonl): <= 2 units per 4096-row batch of 12.6 M
# pre-activations, <= 1 PPO row; the bounds leave a factor ~4 for another GEMM selection / another box
MAX_RELU6_FLIPS_DPPO, MAX_PPO_FLIPS = 8, 4
MAX_RELU6_FLIPS_SAC, MAX_SEL_FLIPS = 8, 4


def test_flip_census_dppo_step_4096_rows_uncurated(oracle, dev):
    """BASELINE.json config 2, plain `make_batch` draws: no near_kink_rows redraw, no away_from_clip."""
    from test_gpu_dppo_step import make_batch, make_net, sync_oracle_state
    from pfpn_amd.models import SyncReplicasApplier
    B, A, N = 4096, 36, 35
    net = make_net(dev)
    opt = SyncReplicasApplier(net, [{"names": None, "lr": 1e-4}], norm_clip=1.0, world_size=1)
    st = oracle.DPPOState(np.random.default_rng(0))
    sync_oracle_state(oracle, net, st)
    rng = np.random.default_rng(40960)
    total = {}
    for it in range(3):
        batch = make_batch(rng, B)
        x = oracle.state_normalize(batch["state"], st.norm_mean, st.norm_std, 5.0)
        logits, _ = oracle.actor_forward(st.P, x, A, N)
        lp, _ = oracle.mixture_log_prob(logits, batch["action"], st.P["actor/samples"], st.P["actor/samples_std"])
        batch["log_prob"] = (lp + rng.normal(0, 0.1, B)).astype(F)              # ratios on both sides of both clip bounds
        P0 = {k: v.copy() for k, v in st.P.items()}
        m0, s0 = st.norm_mean.copy(), st.norm_std.copy()
        _, G_ref, aux = oracle.dppo_grads(P0, m0, s0, batch, A, N)
        _, (_, c2) = oracle.critic_forward(P0, aux["x"])
        v_terms = np.abs(c2 * P0["critic/fc3/weight"][:, 0]).sum(1).max()
        net.train(None, opt, [], batch["state"], batch["action"], batch["value"], batch["log_prob"], batch["advantage"])
        # (iii) continuous outputs on EVERY row
        np.testing.assert_allclose(net.train_out["logp"].cpu().numpy(), aux["logp"], rtol=1e-5, atol=0)
        close_scaled(net.value_buf.cpu().numpy(), aux["value"], 1e-5, "value", floor=v_terms)
        # (i) the HIP path's own branch decisions
        act = net.trunk._last["act"]                                             # [G = actor, critic][B, H] per layer
        assert net.trunk.prefixes == ["actor", "critic"]
        hip = {"%s/fc%d" % (p, i + 1): relu6_on(act[i][g]) for g, p in enumerate(("actor", "critic")) for i in range(2)}
        hip["ppo_on"] = (net.train_out["dlogits"].abs().amax(dim=1) > 0).cpu().numpy()
        flips = census(hip, aux["branches"])
        G = {k: v.cpu().numpy() for k, v in net.gradients().items()}
        moved = {k: beyond(G[k], g, 2e-4) for k, g in G_ref.items()}
        n = report("dppo_b4096_it%d" % it, flips, {"entries_beyond_2e-4_before_forcing": {k: v for k, v in moved.items() if v}})
        for k, c in n.items():
            total[k] = total.get(k, 0) + c
        assert sum(c for k, c in n.items() if k != "ppo_on") <= MAX_RELU6_FLIPS_DPPO, n
        assert n.get("ppo_on", 0) <= MAX_PPO_FLIPS, n
        if not flips:
            assert not any(moved.values()), moved                                # nothing flipped: nothing may be off
        # (ii) same decisions -> same gradients at the ledger bound, every entry
        _, G_forced, _ = oracle.dppo_grads(P0, m0, s0, batch, A, N, force=hip)
        for name, g_ref in G_forced.items():
            close_scaled(G[name], g_ref, 2e-4, "grad %s (branches forced)" % name)
        # the next batch meets moved parameters: the oracle follows the HIP path's update (the update itself is what
        # tests/test_gpu_dppo_step.py checks; after a flip the two sides' Adam steps differ by construction)
        sync_oracle_state(oracle, net, st)
        st.norm_mean = net.state_normalizer.mean.cpu().numpy().copy()
        st.norm_std = net.state_normalizer.std.cpu().numpy().copy()
    report("dppo_b4096_total", {k: [0] * c for k, c in total.items()})


def test_flip_census_sac_step_1024_rows_uncurated(oracle, dev):
    """BASELINE.json config 3 (replay batch 1024), plain draws.  Branches: Relu6Grad of the actor trunk (on s), of both live
    critics on (s, a_pi) and on (s, action_hist), and the twin minimum's gradient routing; the reparameterised sample's
    argmax is index parity (bit-exact by construction) and is asserted, not counted."""
    from test_gpu_sac_step import make_net
    S, A, N, B = 197, 36, 35, 1024
    net = make_net(dev)
    for op in net.init_ops:
        op()
    st = oracle.SACState(np.random.default_rng(0), S, A, N)
    for k, v in net.variables().items():
        st.P[k] = v.detach().cpu().numpy().reshape(st.P[k].shape).copy()
    for k, v in net.target_variables().items():
        st.PT[k] = v.detach().cpu().numpy().reshape(st.PT[k].shape).copy()
    rng = np.random.default_rng(10240)
    for it in range(3):
        batch = {"state": (rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F),
                 "action": np.tanh(rng.normal(0, 1, (B, A))).astype(F),
                 "reward": rng.random(B).astype(F),
                 "not_terminal": (rng.random(B) < 0.9).astype(F),
                 "state_": (rng.normal(0, 1, (B, S)) * 2 + 0.3).astype(F)}
        noise = {k: (-np.log(-np.log(rng.uniform(1e-10, 1, (B, A, N)))) if k.startswith("g")
                     else rng.normal(0, 1, (B, A, N))).astype(F) for k in ("gumbel", "eps", "gumbel_", "eps_")}
        _, G_ref, aux = oracle.sac_grads(st.P, st.PT, st.norm_mean, st.norm_std, batch, noise, A, N)
        dn = {k: torch.from_numpy(v.reshape(B, A * N)).to(dev) for k, v in noise.items()}
        buf = torch.stack([dn[k] for k in ("gumbel", "gumbel_", "eps", "eps_")])
        dn = {"gumbel": buf[0], "gumbel_": buf[1], "eps": buf[2], "eps_": buf[3], "_buf": buf}
        args = [net._to_dev(batch[k]) for k in ("state", "action", "reward", "not_terminal", "state_")]
        net.compute_gradients(*args, noise=dn)
        assert np.array_equal(net.last["k"].cpu().numpy(), aux["k"]), "rsample particle indices must be bit-exact"
        # (iii) continuous outputs on every row
        np.testing.assert_allclose(net.last["logp"].cpu().numpy(), aux["logp"], rtol=1e-5, atol=2e-4)
        q_pi = net.last["q_pi"].cpu().numpy()
        q_terms = max(np.abs(aux["q1p"]).max(), np.abs(aux["q2p"]).max(), 1e-3)
        close_scaled(q_pi[0], aux["q1p"], 1e-5, "Q1(s, a_pi)", floor=10 * q_terms)
        close_scaled(q_pi[1], aux["q2p"], 1e-5, "Q2(s, a_pi)", floor=10 * q_terms)
        # (i) the HIP path's decisions: actor trunk over [s; s'] (first B rows = s), twin critic over [(s, a_pi); (s, a_hist)]
        a_act = net.actor_trunk._bufs[(2 * B, 0)]["act"]
        q_act = net.q_stack._bufs[(2 * B, 0)]["act"]
        hip = {"actor/fc%d" % (i + 1): relu6_on(a_act[i][0][:B]) for i in range(2)}
        for g, q in enumerate(("q1", "q2")):
            for i in range(2):
                hip["%s/pi/fc%d" % (q, i + 1)] = relu6_on(q_act[i][g][:B])
                hip["%s/run/fc%d" % (q, i + 1)] = relu6_on(q_act[i][g][B:])
        hip["sel1"] = (net.sac_out["dq_pi"][0] != 0).cpu().numpy()
        flips = census(hip, aux["branches"])
        G = {k: v.cpu().numpy() for k, v in net.gradients().items()}
        moved = {k: beyond(G[k], np.asarray(g, F), 3e-4) for k, g in G_ref.items()}
        n = report("sac_b1024_it%d" % it, flips, {"entries_beyond_3e-4_before_forcing": {k: v for k, v in moved.items() if v}})
        assert sum(c for k, c in n.items() if k != "sel1") <= MAX_RELU6_FLIPS_SAC, n
        assert n.get("sel1", 0) <= MAX_SEL_FLIPS, n
        if not flips:
            assert not any(moved.values()), moved
        # (ii) same decisions -> same gradients, every entry (3e-4 of scale: the bound of test_gpu_sac_step.py's atol,
        # here with no fraction of free elements)
        _, G_forced, _ = oracle.sac_grads(st.P, st.PT, st.norm_mean, st.norm_std, batch, noise, A, N, force=hip)
        for name, g_ref in G_forced.items():
            close_scaled(G[name], np.asarray(g_ref, F), 3e-4, "grad %s (branches forced)" % name)
        # next batch on moved parameters: the oracle follows the HIP path's update
        oracle.sac_learner_step(st, batch, noise)
        from pfpn_amd.models import SyncReplicasApplier
        if it == 0:
            opt = SyncReplicasApplier(net, [{"names": net.critic_param_names(), "lr": 1e-4},
                                            {"names": net.actor_param_names(), "lr": 1e-4}], norm_clip=1.0, world_size=1)
        opt.apply(net)
        for k, v in net.variables().items():                                    # keep both sides on the SAME parameters
            st.P[k] = v.detach().cpu().numpy().reshape(st.P[k].shape).copy()
        for k, v in net.target_variables().items():
            st.PT[k] = v.detach().cpu().numpy().reshape(st.PT[k].shape).copy()
        st.norm_mean = net.state_normalizer.mean.cpu().numpy().copy()
        st.norm_std = net.state_normalizer.std.cpu().numpy().copy()
