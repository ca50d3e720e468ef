"""Does the assembled learner LEARN?  Harness shared by tests/test_gpu_learning.py and tools/learn_curve.py.

Evidence that does not pass through the oracle: the reference's only integration test is to start
`benchmark.sh ... --train` and watch the [PERFORM] / [TRAIN] lines (reference README.md:24-33,
models/distributed_model.py:154-193 the rollout loop, :241-259 the [PERFORM] line, networks/actor_critic/a2c.py:367-383
the resample tick).  Here the same launcher (`pfpn_amd.main.parse` -> `main.build` -> `model.start`) runs on the synthetic
DeepMimic-shaped environment (pfpn_amd/envs.py: reward exp(-0.05 |s[:8]|^2), s' = 0.95 s + a W + noise -- the reward
is controllable through the action) and three things are recorded from the OUTSIDE of the product:

* the return of every training episode (hook on `worker.after_episode`; the worker's own `total_reward` bookkeeping);
* every resample tick that fires in training (hook on `net.resample_now`): which particles the rollout statistics
  declared dead (max_active < 0.05 / N, a2c.py:392-396 -- never hand-set here), and whether the tick kept the
  unnormalised mixture mass: a source t with c clones ends with bias_t - log(c + 1) on all c + 1 columns and its
  weight column copied (a2c.py:457-471), so  sum_{j in {t} + clones} exp(logit_j)  after  ==  exp(logit_t)  before,
  for ANY hidden vector -- evaluated in float64 torch on random hidden vectors, not by the kernels under test;
* where the particles are at the end (distance from the initial grid, a2c.py:476-523).
"""
import time

import numpy as np

SETTINGS = {"dppo": ("settings.deepmimic.deepmimic_dppo_pfpn", "DeepMimicWalk-v0"),
            "sac": ("settings.deepmimic.deepmimic_sac_pfpn", "DeepMimicKick-v0"),
            "a3c": ("settings.deepmimic.deepmimic_a3c_pfpn", "DeepMimicWalk-v0"),
            "impala": ("settings.deepmimic.deepmimic_impala_pfpn", "DeepMimicWalk-v0")}


class Record(object):
    def __init__(self):
        self.episodes = []          # (global_step, total_samples, length, return)
        self.ticks = []             # one dict per resample tick that fired
        self.t0 = time.perf_counter()

    def returns(self, full_length=None):
        return np.array([e[3] for e in self.episodes if full_length is None or e[2] >= full_length], np.float64)

    def summary(self, k=10):
        r = self.returns()
        return {"episodes": len(r), "first_k": float(r[:k].mean()) if len(r) else None,
                "last_k": float(r[-k:].mean()) if len(r) else None, "k": k,
                "ticks": len(self.ticks), "dead_per_tick": [t["dead"] for t in self.ticks]}


def _tick_probe(net, rec):
    """wrap net.resample_now: snapshot the head before, analyse after (float64 torch, no product kernel involved)."""
    import torch
    plain = net.resample_now

    def resample_now(*a, **kw):
        A, N = net.n_actions, net.particles
        thr = float(net.resample_threshold) if net.resample_threshold else 0.05 / N
        mx, sm = net.max_active.clone(), net.sum_active.clone()
        W0, b0 = net.policy_weight.double().clone(), net.policy_bias.double().clone()
        loc0 = net.loc.clone()
        out = plain(*a, **kw)
        dead, src = [t.cpu().numpy().astype(bool if i == 0 else np.int64) for i, t in enumerate(out)]
        dead_by_stats = (mx < thr).cpu().numpy()
        W1, b1 = net.policy_weight.double(), net.policy_bias.double()
        g = torch.Generator(device="cpu").manual_seed(len(rec.ticks))
        h = (torch.rand(32, W0.shape[0], generator=g, dtype=torch.float64) * 2.0).to(W0.device)   # ReLU6 range sample
        e0 = torch.exp((h @ W0 + b0).view(32, A, N))
        e1 = torch.exp((h @ W1 + b1).view(32, A, N))
        worst, groups, chained, copied = 0.0, 0, 0, True
        for a_ in range(A):
            dn = np.flatnonzero(dead[a_])
            for t in np.unique(src[a_][dn]) if len(dn) else []:
                clones = dn[src[a_][dn] == t]
                if dead[a_][t]:
                    chained += 1            # a dead particle drawn as a source (the ordered scatters of a2c.py:465-466)
                    continue
                after = e1[:, a_, t] + e1[:, a_, clones].sum(1)
                worst = max(worst, float(((after - e0[:, a_, t]).abs() / e0[:, a_, t]).max().item()))
                cols = a_ * N + clones
                copied = copied and bool(torch.equal(W1[:, cols], W0[:, a_ * N + t:a_ * N + t + 1].expand(-1, len(cols))))
                groups += 1
        moved = (net.loc != loc0).cpu().numpy()
        rec.ticks.append({"step": int(net.global_step) + 1, "dead": int(dead.sum()),
                          "dead_matches_statistics": bool(np.array_equal(dead, dead_by_stats)),
                          "dead_dims": int(dead.any(1).sum()), "groups": groups, "chained": chained,
                          "mass_rel_err": worst, "weights_copied": copied,
                          "moved_only_dead": bool(np.array_equal(moved, dead) or not (moved & ~dead).any()),
                          "max_active_min": float(mx.min().item()), "threshold": thr,
                          "sum_active_rows": float(sm.sum().item() / A)})
        return out
    net.resample_now = resample_now


def launch(family, seed, max_iterations, worker_over=None, net_over=None, frozen=False, max_samples=None, particles=35,
           quiet=True):
    """One training run through the real launcher.  Returns (worker, Record)."""
    from pfpn_amd import main
    setting, env = SETTINGS[family]
    st = main.parse(["--env", env, "--seed", str(seed), "--setting", setting, "--particles", str(particles), "--train"])
    st.worker_opts = dict(st.worker_opts, **(worker_over or {}))
    st.network_opts = dict(st.network_opts, **(net_over or {}))
    if frozen:                      # random-init policy that never moves: no learning rate, no resampling
        st.worker_opts.update(lr_actor=0.0, lr_critic=0.0)
        st.network_opts.update(resample_interval=1 << 40)
    st.max_iterations = max_iterations
    if max_samples is not None:
        st.max_samples = max_samples
    st.checkpoint_dir = None
    st.log_dir = None
    model = main.build(st)
    rec = Record()
    plain_build = model.build_worker

    def build_worker(*a, **kw):
        worker = plain_build(*a, **kw)
        net = worker.global_net
        if getattr(net, "max_active", None) is not None:
            _tick_probe(net, rec)
        plain_after = worker.after_episode

        def after_episode(sess, info):
            if not worker.request_stop:                      # (the episode a stop request cut short is not a return)
                rec.episodes.append((int(worker.global_step), int(worker.total_samples), int(worker.episode_step),
                                     float(worker.total_reward)))
            plain_after(sess, info)
        worker.after_episode = after_episode
        if quiet:
            worker.log_interval_time = 1e9
        rec.grid = net.loc.clone() if hasattr(net, "loc") else None
        return worker
    model.build_worker = build_worker
    worker = model.start({"worker": []})
    rec.seconds = time.perf_counter() - rec.t0
    net = worker.global_net
    if rec.grid is not None:
        rec.grid_shift = float((net.loc - rec.grid).abs().max().item())
        rec.grid_moved = int((net.loc != rec.grid).sum().item())
    rec.finite = all(np.isfinite(v.detach().cpu().numpy()).all() for v in net.variables().values())
    return worker, rec


def rollout_batch(worker, rows):
    """`rows` more environment steps with the worker's CURRENT policy through its own `run()` (no learner step):
    the learner inputs of one big DPPO batch -- real states, sampled actions, the policy's own log-probs and values,
    GAE advantages -- from a trained net."""
    net, env = worker.global_net, worker.env
    S, A = net.S, net.n_actions
    out = {"state": np.zeros((rows, S), np.float32), "action": np.zeros((rows, A), np.float32),
           "value": np.zeros(rows, np.float32), "log_prob": np.zeros(rows, np.float32),
           "advantage": np.zeros(rows, np.float32)}
    reward = np.zeros(rows, np.float32)
    s, lo, i = env.reset(), 0, 0

    def close(hi, boot):
        v = np.append(out["value"][lo:hi], np.float32(boot))
        out["advantage"][lo:hi] = net.generalized_advantage_estimate(reward[lo:hi], v)
    while i < rows:
        a, lp, v = net.run(None, s)
        out["state"][i], out["action"][i], out["log_prob"][i], out["value"][i] = s, a, lp, v
        s, r, over, info = env.step(np.clip(a, env.action_space.low, env.action_space.high))
        reward[i] = r
        i += 1
        if over or i == rows:
            true_end = over and not info.get("TimeLimit.truncated", False)
            close(i, 0.0 if true_end else net.evaluate(None, s))
            lo = i
            if over:
                s = env.reset()
    return out
