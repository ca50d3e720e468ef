#!/usr/bin/env python3
"""Golden OUTPUT vectors evaluated from the reference's own saved training graph.

Build container only (reads /root/reference/ckpt_DeepMimicWalk-v0/.../model.ckpt-78000.meta; writes
tests/golden/graph_*.npz).  `graphdef_eval.Evaluator` interprets the GraphDef node by node, so every fetched tensor
below went through the reference's dataflow -- its forward ops, the gradient sub-graph `tf.gradients` emitted, its
`clip_by_global_norm`, `ApplyAdam` and `cond/...` resample nodes -- and NOT through oracle/pfpn_oracle.py.  The oracle
is only compared against the results at the end (the report printed here is pasted into tests/golden/README.md).

One learner step in the reference is two session runs (sync_model.py:62-157): a worker computes its clipped gradients
and updates its local statistics ("phase 1"), the chief takes the accumulated means, assigns the statistics, applies
Adam, ticks the resample counter and -- every 368 applies -- runs the resample cond ("phase 2").
Random ops are fed: Multinomial with the index the oracle's stated TF-CPU rule yields on the GRAPH's own logits,
RandomStandardNormal / RandomUniform with hash-generated noise stored in the fixture.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import fixture_inputs as fi          # noqa: E402
import graphdef_eval as ge           # noqa: E402
import make_meta_golden as wire      # noqa: E402
from oracle import pfpn_oracle as orc   # noqa: E402  (comparison only)

F = np.float32
G = "global_net/"
REPORT = []


def grad_node_names(nodes):
    """raw gradient i = data input of optimizer/global_norm/L2Loss_i; clipped = clip_by_global_norm/_i."""
    raw = []
    for i in range(14):
        l2 = nodes["optimizer/global_norm/L2Loss" + ("_%d" % i if i else "")]
        raw.append(l2["input"][0])
    clipped = ["optimizer/clip_by_global_norm/optimizer/clip_by_global_norm/_%d" % i for i in range(14)]
    return raw, clipped


def rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-300))


def report(tag, name, got, ref):
    e = rel_err(got, ref)
    REPORT.append((tag, name, e))
    return e


def fresh_vars(seed, step, train_flag, workers=("worker_0",)):
    init = fi.initial_state(seed, step)
    P, M, V, mean, std = init["P"], init["M"], init["V"], init["mean"], init["std"]
    v = {}
    for k in fi.TRAINABLES:
        v[G + k] = P[k].copy()
        v[G + k + "/Adam"] = M[k].copy()
        v[G + k + "/Adam_1"] = V[k].copy()
    v[G + "state_normalizer/mean"], v[G + "state_normalizer/std"] = mean, std
    v[G + "max_active_degree"] = np.zeros((fi.A, fi.N), F)
    v[G + "sum_active_degree"] = np.zeros((fi.A, fi.N), F)
    v[G + "resample/train_flag"] = np.asarray(train_flag, F)
    v["beta1_power"], v["beta2_power"] = np.asarray(init["b1p"], F), np.asarray(init["b2p"], F)
    v["step/global_step"] = np.asarray(step, np.int64)
    for w in workers:       # pull_weights: the local nets start as copies of the global one
        for k in fi.TRAINABLES:
            v[w + "/" + k] = P[k].copy()
        v[w + "/state_normalizer/mean"], v[w + "/state_normalizer/std"] = mean.copy(), std.copy()
        v[w + "/max_active_degree"] = np.zeros((fi.A, fi.N), F)
        v[w + "/sum_active_degree"] = np.zeros((fi.A, fi.N), F)
    return v


def oracle_state(v, step, train_flag):
    st = orc.DPPOState(np.random.default_rng(0))
    for k in fi.TRAINABLES:
        st.P[k] = v[G + k].copy()
        st.m[k] = v[G + k + "/Adam"].copy()
        st.v[k] = v[G + k + "/Adam_1"].copy()
    st.norm_mean, st.norm_std = v[G + "state_normalizer/mean"].copy(), v[G + "state_normalizer/std"].copy()
    st.b1p, st.b2p = F(v["beta1_power"]), F(v["beta2_power"])
    st.global_step, st.train_flag = step, train_flag
    return st


def phase1(ev, nodes, batch):
    """Worker side: losses, log-prob, value, raw + clipped gradients on global_net's placeholders; the local
    state-normaliser update (its placeholder is the local net's `state`)."""
    raw, clipped = grad_node_names(nodes)
    feeds = {G + "state": batch["state"], G + "action_hist": batch["action"], G + "value_running": batch["value"],
             G + "pi_running": batch["log_prob"], G + "advantage": batch["advantage"]}
    fetch = [G + "policy_loss/pi_target/Sum_1", G + "policy_loss/pi_target/Sum", G + "critic/Squeeze",
             G + "loss/add", G + "policy_loss/Neg", G + "value_loss/mul", "optimizer/global_norm/global_norm",
             G + "state_normalizer/clip_state/clip_by_value", G + "actor/Reshape"] + raw + clipped
    out = ev.run(fetch, feeds)
    res = {"logp": out[0], "p": out[1], "value": out[2], "loss": out[3], "policy_loss": out[4],
           "value_loss": out[5], "global_norm": out[6], "x": out[7], "logits": out[8],
           "grads": dict(zip(fi.TRAINABLES, out[9:23])), "clipped": dict(zip(fi.TRAINABLES, out[23:37]))}
    return res


def ema_update(ev, worker, state):
    ev.run([worker + "/state_normalizer/update/Assign", worker + "/state_normalizer/update/Assign_1"],
           {worker + "/state": state})
    return ev.vars[worker + "/state_normalizer/mean"].copy(), ev.vars[worker + "/state_normalizer/std"].copy()


def phase2(ev, nodes, gmean, stats, resample_feeds=None):
    """Chief side.  gmean: 14 averaged clipped gradients; stats: averaged (mean, std, max_active, sum_active)."""
    feeds = {}
    for i, k in enumerate(fi.TRAINABLES):
        feeds["optimizer/sync_replicas/AccumulatorTakeGradient" + ("_%d" % i if i else "")] = gmean[k]
    for i, s in enumerate(stats):
        feeds["optimizer/sync_replicas/AccumulatorTakeGradient_%d" % (14 + i)] = s
    cache = {}
    ev.run(["Adam", "GreaterEqual"], feeds, cache)           # Assign..Assign_3 -> 14 x ApplyAdam -> powers -> step -> train_flag
    fired = bool(ev.value("GreaterEqual"))
    info = {"fired": fired}
    if fired:
        # the graph orders the reads of max/sum_active and the zeroing Assigns only through scheduling;
        # the consistent order (reads first) is made explicit here by fetching the pure values before group_deps
        log_avg = ev.run(["cond/Log", "cond/Cast"], feeds, cache)
        info["log_avg"], info["invalid"] = log_avg
        u_res, u01 = resample_feeds
        cand = orc.multinomial_index(info["log_avg"], u_res)          # [A, N] draws, the oracle's stated rule
        M = info["invalid"].shape[0]
        feeds["cond/categorical/Multinomial"] = cand.astype(np.int64)
        feeds["cond/random_uniform/RandomUniform"] = u01[:M].astype(F)
        pure = ["cond/add_4", "cond/clip_by_value", "cond/sub_1", "cond/Reshape_4", "cond/add_2", "cond/add_1",
                "cond/random_uniform", "cond/GatherV2_8"]
        vals = ev.run(pure, feeds, cache)
        info.update(dict(zip(["loc_new", "logstd_new", "bias_new", "weight_cols", "tc", "c", "noise_u", "log_cnt"],
                             vals)))
        info["cand"] = cand
    ev.run(["group_deps"], feeds, cache)
    return info


def save_vars(v, prefix=G):
    out = {}
    for k in fi.TRAINABLES:
        for suf, tag in (("", "P"), ("/Adam", "M"), ("/Adam_1", "V")):
            x = v[prefix + k + suf]
            out["%s:%s" % (tag, k)] = fi.subsample(k, x)
            out["%s_sum:%s" % (tag, k)] = fi.summarize(x)
    return out


def scenario(nodes, tag, B, W, step, train_flag, seed, dead_pattern=False, guard=False, compact=False):
    """compact: the big scenarios (the bench's 4096-row batch, eight 512-row shards) keep only what cannot be recreated:
    the batches come back from fixture_inputs.make_batch(B, seed * 10 + w) (fixture_inputs.batch_of), only the old
    log-prob (a graph output) is stored, and the [B, A] mixture probabilities are kept for every 16th row."""
    workers = tuple("worker_%d" % i for i in range(W))
    v = fresh_vars(seed, step, train_flag, workers)
    ev = ge.Evaluator(nodes, v)
    st = oracle_state(v, step, train_flag)
    fx = {"meta": np.asarray([B, W, step, train_flag, seed], np.int64)}
    shards, clipped_all, stats_all = [], [], []
    # stats pushed by the workers: their local max/sum_active (rollout side fills them; given here)
    for w in range(W):
        if dead_pattern:
            mx = (fi.hash_uniform((fi.A, fi.N), seed * 1000 + 50 + w) * 0.02).astype(F)
            dead = fi.hash_uniform((fi.A, fi.N), seed * 1000 + 60) < 0.12          # same slots on every worker
            dead[3, :] = False                                                       # a dim with no dead particle
            dead[5, :30] = True                                                      # a dim that is mostly dead
            mx[dead] = (fi.hash_uniform((fi.A, fi.N), seed * 1000 + 70 + w)[dead] * 0.001).astype(F)
            sm = (mx * (50 + 100 * fi.hash_uniform((fi.A, fi.N), seed * 1000 + 80 + w))).astype(F)
            sm[dead] *= F(0.01)
            # a dead particle with most of its dimension's activation mass: it is drawn as a source although it is
            # dead itself (the graph's delta = 1 case, cond/map/while/Equal), and it is cloned several times
            dead[7, 2] = True
            dead[7, 20:26] = True
            mx[7, 2] = F(0.0005)
            mx[7, 20:26] = F(0.0003)
            sm[7, 2] = F(50.0)
        else:
            mx = (fi.hash_uniform((fi.A, fi.N), seed * 1000 + 50 + w) * 0.5).astype(F)
            sm = (mx * 40).astype(F)
        v[workers[w] + "/max_active_degree"][...] = mx
        v[workers[w] + "/sum_active_degree"][...] = sm
    for w in range(W):
        b = fi.make_batch_off_kinks(B, seed * 10 + w, seed, step) if compact else fi.make_batch(B, seed * 10 + w)
        # old log-prob = the graph's own log-prob of the action, perturbed so ratios straddle the clip range
        b["log_prob"] = np.zeros(B, F)
        if guard:       # rows whose mixture probability underflows to 0: log p = -inf forward, `foo` zeroes 0/0 backward
            b["action"][3, 7] = F(60.0)
            b["action"][17, 0] = F(-45.0)
        r0 = phase1(ev, nodes, b)
        b["log_prob"] = (r0["logp"] + fi.hash_normal((B,), seed * 10 + w + 5) * 0.15).astype(F)
        if compact:
            # PPO's surrogate has a kink in the ratio at 1 -+ 0.2: among thousands of rows one ALWAYS sits within the fp32
            # error of a log-prob of size ~1e2 (~1e-4 in the ratio) of a bound, and the implementation that rounds it to
            # the other side moves every actor gradient by that row's whole contribution.  Such rows get ratio 1.
            ratio = np.exp(r0["logp"].astype(np.float64) - b["log_prob"].astype(np.float64))
            near = (np.abs(ratio - 0.8) < 2e-3) | (np.abs(ratio - 1.2) < 2e-3)
            b["log_prob"][near] = r0["logp"][near]
        if guard:
            assert np.isneginf(r0["logp"][[3, 17]]).all()
            b["log_prob"][[3, 17]] = F(-200.0)
        r = phase1(ev, nodes, b)
        shards.append(b)
        for k in (("log_prob",) if compact else ("state", "action", "value", "advantage", "log_prob")):
            fx["in%d:%s" % (w, k)] = b[k]
        for k in ("logp", "p", "value", "loss", "policy_loss", "value_loss", "global_norm"):
            fx["out%d:%s" % (w, k)] = r[k][::fi.P_ROW_STRIDE] if (compact and k == "p") else r[k]
        for k in fi.TRAINABLES:
            fx["out%d:grad:%s" % (w, k)] = fi.subsample(k, r["grads"][k])
            fx["out%d:grad_sum:%s" % (w, k)] = fi.summarize(r["grads"][k])
        # oracle on the same inputs
        (l, pl, vl), Go, aux = orc.dppo_grads(st.P, st.norm_mean, st.norm_std, b, fi.A, fi.N)
        report(tag, "w%d logp" % w, aux["logp"], r["logp"])
        report(tag, "w%d value" % w, aux["value"], r["value"])
        report(tag, "w%d loss" % w, [l, pl, vl], [r["loss"], r["policy_loss"], r["value_loss"]])
        for k in fi.TRAINABLES:
            report(tag, "w%d grad %s" % (w, k), Go[k], r["grads"][k])
        cl, norm = orc.clip_by_global_norm([Go[k] for k in fi.TRAINABLES], 1.0)
        report(tag, "w%d global_norm" % w, norm, r["global_norm"])
        for k, c in zip(fi.TRAINABLES, cl):
            report(tag, "w%d clipped %s" % (w, k), c, r["clipped"][k])
        # local EMA (every worker_i owns an identical update sub-graph on its own copies)
        lm, ls = ema_update(ev, workers[w], b["state"])
        om, os_ = orc.state_norm_update(b["state"], st.norm_mean, st.norm_std, step)
        report(tag, "w%d ema mean" % w, om, lm)
        report(tag, "w%d ema std" % w, os_, ls)
        fx["out%d:ema_mean" % w], fx["out%d:ema_std" % w] = lm, ls
        clipped_all.append(r["clipped"])
        stats_all.append([lm, ls, v[workers[w] + "/max_active_degree"].copy(), v[workers[w] + "/sum_active_degree"].copy()])
        fx["in%d:max_active" % w], fx["in%d:sum_active" % w] = stats_all[-1][2], stats_all[-1][3]
    # accumulator means (ConditionalAccumulator: sum of applied tensors / count)
    gmean = {}
    for k in fi.TRAINABLES:
        s = clipped_all[0][k].copy()
        for w in range(1, W):
            s = s + clipped_all[w][k]
        gmean[k] = (s / F(W)).astype(F)
    stats = []
    for i in range(4):
        s = stats_all[0][i].copy()
        for w in range(1, W):
            s = s + stats_all[w][i]
        stats.append((s / F(W)).astype(F))
    u_res = fi.hash_uniform((fi.A, fi.N), seed * 1000 + 1)
    u01 = fi.hash_uniform((fi.A * fi.N,), seed * 1000 + 2).astype(F)
    info = phase2(ev, nodes, gmean, stats, (u_res, u01))
    fx.update(save_vars(v))
    for k in ("state_normalizer/mean", "state_normalizer/std", "max_active_degree", "sum_active_degree",
              "resample/train_flag"):
        fx["post:" + k] = v[G + k].copy()
    fx["post:beta1_power"], fx["post:beta2_power"] = v["beta1_power"].copy(), v["beta2_power"].copy()
    fx["post:global_step"] = v["step/global_step"].copy()
    fx["fired"] = np.asarray(info["fired"])
    # oracle: the same step
    u_noise = None
    if info["fired"]:
        inv = info["invalid"]
        fx["resample:u_resample"] = u_res
        fx["resample:u01"] = u01
        fx["resample:invalid"] = inv
        fx["resample:cand"] = info["cand"].astype(np.int32)
        fx["resample:log_avg"] = info["log_avg"]
        fx["resample:tc"], fx["resample:c"] = info["tc"], info["c"]
        fx["resample:loc_new"], fx["resample:logstd_new"], fx["resample:bias_new"] = \
            info["loc_new"], info["logstd_new"], info["bias_new"]
        u_noise = np.zeros((fi.A, fi.N), F)
        u_noise[inv[:, 0], inv[:, 1]] = info["noise_u"]        # graph: U*2 + (-1) by rank; oracle: by slot
        fx["resample:u_noise_slot"] = u_noise
    st.max_active, st.sum_active = None, None
    # oracle's W-shard step with the same pushed statistics
    st.max_active = stats_all[0][2]   # placeholders; dppo_learner_step averages st.* over W (identical copies)
    _oracle_step(st, shards, stats_all, u_res, u_noise)
    for k in fi.TRAINABLES:
        report(tag, "post P %s" % k, st.P[k], v[G + k])
        report(tag, "post M %s" % k, st.m[k], v[G + k + "/Adam"])
        report(tag, "post V %s" % k, st.v[k], v[G + k + "/Adam_1"])
    report(tag, "post norm_mean", st.norm_mean, v[G + "state_normalizer/mean"])
    report(tag, "post norm_std", st.norm_std, v[G + "state_normalizer/std"])
    report(tag, "post max_active", st.max_active, v[G + "max_active_degree"])
    report(tag, "post sum_active", st.sum_active, v[G + "sum_active_degree"])
    assert F(st.b1p) == v["beta1_power"] and F(st.b2p) == v["beta2_power"], "beta powers differ"
    assert st.global_step == int(v["step/global_step"]) and st.train_flag == int(v[G + "resample/train_flag"])
    np.savez_compressed(os.path.join(HERE, "graph_%s.npz" % tag), **fx)
    return fx


def _oracle_step(st, shards, stats_all, u_res, u_noise):
    """oracle.dppo_learner_step with per-worker pushed max/sum_active (the oracle's own version pushes the global
    copies; workers differ here to exercise the mean)."""
    W = len(shards)
    gsum, ssum = None, None
    for b, stt in zip(shards, stats_all):
        lm, ls = orc.state_norm_update(b["state"], st.norm_mean, st.norm_std, st.global_step)
        _, Go, _ = orc.dppo_grads(st.P, st.norm_mean, st.norm_std, b, st.A, st.N)
        cl, _ = orc.clip_by_global_norm([Go[k] for k in fi.TRAINABLES], 1.0)
        s4 = [lm, ls, stt[2], stt[3]]
        if gsum is None:
            gsum, ssum = [c.copy() for c in cl], [s.copy() for s in s4]
        else:
            gsum = [(a + c).astype(F) for a, c in zip(gsum, cl)]
            ssum = [(a + s).astype(F) for a, s in zip(ssum, s4)]
    st.norm_mean, st.norm_std, st.max_active, st.sum_active = [(s / F(W)).astype(F) for s in ssum]
    for k, g in zip(fi.TRAINABLES, gsum):
        st.P[k], st.m[k], st.v[k] = orc.tf_adam(st.P[k], (g / F(W)).astype(F), st.m[k], st.v[k], st.b1p, st.b2p, 1e-4)
    st.b1p, st.b2p = F(st.b1p * F(0.9)), F(st.b2p * F(0.999))
    st.global_step += 1
    st.train_flag += 1
    if st.train_flag >= st.resample_interval:
        (st.P["actor/samples"], st.P["actor/samples_std"], st.P["actor/fc_policy/bias"],
         st.P["actor/fc_policy/weight"], _) = orc.resample_particles(
            st.P["actor/samples"], st.P["actor/samples_std"], st.P["actor/fc_policy/bias"],
            st.P["actor/fc_policy/weight"], st.max_active, st.sum_active, u_res, u_noise)
        st.max_active = np.zeros_like(st.max_active)
        st.sum_active = np.zeros_like(st.sum_active)
        st.train_flag = 0


def rollout(nodes, tag, E, seed):
    """worker_0's rollout sub-graph: sampled action, its log-prob, value, activation statistics (a2c.py:346-365)."""
    v = fresh_vars(seed, 100, 0)
    v["worker_0/max_active_degree"][...] = (fi.hash_uniform((fi.A, fi.N), seed + 500) * 0.3).astype(F)
    v["worker_0/sum_active_degree"][...] = (fi.hash_uniform((fi.A, fi.N), seed + 501) * 3).astype(F)
    mx0, sm0 = v["worker_0/max_active_degree"].copy(), v["worker_0/sum_active_degree"].copy()
    ev = ge.Evaluator(nodes, v)
    state = fi.make_batch(E, seed * 10 + 7)["state"]
    u_cat = fi.hash_uniform((E, fi.A), seed * 1000 + 3)
    eps = fi.hash_normal((E, fi.A, fi.N), seed * 1000 + 4).astype(F)
    feeds = {"worker_0/state": state}
    logits = ev.run(["worker_0/actor/Reshape"], feeds)[0]
    k = orc.multinomial_index(logits.reshape(E * fi.A, fi.N), u_cat.reshape(-1))
    mult = [n for n in nodes if n.startswith("worker_0/actor/Categorical/sample") and nodes[n]["op"] == "Multinomial"]
    rsn = [n for n in nodes if n.startswith("worker_0/actor/Normal/sample") and nodes[n]["op"] == "RandomStandardNormal"]
    assert len(mult) == 1 and len(rsn) == 1, (mult, rsn)
    feeds[mult[0]] = k.reshape(E * fi.A, 1).astype(np.int32)
    feeds[rsn[0]] = eps                           # sample_shape [E] + batch_shape [A, N] (utils.py:189)
    cache = {}
    act, logp, val = ev.run(["worker_0/action/Squeeze", "worker_0/action/action_logp/Sum_1", "worker_0/critic/Squeeze"],
                            feeds, cache)
    ev.run(["worker_0/resample/Assign", "worker_0/resample/Assign_1"], feeds, cache)
    mx1, sm1 = v["worker_0/max_active_degree"].copy(), v["worker_0/sum_active_degree"].copy()
    fx = {"state": state, "u_cat": u_cat, "eps": eps, "k": k.reshape(E, fi.A), "action": act.reshape(E, fi.A),
          "logp": np.asarray(logp).reshape(E), "value": np.asarray(val).reshape(E),
          "max_active_in": mx0, "sum_active_in": sm0, "max_active_out": mx1, "sum_active_out": sm1,
          "logits": logits, "meta": np.asarray([E, seed], np.int64)}
    # oracle
    P = {kk: v["worker_0/" + kk] for kk in fi.TRAINABLES}
    x = orc.state_normalize(state, v["worker_0/state_normalizer/mean"], v["worker_0/state_normalizer/std"], 5.0)
    lg, _ = orc.actor_forward(P, x, fi.A, fi.N)
    eps_sel = np.take_along_axis(eps, k.reshape(E, fi.A, 1).astype(np.int64), axis=-1)[..., 0]
    k2, a2 = orc.sample_plain(lg, P["actor/samples"], P["actor/samples_std"], u_cat, eps_sel)
    lp2, _ = orc.mixture_log_prob(lg, a2, P["actor/samples"], P["actor/samples_std"])
    v2, _ = orc.critic_forward(P, x)
    m2, s2 = orc.active_stats(lg, mx0, sm0)
    report(tag, "logits", lg, logits)
    REPORT.append((tag, "k mismatches", float((k2 != k.reshape(E, fi.A)).sum())))
    report(tag, "action", a2, fx["action"])
    report(tag, "logp", lp2, fx["logp"])
    report(tag, "value", v2, fx["value"])
    report(tag, "max_active", m2, mx1)
    report(tag, "sum_active", s2, sm1)
    np.savez_compressed(os.path.join(HERE, "graph_%s.npz" % tag), **fx)


def main():
    nodes, order = ge.load_graph(os.path.join(wire.REF, wire.META["walk"]))
    only = set(sys.argv[1:])                        # `make_graph_golden.py tag ...`: just these scenarios
    want = lambda tag: not only or tag in only
    if want("step_b32"): scenario(nodes, "step_b32", B=32, W=1, step=1234, train_flag=5, seed=1)
    if want("step_b256"): scenario(nodes, "step_b256", B=256, W=1, step=77, train_flag=100, seed=2)
    if want("step_b64_w2"): scenario(nodes, "step_b64_w2", B=64, W=2, step=500, train_flag=9, seed=3)
    if want("resample_b32"): scenario(nodes, "resample_b32", B=32, W=1, step=367, train_flag=367, seed=4, dead_pattern=True)
    if want("resample_b32_w2"):
        scenario(nodes, "resample_b32_w2", B=32, W=2, step=735, train_flag=367, seed=5, dead_pattern=True)
    if want("guard_b32"): scenario(nodes, "guard_b32", B=32, W=1, step=3, train_flag=0, seed=7, guard=True)
    # round 3: the bench's headline batch (BASELINE.json config 2) and the 8-GPU strong-scaling exchange (config 4's
    # shard size: eight workers x 512 rows, accumulator mean over 8) -- straight from the reference's graph
    if want("step_b4096"): scenario(nodes, "step_b4096", B=4096, W=1, step=4321, train_flag=50, seed=8, compact=True)
    if want("step_b512_w8"): scenario(nodes, "step_b512_w8", B=512, W=8, step=910, train_flag=120, seed=9, compact=True)
    if want("rollout_e32"): rollout(nodes, "rollout_e32", E=32, seed=6)
    worst = {}
    for tag, name, e in REPORT:
        key = (tag, name.split(" ")[1] if name.startswith("w") or name.startswith("post") else name)
        worst[key] = max(worst.get(key, 0.0), e)
    for tag, name, e in REPORT:
        print("%-18s %-40s %.3e" % (tag, name, e))
    print("max over everything:", max(e for t, n, e in REPORT if "mismatch" not in n))


if __name__ == "__main__":
    main()
