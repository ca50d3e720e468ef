"""The oracle against golden OUTPUT vectors evaluated from the reference's own saved training graph.

tests/golden/graph_*.npz were produced in the build container by `tests/golden/make_graph_golden.py`: a NumPy GraphDef
interpreter (`tests/golden/graphdef_eval.py`) walked `ckpt_DeepMimicWalk-v0/.../model.ckpt-78000.meta` node by node --
the reference's forward ops, the gradient sub-graph `tf.gradients` emitted (`optimizer/optimizer/grads/gradients/...`),
`optimizer/global_norm/*`, `optimizer/clip_by_global_norm/*`, `Adam/update_global_net/*/ApplyAdam`, the worker's
`state_normalizer/update/*`, `resample/{Max,Sum}` and the `cond/...` resample sub-graph.  The expected tensors therefore
come from the reference's dataflow, not from this repo's reading of its Python.  Node names are cited per check.

Tolerance: both sides are fp32 with different summation orders (oracle: NumPy pairwise / BLAS; graph evaluator: NumPy
per-op), so tensors are compared with `atol = RTOL * max|ref|` (RTOL = 2e-5; the generator measured <= 7.6e-6 everywhere,
tests/golden/graph_golden_report.txt).  Integer outputs (sampled particle, resample sources, dead set) are bit-exact.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import fixture_inputs as fi   # noqa: E402

F = np.float32
RTOL = 2e-5
STEP_CASES = ["step_b32", "step_b256", "step_b64_w2", "resample_b32", "resample_b32_w2", "guard_b32",
              "step_b4096", "step_b512_w8"]      # round 3: the bench's batch (config 2) and eight 512-row shards (config 4)


def load(tag):
    return np.load(os.path.join(HERE, "golden", "graph_%s.npz" % tag))


def close(got, ref, name, rtol=RTOL, floor=0.0):
    """|got - ref| <= rtol * max(max|ref|, floor).  `floor`: the magnitude of the TERMS of a cancelling sum (the PPO
    policy loss is the mean of B signed O(1) surrogates and can land near 0)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(got)), name
    assert np.array_equal(got[~fin], ref[~fin]), name          # -inf stays -inf
    scale = max(np.abs(ref[fin]).max() if fin.any() else 1.0, floor)
    err = np.abs(got[fin] - ref[fin]).max() if fin.any() else 0.0
    assert err <= rtol * scale + 1e-30, "%s: max abs err %.3g vs scale %.3g" % (name, err, scale)


def close_big(got, fx, key, name, rtol=RTOL):
    """A tensor stored as [::ROW_STRIDE] rows under `<kind>:<variable>` + float64 checksums (sum, sum of squares,
    sum of |x|) under `<kind>_sum:<variable>`."""
    kind, var = key.rsplit(":", 1)
    close(fi.subsample(var, got), fx[key], name, rtol)
    sums = fx[kind + "_sum:" + var]
    mine = fi.summarize(got)
    assert abs(mine[1] - sums[1]) <= 4 * rtol * sums[1] + 1e-30, name + " (sum of squares)"
    assert abs(mine[2] - sums[2]) <= 4 * rtol * sums[2] + 1e-30, name + " (sum of |x|)"


def oracle_state(oracle, fx):
    B, W, step, train_flag, seed = [int(v) for v in fx["meta"]]
    init = fi.initial_state(seed, step)
    st = oracle.DPPOState(np.random.default_rng(0))
    for k in fi.TRAINABLES:
        st.P[k], st.m[k], st.v[k] = init["P"][k].copy(), init["M"][k].copy(), init["V"][k].copy()
    st.norm_mean, st.norm_std = init["mean"].copy(), init["std"].copy()
    st.b1p, st.b2p = F(init["b1p"]), F(init["b2p"])
    st.global_step, st.train_flag = step, train_flag
    return st, B, W


def shard(fx, w):
    return fi.batch_of(fx, w)         # (compact fixtures recreate the batch from the hash streams)


@pytest.mark.parametrize("tag", STEP_CASES)
def test_worker_phase(oracle, tag):
    """forward, losses, 14 gradients, global norm, local normaliser update of every worker shard."""
    fx = load(tag)
    st, B, W = oracle_state(oracle, fx)
    for w in range(W):
        b = shard(fx, w)
        (loss, pl, vl), G, aux = oracle.dppo_grads(st.P, st.norm_mean, st.norm_std, b, fi.A, fi.N)
        pre = "out%d:" % w
        close(aux["logp"], fx[pre + "logp"], "global_net/policy_loss/pi_target/Sum_1")
        _, p = oracle.mixture_log_prob(aux["logits"], b["action"], st.P["actor/samples"], st.P["actor/samples_std"])
        close(p[::fi.P_ROW_STRIDE] if len(fx[pre + "p"]) != len(p) else p, fx[pre + "p"],
              "global_net/policy_loss/pi_target/Sum")
        close(aux["value"], fx[pre + "value"], "global_net/critic/Squeeze")
        close(loss, fx[pre + "loss"], "global_net/loss/add", floor=1.0)
        close(pl, fx[pre + "policy_loss"], "global_net/policy_loss/Neg", floor=1.0)
        close(vl, fx[pre + "value_loss"], "global_net/value_loss/mul")
        for k in fi.TRAINABLES:
            close_big(G[k], fx, pre + "grad:" + k, "optimizer/optimizer/grads/gradients -> " + k)
        _, norm = oracle.clip_by_global_norm([G[k] for k in fi.TRAINABLES], 1.0)
        close(norm, fx[pre + "global_norm"], "optimizer/global_norm/global_norm")
        m, s = oracle.state_norm_update(b["state"], st.norm_mean, st.norm_std, st.global_step)
        close(m, fx[pre + "ema_mean"], "worker_%d/state_normalizer/update/add_2" % w)
        close(s, fx[pre + "ema_std"], "worker_%d/state_normalizer/update/Maximum" % w)
    if tag == "guard_b32":      # the rows whose mixture probability underflowed
        assert np.isneginf(fx["out0:logp"][[3, 17]]).all() and np.isfinite(fx["out0:loss"])


def run_oracle_step(oracle, fx):
    st, B, W = oracle_state(oracle, fx)
    shards = [shard(fx, w) for w in range(W)]
    # the pushed activation statistics differ per worker in the fixtures: feed their mean as the global copy
    mx = fx["in0:max_active"].astype(F).copy()
    sm = fx["in0:sum_active"].astype(F).copy()
    for w in range(1, W):
        mx, sm = mx + fx["in%d:max_active" % w], sm + fx["in%d:sum_active" % w]
    st.max_active, st.sum_active = (mx / F(W)).astype(F), (sm / F(W)).astype(F)
    kw = {}
    if bool(fx["fired"]):
        kw = {"u_resample": fx["resample:u_resample"], "u_noise": fx["resample:u_noise_slot"]}
    oracle.dppo_learner_step(st, shards, **kw)
    return st


@pytest.mark.parametrize("tag", STEP_CASES)
def test_chief_phase(oracle, tag):
    """mean over workers -> Assign statistics -> 14 x ApplyAdam -> powers -> step -> train_flag -> cond resample."""
    fx = load(tag)
    st = run_oracle_step(oracle, fx)
    for k in fi.TRAINABLES:
        close_big(st.P[k], fx, "P:" + k, "Adam/update_global_net/%s/ApplyAdam (var)" % k)
        close_big(st.m[k], fx, "M:" + k, "Adam/update_global_net/%s/ApplyAdam (m)" % k)
        close_big(st.v[k], fx, "V:" + k, "Adam/update_global_net/%s/ApplyAdam (v)" % k)
    close(st.norm_mean, fx["post:state_normalizer/mean"], "Assign")
    close(st.norm_std, fx["post:state_normalizer/std"], "Assign_1")
    close(st.max_active, fx["post:max_active_degree"], "Assign_2 / cond/Assign")
    close(st.sum_active, fx["post:sum_active_degree"], "Assign_3 / cond/Assign_1")
    assert F(st.b1p) == fx["post:beta1_power"] and F(st.b2p) == fx["post:beta2_power"]       # Adam/mul, Adam/mul_1
    assert st.global_step == int(fx["post:global_step"])                                       # Adam (AssignAdd)
    assert st.train_flag == int(fx["post:resample/train_flag"])                                # AssignAdd / cond/Assign_2


@pytest.mark.parametrize("tag", ["resample_b32", "resample_b32_w2"])
def test_resample_subgraph(oracle, tag):
    """cond/...: dead set (`cond/Where`), source columns (`cond/add_2`), jittered locations (`cond/add_4`), clipped
    log-std (`cond/clip_by_value`), split bias (`cond/sub_1`) -- bit-exact indices, fp32 values."""
    fx = load(tag)
    st, B, W = oracle_state(oracle, fx)
    # state right before the cond: run the step without the tick, then call the resampler alone
    st.resample_interval = 1 << 30
    shards = [shard(fx, w) for w in range(W)]
    mx = sum(fx["in%d:max_active" % w].astype(F) for w in range(W)) / F(W)
    sm = sum(fx["in%d:sum_active" % w].astype(F) for w in range(W)) / F(W)
    st.max_active, st.sum_active = mx.astype(F), sm.astype(F)
    oracle.dppo_learner_step(st, shards)
    loc, logstd, bias, weight, info = oracle.resample_particles(
        st.P["actor/samples"], st.P["actor/samples_std"], st.P["actor/fc_policy/bias"], st.P["actor/fc_policy/weight"],
        st.max_active, st.sum_active, fx["resample:u_resample"], fx["resample:u_noise_slot"])
    inv = fx["resample:invalid"]                                  # [M, 2] row-major (cond/Where -> cond/Cast)
    dead = np.argwhere(info["dead"])
    assert np.array_equal(dead, inv)
    assert np.array_equal(info["src"], fx["resample:cand"])       # every slot's draw (cond/categorical/Multinomial)
    a, j = inv[:, 0], inv[:, 1]
    assert np.array_equal(fi.N * a + info["src"][a, j], fx["resample:tc"])     # cond/add_2
    assert np.array_equal(fi.N * a + j, fx["resample:c"])                      # cond/add_1
    close(loc[a, j], fx["resample:loc_new"], "cond/add_4")
    close(logstd[a, j], fx["resample:logstd_new"], "cond/clip_by_value")
    close(bias[fi.N * a + j], fx["resample:bias_new"], "cond/sub_1")
    # some dead particle is itself a source (delta = 1) and some source is cloned more than once
    tc, c = fx["resample:tc"], fx["resample:c"]
    assert np.isin(tc, c).any() and (np.bincount(tc).max() > 1)


def test_rollout_subgraph(oracle):
    """worker_0's sampler (`worker_0/action/*`), its log-prob (`worker_0/action/action_logp/Sum_1`), the critic and the
    activation statistics (`worker_0/resample/{Maximum,add}`); Multinomial / RandomStandardNormal outputs are inputs."""
    fx = load("rollout_e32")
    E, seed = [int(v) for v in fx["meta"]]
    init = fi.initial_state(seed, 100)
    P = init["P"]
    x = oracle.state_normalize(fx["state"], init["mean"], init["std"], 5.0)
    logits, _ = oracle.actor_forward(P, x, fi.A, fi.N)
    close(logits, fx["logits"], "worker_0/actor/Reshape")
    eps_sel = np.take_along_axis(fx["eps"], fx["k"][..., None].astype(np.int64), axis=-1)[..., 0]
    k, a = oracle.sample_plain(logits, P["actor/samples"], P["actor/samples_std"], fx["u_cat"], eps_sel)
    assert np.array_equal(k, fx["k"])
    close(a, fx["action"], "worker_0/action/Squeeze")
    lp, _ = oracle.mixture_log_prob(logits, a, P["actor/samples"], P["actor/samples_std"])
    close(lp, fx["logp"], "worker_0/action/action_logp/Sum_1")
    v, _ = oracle.critic_forward(P, x)
    close(v, fx["value"], "worker_0/critic/Squeeze")
    mx, sm = oracle.active_stats(logits, fx["max_active_in"], fx["sum_active_in"])
    close(mx, fx["max_active_out"], "worker_0/resample/Maximum")
    close(sm, fx["sum_active_out"], "worker_0/resample/add")
