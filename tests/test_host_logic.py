"""CPU: host-side logic of the reference-shaped runtime -- replay Buffer, minibatch scheduling
(flat_train), PPO worker bookkeeping (bootstrap value, GAE segments, buffer_stamp), SAC worker
(not_terminal, warm-up), launcher flags.  The network is a duck-typed stub (the real one needs a GPU)."""
import numpy as np
import pytest

from pfpn_amd.envs import SyntheticDeepMimicEnv
from pfpn_amd.models import DPPOWorker, SyncSACWorker
from pfpn_amd.models.workers.ddpg import Buffer


class StubNet(object):
    """records what the workers feed; GAE via the oracle (checker) so numbers are the reference's."""
    def __init__(self, oracle, trainable=True, sac=False):
        self.trainable, self.sac, self.oracle = trainable, sac, oracle
        self.init_ops, self.train_ops = [], []
        self.global_step = 0
        self.trained = []
        self.rng = np.random.default_rng(0)
        self.evaluated = []

    def init(self):
        pass

    def reset_running_state(self):
        pass

    def run(self, sess, state, ops=None):
        a = self.rng.uniform(-1.5, 1.5, 36).astype(np.float32)
        return [a] if self.sac else [a, float(self.rng.normal()), float(np.sum(state[:3]))]

    def evaluate(self, sess, state):
        self.evaluated.append(np.asarray(state).copy())
        return 7.0

    def generalized_advantage_estimate(self, reward, value):
        return list(self.oracle.gae(reward, value, 0.95, 0.95 * 0.95))

    def train(self, sess, optimizer, ops, *args):
        self.trained.append([np.asarray(a).copy() for a in args])
        self.global_step += 1
        return (0.1, None, 0.05, 0.05), [self.global_step]


def make_worker(cls, net, **opts):
    w = cls(_env_wrapper=lambda name, render: SyntheticDeepMimicEnv(episode_length=50, seed=3),
            _network_wrapper=lambda env, trainable: net, _seed=5, n_workers=1, **opts)
    w.build_optimizer = lambda n: "opt"
    w.init()
    return w


def test_ring_buffer():
    b = Buffer(3)
    for i in range(5):
        b.append(i)
    assert len(b) == 3 and sorted(b.data) == [2, 3, 4] and b.pointer == 2
    b.clear()
    assert len(b) == 0


def test_ppo_worker_rollout_and_minibatches(oracle):
    net = StubNet(oracle)
    w = make_worker(DPPOWorker, net, lr_critic=1e-4, lr_actor=1e-4, norm_clip=1.0, opt_epochs=1, batch_size=32,
                    unroll_length=128)
    w.work(None, max_iterations=8)
    # 128 samples -> 4 minibatches of 32 per round, two rounds
    assert len(net.trained) == 8 and w.global_step == 8
    state, action, value, log_prob, adv = net.trained[0]
    assert state.shape == (32, 197) and action.shape == (32, 36) and adv.shape == (32,) and adv.dtype == np.float32
    # the unclipped action is stored (distributed_model.py:264-290)
    assert np.abs(np.concatenate([t[1] for t in net.trained])).max() > 1.0
    # one round's minibatches are a permutation of the 128 samples (no duplicates)
    rows = np.concatenate([t[0] for t in net.trained[:4]])
    assert len({r.tobytes() for r in rows}) == 128
    # episodes are 50 steps with a time-limit truncation -> bootstrap by evaluate(last_state), not 0
    assert len(net.evaluated) >= 2


class RecordingEnv(SyntheticDeepMimicEnv):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.log = []          # (reward, terminal, truncated)

    def step(self, a):
        s, r, t, info = super().step(a)
        self.log.append((r, t, bool(info.get("TimeLimit.truncated", False))))
        return s, r, t, info


def test_ppo_worker_gae_segments_match_reference_loop(oracle):
    """advantages = per-segment GAE over [buffer_stamp:] with bootstrap 0.0 on a true terminal and
    V(last_state) on a time-limit truncation or unroll boundary (workers/ppo.py:50-73)."""
    net = StubNet(oracle)
    env = RecordingEnv(episode_length=50, seed=3)
    w = DPPOWorker(_env_wrapper=lambda name, render: env, _network_wrapper=lambda e, t: net, _seed=5, n_workers=1,
                   lr_critic=1e-4, lr_actor=1e-4, norm_clip=1.0, opt_epochs=1, batch_size=None, unroll_length=120)
    w.build_optimizer = lambda n: "opt"
    w.init()
    class NoShuffle(object):                         # keep rollout order so rows can be matched
        def shuffle(self, ids):
            pass
    w.np_random = NoShuffle()
    w.work(None, max_iterations=1)
    state, action, value, log_prob, adv = net.trained[0]
    assert len(adv) == 120 and len(env.log) == 120
    expect, start = [], 0
    for t, (r, term, trunc) in enumerate(env.log):
        end_of_unroll = (t == 119)
        if term or end_of_unroll:
            boot = 0.0 if (term and not trunc) else 7.0
            rs = [e[0] for e in env.log[start:t + 1]]
            vs = list(value[start:t + 1]) + [boot]
            expect.extend(oracle.gae(rs, vs, 0.95, 0.95 * 0.95))
            start = t + 1
    np.testing.assert_array_equal(adv, np.asarray(expect, np.float32))


def test_sac_worker_not_terminal_and_replay(oracle):
    net = StubNet(oracle, sac=True)
    w = make_worker(SyncSACWorker, net, lr_critic=1e-4, lr_actor=1e-4, norm_clip=1, opt_epochs=None, batch_size=16,
                    unroll_length=1, buffer_capacity=1000, observations=0)
    w.work(None, max_iterations=40)
    assert w.separate_optimizer is True
    assert len(net.trained) == 40
    s, a, r, nt, s2 = net.trained[-1]
    assert s.shape == (16, 197) and s2.shape == (16, 197) and nt.shape == (16,)
    # time-limit truncation keeps not_terminal = 1 (ddpg.py:72)
    allnt = np.concatenate([t[3] for t in net.trained])
    assert set(np.unique(allnt)) <= {0.0, 1.0} and allnt.mean() > 0.9
    # no training until the buffer holds batch_size samples
    assert w.total_samples >= 16 + 40 - 1


def test_flat_train_epochs(oracle):
    net = StubNet(oracle)
    w = make_worker(DPPOWorker, net, lr_critic=1e-4, lr_actor=1e-4, norm_clip=1.0, opt_epochs=3, batch_size=20,
                    unroll_length=50)
    w.work(None, max_iterations=9)
    # 50 samples / batch 20 -> 3 minibatches (20,20,10) x 3 epochs
    assert [len(t[0]) for t in net.trained[:9]] == [20, 20, 10] * 3


def test_launcher_flags_and_names():
    from pfpn_amd import main, models, networks
    st = main.parse(["--env", "DeepMimicWalk-v0", "--seed", "34114", "--setting",
                     "settings.deepmimic.deepmimic_dppo_pfpn", "--particles", "35", "--train"])
    assert st.network == "ParticleFilteringClipPPONetwork" and st.worker == "DPPOWorker" and st.model == "SyncModel"
    assert st.network_opts["particles"] == 35 and st.train
    assert st.checkpoint_dir == "ckpt_DeepMimicWalk-v0/deepmimic_dppo_pfpn_particle35/34114"   # main.py:52
    for name in ("ParticleFilteringClipPPONetwork", "ParticleFilteringSACNetwork", "ParticleFilteringA2CNetwork"):
        assert hasattr(networks, name)
    for name in ("SyncModel", "DPPOWorker", "SyncSACWorker"):
        assert hasattr(models, name)


def test_out_of_scope_options_raise():
    from pfpn_amd.networks import ParticleFilteringClipPPONetwork
    base = dict(trainable=True, state_shape=[5], action_shape=[2], action_lower_bound=[-1, -1],
                action_upper_bound=[1, 1], common_net_shape=[], critic_net_shape=[8], actor_net_shape=[8])
    with pytest.raises(NotImplementedError):
        ParticleFilteringClipPPONetwork(normalize_value=True, **base)
    with pytest.raises(ValueError):                                 # a2c.py:390 asserts n >= resample
        ParticleFilteringClipPPONetwork(resample=51, **base)
    with pytest.raises(ValueError):                                 # fixed_sigma=True jitters with init_sigma (a2c.py:438-439)
        ParticleFilteringClipPPONetwork(resample=3, fixed_sigma=True, **base)
    with pytest.raises(NotImplementedError):
        ParticleFilteringClipPPONetwork(**dict(base, common_net_shape=[16]))


# --- A2C / V-trace workers: segment bookkeeping -------------------------------------------------
class StubA2CNet(StubNet):
    def run(self, sess, state, ops=None):
        a = self.rng.uniform(-1.5, 1.5, 36).astype(np.float32)
        return [a, float(np.sum(state[:3]))]                      # a2c.py:67-73: [action, value]


class StubVTraceNet(StubNet):
    def run(self, sess, state, ops=None):
        a = self.rng.uniform(-1.5, 1.5, 36).astype(np.float32)
        return [a, float(self.rng.normal())]                      # vtrace.py:184-192: [action, log_prob]


def test_a2c_worker_targets(oracle):
    from pfpn_amd.models import SyncA2CWorker
    net = StubA2CNet(oracle)
    w = make_worker(SyncA2CWorker, net, lr_critic=1e-4, lr_actor=1e-4, norm_clip=0.5, opt_epochs=1, batch_size=32,
                    unroll_length=32)
    w.work(None, max_iterations=3)
    assert len(net.trained) == 3
    state, action, value_target, adv = net.trained[0]
    assert state.shape == (32, 197) and action.shape == (32, 36) and value_target.shape == (32,) and adv.shape == (32,)
    # value_target = value + advantage (a2c.py:42-49); value = sum(state[:3]) by the stub; rows are shuffled together
    np.testing.assert_allclose(value_target - adv, state[:, :3].sum(1), rtol=1e-5, atol=1e-5)


def test_vtrace_worker_segments(oracle):
    from pfpn_amd.models import SyncVTraceWorker, IMPALAWorker
    assert IMPALAWorker is SyncVTraceWorker
    net = StubVTraceNet(oracle)
    # episodes are 50 steps, unroll 64 > 50: every segment ends at an episode end and is zero-padded to 64
    w = make_worker(SyncVTraceWorker, net, lr_critic=1e-4, lr_actor=1e-4, norm_clip=0.5, opt_epochs=1, batch_size=4,
                    unroll_length=64)
    w.work(None, max_iterations=2)
    assert len(net.trained) == 2
    state, action, reward, log_prob, not_terminal, seq_len = net.trained[0]
    assert list(seq_len) == [50, 50, 50, 50]
    assert state.shape == (4 * 64, 197) and log_prob.shape == (256,) and not_terminal.shape == (256,)
    s = state.reshape(4, 64, 197)
    assert np.all(s[:, 50:] == 0) and np.all(np.abs(s[:, :50]).sum(-1) > 0)          # padding behind every segment
    # the synthetic env ends by time limit: not terminal for the bootstrap (vtrace.py:48-51)
    assert np.all(not_terminal.reshape(4, 64)[:, :50] == 1.0)
    # shorter unroll: segments of exactly `unroll_length` rows, cut mid-episode
    net2 = StubVTraceNet(oracle)
    w2 = make_worker(SyncVTraceWorker, net2, lr_critic=1e-4, lr_actor=1e-4, norm_clip=0.5, opt_epochs=1, batch_size=3,
                     unroll_length=20)
    w2.work(None, max_iterations=1)
    st2, _, _, _, _, sl2 = net2.trained[0]
    assert list(sl2) == [20, 20, 10] and st2.shape == (60, 197)      # 20 + 20 + the 10 rows left of the 50-step episode
    assert np.all(st2.reshape(3, 20, 197)[2, 10:] == 0)


def test_variable_length_packing(oracle):
    from pfpn_amd.models import SyncVTraceWorker
    net = StubVTraceNet(oracle)
    w = make_worker(SyncVTraceWorker, net, lr_critic=1e-4, lr_actor=1e-4, norm_clip=0.5, opt_epochs=1, batch_size=2,
                    unroll_length=None)
    w.before_work(None)
    for n in (3, 5):
        for i in range(n):
            for k in w.exp:
                w.exp[k].append(np.full(2, 10 * n + i, np.float32) if k == "state" else np.float32(10 * n + i))
        w._close_segment(True)
    w._pack_segments()
    assert w.sequence_length == [3, 5]
    st = np.asarray(w.exp["state"])
    assert st.shape == (10, 2) and list(st[:, 0]) == [30, 31, 32, 0, 0, 50, 51, 52, 53, 54]


# --- the redesigned runtime: experience columns, learner-round plans, staged rounds -------------------------------
def test_column_and_ring_semantics():
    from pfpn_amd.models.distributed_model import Column
    c = Column()
    for i in range(200):                              # grows past its first allocation
        c.append(np.full(3, i, np.float32))
    assert len(c) == 200 and np.asarray(c).shape == (200, 3) and c[199][0] == 199
    c.pad(5)
    assert len(c) == 205 and np.all(c[200:] == 0)
    s = Column()
    s += [1.0, 2.0, 3.0]
    s.shift(1, 10.0)
    assert list(s[:]) == [1.0, 12.0, 13.0]
    s.clear()
    assert len(s) == 0 and np.asarray(s).shape == (0,)
    b = Buffer(4)
    for i in range(6):
        b.append(np.full(2, i, np.float32))
    assert len(b) == 4 and b.pointer == 2 and sorted(b.filled()[:, 0]) == [2, 3, 4, 5]
    assert np.array_equal(b[np.array([0, 0, 3])][:, 0], [4, 4, 3])     # one gather, with repeats


@pytest.mark.parametrize("n,bs,ep", [(128, 32, 1), (50, 20, 3), (7, None, 2), (64, 64, 2)])
def test_round_plan_is_the_reference_schedule(oracle, n, bs, ep):
    """same RNG stream -> the same rows in every learner step as reference flat_train (:320-352)."""
    from pfpn_amd.models.distributed_model import minibatch_plan, replay_plan
    got = minibatch_plan(n, bs, ep, np.random.RandomState(11))
    ref = oracle.flat_train_schedule(n, bs, ep, np.random.RandomState(11))
    assert len(got) == len(ref) and all(np.array_equal(a, b) for a, b in zip(got, ref))
    if bs:
        got = replay_plan(n, bs, 5, np.random.RandomState(3))
        ref = oracle.flat_train_schedule(n, bs, 5, np.random.RandomState(3), on_policy=False)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref))
        assert len(oracle.flat_train_schedule(bs - 1, bs, 5, np.random.RandomState(3), on_policy=False)) == 0


class StagedStubNet(StubNet):
    """a network that takes a whole round at once (the product's `train_round`)."""
    def train_round(self, optimizer, ops, names, columns, plan):
        self.rounds = getattr(self, "rounds", 0) + 1
        out = []
        for rows in plan:
            self.trained.append([np.asarray(columns[k])[rows].copy() for k in names])
            self.global_step += 1
            out.append(((0.1, None, 0.05, 0.05), [self.global_step]))
        return out


def test_staged_round_sees_the_same_minibatches_as_stepwise_training(oracle):
    """`train_round` (one call per round) and per-minibatch `train` receive identical rows in identical order."""
    nets = [StubNet(oracle), StagedStubNet(oracle)]
    for net in nets:
        w = make_worker(DPPOWorker, net, lr_critic=1e-4, lr_actor=1e-4, norm_clip=1.0, opt_epochs=2, batch_size=20,
                        unroll_length=50)
        w.work(None, max_iterations=12)
        assert w.global_step == 12
    assert nets[1].rounds == 2 and len(nets[0].trained) == len(nets[1].trained) == 12
    for a, b in zip(nets[0].trained, nets[1].trained):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_evaluator_plays_whole_episodes_past_max_iterations(oracle):
    """a player restored from a finished run (global_step >= max_iterations) still plays complete episodes."""
    net = StubNet(oracle, trainable=False)
    net.global_step = 10 ** 6
    w = make_worker(DPPOWorker, net, lr_critic=1e-4, lr_actor=1e-4, norm_clip=1.0, opt_epochs=1, batch_size=32,
                    unroll_length=128, _is_evaluator=1)
    w.global_step = 10 ** 6
    episodes = []
    plain = w.after_episode

    def after_episode(sess, info):
        plain(sess, info)
        episodes.append(w.episode_step)
        if len(episodes) == 2:
            w.request_stop = True
    w.after_episode = after_episode
    w.work(None, max_iterations=100)
    assert episodes == [50, 50]


def test_polling_evaluator_follows_new_checkpoints(oracle, monkeypatch, capsys):
    """DistributedModel.evaluate (reference distributed_model.py:446-496): wait for a checkpoint newer than the last one
    evaluated, restore it, average n episodes, print one [PERFORM] line per checkpoint; stop once the condition fails
    and nothing new has appeared."""
    from pfpn_amd import checkpoint
    from pfpn_amd.models import DPPOWorker, SyncModel
    net = StubNet(oracle, trainable=False)
    model = SyncModel(lambda **kw: DPPOWorker(lr_critic=1e-4, lr_actor=1e-4, norm_clip=1.0, opt_epochs=1, batch_size=32,
                                              unroll_length=128, **kw),
                      lambda name, render: SyntheticDeepMimicEnv(episode_length=50, seed=3),
                      lambda env, trainable: net, checkpoint_dir="unused", seed=3)
    seen = iter([None, "ckpt/model.ckpt-10", "ckpt/model.ckpt-10", "ckpt/model.ckpt-10", "ckpt/model.ckpt-20"])
    state = {"last": None, "polls": 0}

    def latest(directory):
        state["polls"] += 1
        state["last"] = next(seen, state["last"])
        return state["last"]
    restored = []
    monkeypatch.setattr(checkpoint, "latest_checkpoint", latest)
    monkeypatch.setattr(SyncModel, "restore", lambda self, worker, prefix=None: restored.append(prefix) or prefix)
    worker = model.evaluate(n=2, evaluation_cond=lambda: state["polls"] < 8, poll_seconds=0.0)
    assert restored == ["ckpt/model.ckpt-10", "ckpt/model.ckpt-20"]          # each new checkpoint exactly once
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("[PERFORM]")]
    assert len(lines) == 2 and "Life Time: 50.0" in lines[0]                  # 2 averaged episodes of 50 steps each
    assert worker.request_stop and state["polls"] >= 8


def test_dense_stack_keeps_a_bounded_number_of_activation_sets():
    """Ragged batches must not keep one activation set per size ever seen; sizes a captured graph points at (and the 2B-row
    sets of a pinned B) survive."""
    import torch
    from pfpn_amd.networks.ops import DenseStack, ParamStore
    store = ParamStore(torch.device("cpu"))
    stack = DenseStack(store, ["actor"], 8, [16, 8])
    store.allocate(with_optimizer=False)
    stack.pinned = {3}
    for b in range(1, 200):
        stack._buffers(b, 0)
    sizes = {k[0] for k in stack._bufs}
    assert len(stack._bufs) <= DenseStack.MAX_BUFFER_SETS + 2
    assert 3 in sizes and 6 in sizes and 199 in sizes and 1 not in sizes


def test_bench_extras_deadline_prints_what_it_has_and_exits_cleanly(tmp_path):
    """bench.py's ExtrasDeadline (multi-rank runs): a rank that parks after a failed extra leaves with exit code 0 once the
    timer fires, after the emit callback ran exactly once; finish() before the timer makes it a no-op."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, time; sys.path.insert(0, %r); import bench\n"
        "d = bench.ExtrasDeadline(0.5, lambda: (sys.stdout.write('HEADLINE\\n'), sys.stdout.flush()))\n"
        "if sys.argv[1] == 'park':\n"
        "    d.park()\n"
        "else:\n"
        "    assert d.finish() is True\n"
        "    time.sleep(1.0)\n"
        "    assert d.finish() is False          # already finished: the caller no longer owns a second print\n"
        "    print('NORMAL')\n" % root)
    r = subprocess.run([sys.executable, "-c", code, "park"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout == "HEADLINE\n", (r.returncode, r.stdout, r.stderr[-500:])
    r = subprocess.run([sys.executable, "-c", code, "finish"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout == "NORMAL\n", (r.returncode, r.stdout, r.stderr[-500:])


def test_stream_pair_layout_on_host_tensors(monkeypatch):
    """kernels.stream_pair is plain tensor arithmetic: below the threshold two allocations, above it ONE pool with the second
    buffer at (a multiple of 256 MiB) + 4 KiB behind the first (the distance the streaming backward was measured best at)."""
    import torch
    from pfpn_amd import kernels
    a, b = kernels.stream_pair(8, 16, torch.device("cpu"))
    assert a.shape == b.shape == (8, 16) and a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
    monkeypatch.setattr(kernels, "STREAM_PAIR_MIN_BYTES", 1024)
    a, b = kernels.stream_pair(64, 100, torch.device("cpu"))         # 25,600 bytes: "large" under the patched threshold
    assert a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.is_contiguous() and b.is_contiguous()
    gap = b.data_ptr() - a.data_ptr()
    assert gap == (256 << 20) + 4096 and gap >= a.numel() * 4 and a.shape == b.shape == (64, 100)


def test_bench_reads_the_committed_profiles():
    """bench.py quotes three numbers from committed rocprofv3 summaries (never measured in the run that prints them): the
    headline kernel's begin-to-end duration, the GEMM kernels' MFMA-busy share and the SAC step's launch count.  The files it
    picks must exist, be the newest round's, and parse."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    path, tag = bench.latest_profile("bench_kernel_stats.csv")
    assert path and os.path.basename(path).startswith(tag) and tag >= "r04"
    us, calls, src = bench.rocprof_kernel_us("k_mix_tile<35, 2>")
    assert 8.0 < us < 20.0 and calls > 100 and src.startswith("profiles/")
    n, src = bench.timeline_launches("sac_step_timeline.txt")
    assert 20 <= n <= 60 and src.endswith("sac_step_timeline.txt")
    n, _ = bench.timeline_launches("step_timeline.txt")
    assert n == 15
    mpath, _ = bench.latest_profile("gemm_mfma.json")
    mj = json.load(open(mpath))
    assert 0.5 < mj["mfma_busy_all_gemm_kernels"] < 1.0 and len([k for k, v in mj["kernels"].items() if "mfma_busy" in v]) >= 8
    tpath, _ = bench.latest_profile("pmc_traffic.json")
    t = json.load(open(tpath))["kernels"]["k_mix_tile<35,PPO>"]["traffic_bytes"]
    assert bench.algorithmic_bytes_fused(4096) <= t <= 1.2 * bench.algorithmic_bytes_fused(4096)
    # the printed line: a full result dict of a real run (committed) comes out under the driver's ~8.6 KB window, numbers intact,
    # the digest LAST with the figures round 4's truncated record lost
    full_path, _ = bench.latest_profile("bench_line_full.json")
    full = json.load(open(full_path or os.path.join(root, "profiles", "r04_bench_line.json")))
    line = bench.compact_line(full)
    assert len(line) <= 8000 and "\n" not in line
    d = json.loads(line)
    assert list(d)[-1] == "digest" and len(json.dumps(d["digest"])) <= 700
    assert d["value"] > 1000 and d["metric"].startswith("DPPO learner steps/sec") and d["config"]["workload"]
    assert {"stress_fwd", "stress_bwd", "stress_ppo", "sac_ms_1024", "sac_ms_256", "config5_ms", "value_200", "ms_min",
            "ms_median", "ms_max", "shard_us_512", "mfma_busy"} <= set(d["digest"])
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["sample"]
    for leg in ("fwd", "bwd", "ppo"):
        assert 0 < d["roofline_stress"][leg]["frac"] < 1
    assert "note" not in json.dumps(d)                     # the prose lives in profiles/README.md
    # a line that would still be too long loses whole blocks, least essential first, and says which
    fat = dict(full, padding={"k%d" % i: "x" * 90 for i in range(60)})
    cut = json.loads(bench.compact_line(fat))
    assert len(bench.compact_line(fat)) <= bench.LINE_BUDGET and cut["dropped"] and "value" in cut and "digest" in cut
    # multiply-accumulates per row of the DeepMimic MLPs (5,889,536 per row forward + backward incl. the critic's
    # 512-wide fc3, which runs inside an epilogue here and is not counted as GEMM work)
    f, dw, dx = bench.dense_macs(197, (1024, 512), 1260)
    assert f == 197 * 1024 + 1024 * 512 + 512 * 1260 and dx == f - 197 * 1024
    assert bench.gemm_flops(4096) == 2.0 * 4096 * (5889536 - 3 * 512)


def test_reach_environment_has_the_documented_optimum():
    """pfpn_amd.envs.SyntheticReachEnv (the second learning-test environment): the optimum a = g(s) collects reward 1 per step,
    the best state-blind policy (a = 0) ~ 0.67, uniform random actions ~ 0.18; episodes end by the time limit only."""
    import numpy as np
    from pfpn_amd.envs import SyntheticReachEnv, make_env
    env = make_env("SyntheticReach-v0", seed=3)
    assert isinstance(env, SyntheticReachEnv) and env.observation_space.shape == (197,) and env.action_space.shape == (36,)
    rng = np.random.RandomState(0)
    totals = {}
    for name, policy in (("optimal", lambda s: env.target(s)), ("zero", lambda s: np.zeros(36, np.float32)),
                         ("random", lambda s: rng.uniform(-1, 1, 36).astype(np.float32))):
        s, ret, steps, done = env.reset(), 0.0, 0, False
        while not done:
            s, r, done, info = env.step(policy(s))
            ret += r
            steps += 1
        assert steps == 200 and info.get("TimeLimit.truncated") is True
        totals[name] = ret
    assert abs(totals["optimal"] - 200.0) < 1e-3 and 110 < totals["zero"] < 160 and 20 < totals["random"] < 60, totals


def test_sac_autograd_twin_starts_from_the_uniform_mixture():
    """tools/sac_twin_torch.py (the independent SAC-PFPN implementation the GPU learning test compares the product with) runs on
    the CPU and starts where the algorithm must start: log_alpha = 0, the tanh-mode particle grid with uniform weights --
    log-prob of its own 36-dim sample ~ -24.7, mean |a| ~ 0.5, Q ~ 0 -- and its first learner step happens when the replay ring
    holds one minibatch."""
    import importlib.util
    import os
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("sac_twin_torch", os.path.join(root, "tools", "sac_twin_torch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    threads = torch.get_num_threads()
    try:
        rows, episodes = mod.run(steps=3, every=1, seed=5, threads=4, verbose=False)
    finally:
        torch.set_num_threads(threads)
    assert [r["step"] for r in rows] == [0, 1, 2] and not episodes              # 256 + 3 environment steps: no episode ended
    r0 = rows[0]
    assert abs(r0["alpha"] - 1.0) < 1e-6 and abs(r0["logp"] + 24.7) < 0.8 and 0.4 < r0["abs_a"] < 0.6 and abs(r0["q_pi"]) < 0.1
    assert all(abs(r["alpha"] - 1.0) < 1e-3 and r["value_loss"] > 100 for r in rows)


def test_fault_check_costs_nothing_where_no_waiting_kernel_ever_ran():
    """kernels.check_faults() sits on every loss read-back and in front of every checkpoint: with no fault word handed out
    (no GPU here; on a GPU: no wide particle grid ever took the one-launch PPO pass) it reads nothing and raises nothing, and
    the agreement of the replicas (graphs.any_replica) counts its rounds per process group."""
    from pfpn_amd import graphs, kernels
    assert kernels._fault_words == {}
    kernels.check_faults()
    kernels.clear_faults()
    assert issubclass(kernels.DeviceFault, RuntimeError) and issubclass(graphs.ReplicaLost, RuntimeError)
    assert graphs.inject_capture_failure is None              # "off" is None (False == 0 would read as "replica 0")
