"""Worker runtime of the MI355X build: experience columns, rollout driver, learner rounds, one process per GPU.

What the reference does here (models/distributed_model.py): `AbstractDistributedWorker.work` drives one environment
step by step (:154-193), experience is appended to Python lists (:264-290), a training round re-slices those lists
into minibatches and feeds every minibatch through `sess.run` (:320-352, :372-385); `DistributedModel` builds a TF
cluster with a parameter server and an evaluator process (:406-638).

What this file does instead -- same observable behaviour (what is stored, when a round starts, which rows form which
minibatch, what the learner is called with), different machinery:

* experience lives in `Column`s (growable float32 arrays) and, for replay, `Buffer` rings with vectorised gathers;
* a learner round is a PLAN -- the list of row-index sets, drawn with exactly the reference's RNG calls
  (`shuffle` once per epoch, `choice(n, batch)` per replay step) -- handed to the network in ONE call
  (`net.train_round`): the rollout is staged on the GPU once, minibatches are gathered on the device into the static
  inputs of the captured learner step, and all losses come back in one read (the per-minibatch H2D copies and loss
  read-backs were 2/3 of a 32-row step).  Networks without `train_round` (the duck-typed stubs of the host tests, any
  reference-shaped network) get the plan minibatch by minibatch through `train()`;
* the TF cluster / parameter server / MonitoredTrainingSession become `torch.distributed` (backend "nccl" = RCCL on
  ROCm; "gloo" in the CPU tests) launched by torchrun; stop decisions are agreed across ranks because every learner
  step contains a collective; the evaluator polls the checkpoint directory like the reference's (:446-496).
PyBullet stays on the host (one environment per worker, or `pfpn_amd.env_pool` for E of them).
"""
import os
import time

import numpy as np


class Column(object):
    """One experience field of the current rollout round: rows of equal shape in a float32 array that doubles when
    full.  List-like where the workers need it (`append`, `extend`, `+=`, `len`, slicing, `clear`), array-like for
    the learner (`np.asarray(col)` is a view of the filled part)."""

    def __init__(self, rows=()):
        self._a, self._n = None, 0
        self.extend(rows)

    def _room(self, k, like):
        if self._a is None:
            self._a = np.zeros((max(64, k),) + np.shape(like), np.float32)
        elif self._n + k > len(self._a):
            grown = np.zeros((max(2 * len(self._a), self._n + k),) + self._a.shape[1:], np.float32)
            grown[:self._n] = self._a[:self._n]
            self._a = grown

    def append(self, row):
        self._room(1, row)
        self._a[self._n] = row
        self._n += 1

    def extend(self, rows):
        rows = np.asarray(rows, np.float32) if not isinstance(rows, np.ndarray) else rows
        if len(rows):
            self._room(len(rows), rows[0])
            self._a[self._n:self._n + len(rows)] = rows
            self._n += len(rows)
        return self

    __iadd__ = extend

    def pad(self, k):
        """k zero rows (closing a rollout segment shorter than the unroll length)."""
        if k > 0 and self._a is not None:
            self._room(k, self._a[0])
            self._a[self._n:self._n + k] = 0
            self._n += k

    def shift(self, start, delta):
        """rows[start:] += delta (an end-of-episode bonus spread over the episode's rows)."""
        self._a[start:self._n] += delta

    def assign(self, rows):
        self._n = 0
        self.extend(rows)

    def rows(self):
        return self._a[:self._n] if self._a is not None else np.zeros((0,), np.float32)

    def clear(self):
        self._n = 0

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self.rows()[i]

    def __array__(self, dtype=None, copy=None):
        r = self.rows()
        return r if dtype is None else r.astype(dtype, copy=False)


def minibatch_plan(n, batch_size, epochs, rng):
    """On-policy round: `epochs` passes over the n rows, each pass one fresh `rng.shuffle` of the SAME id array (so a
    pass permutes the previous pass's order, as the reference's in-place shuffle does, :327-331), cut into consecutive
    minibatches of `batch_size` (the last one ragged); no batch size: one step per pass on all rows."""
    ids, plan = np.arange(n), []
    for _ in range(int(epochs)):
        rng.shuffle(ids)
        step = int(batch_size) if batch_size else n
        plan += [ids[s:s + step].copy() for s in range(0, n, step)]
    return plan


def replay_plan(n, batch_size, steps, rng):
    """Off-policy round: `steps` minibatches drawn WITH replacement (`rng.choice(n, batch)`, reference :347)."""
    return [rng.choice(n, int(batch_size)) for _ in range(int(steps))]


class AbstractDistributedWorker(object):
    GLOBAL_NET_NAME = "global_net"
    LEARNER_INPUT = ("state", "action", "reward")          # columns handed to net.train, in order
    BUFFERS = ("state", "action", "reward")

    def __init__(self, _env_wrapper=None, _network_wrapper=None, _job_name="worker", _task_id=0,
                 _is_chief=True, _is_evaluator=0, _seed=None, _on_policy=True, _log_dir=None,
                 _max_samples=None, render=False,
                 batch_size=None, unroll_length=None, opt_epochs=1, episodic_training=False, **kwargs):
        self.env_wrapper, self.network_wrapper = _env_wrapper, _network_wrapper
        self.job_name, self.task_id = _job_name, _task_id
        self.name = "{}_{}".format(_job_name, _task_id)
        self.is_chief, self.is_evaluator = _is_chief, _is_evaluator
        self.seed, self.on_policy = _seed, _on_policy
        self.max_samples, self.render = _max_samples, render
        self.log_dir = _log_dir
        self.batch_size, self.unroll_length, self.opt_epochs = batch_size, unroll_length, opt_epochs
        # episodic training: the learner gets whole zero-padded rollout segments plus their lengths (V-trace)
        self.episodic_training = episodic_training
        self.logger = None
        self.request_stop = False
        self.global_step = self.total_samples = 0
        self.np_random = np.random.RandomState(_seed)     # `_seed` is already per replica (DistributedModel.build_worker)
        self.update_ops = None
        self.sequence_length = None
        self.current_opt_epoch = 0
        self.last_losses = []

    # -- construction ------------------------------------------------------------------------------
    def init(self):
        self.env = self.env_wrapper(self.name, self.render) if self.env_wrapper else None
        if self.env is not None and self.seed is not None and hasattr(self.env, "seed"):
            self.env.seed(self.seed)
        self.global_net = self.network_wrapper(self.env, not self.is_evaluator)
        self.global_net.init()
        # replicas share their initial weights (same network seed + broadcast) and the resampler's stream, but explore
        # independently: the sampling generator restarts from the worker's own seed (reference: every worker calls
        # tf.set_random_seed(seed + task_id + 1), :568, :73-78)
        rng = getattr(self.global_net, "rng", None)
        if hasattr(rng, "manual_seed"):
            # (no seed given: every worker still explores with its own stream -- entropy from the OS, as the reference
            # draws its seed from os.urandom in that case, :565-568)
            rng.manual_seed(int(self.seed) if self.seed is not None else int.from_bytes(os.urandom(4), "little"))
        self.init_ops = list(self.global_net.init_ops)
        self.build_local_net(self.name)

    def build_local_net(self, name):
        raise NotImplementedError

    exp_buffers = property(lambda self: list(self.BUFFERS))
    train_ops = property(lambda self: ["global_step"])
    train_args = property(lambda self: [self.exp[k] for k in self.LEARNER_INPUT])

    # -- rollout driver ------------------------------------------------------------------------------
    def work(self, sess=None, max_iterations=None):
        """Episodes until stopped.  `max_iterations` bounds LEARNER steps, so it only applies to a training worker:
        an evaluator restored from a finished run (global_step already past it) still plays whole episodes."""
        for k in ("batch_size", "unroll_length", "opt_epochs"):
            v = getattr(self, k)
            setattr(self, k, None if v is None else int(v))
        self.step_limit = max_iterations if (max_iterations is not None and self.target_net.trainable) else None
        self.request_stop = False
        self.before_work(sess)
        while not self.request_stop:
            self.play_episode(sess)
        self.after_work(sess)

    def play_episode(self, sess):
        state, info, over = self.before_episode(sess), {}, False
        while not (over or self.request_stop):
            action = self.sample_action(sess, state)[0]
            state, _, over, info = self.interact_with_env(sess, state, action)
            if self.target_net.trainable:
                self.after_env_step(sess, over, state, info)
            if self.step_limit is not None and self.global_step >= self.step_limit:
                self.request_stop = True
        self.after_episode(sess, info)

    def after_env_step(self, sess, over, state, info):
        """Learner side of one environment step: close the open segment when it ended, start a round when due."""
        due = self.need_train(sess, over, state, info)
        if self.sequence_length is not None and (over or self.episodic_terminal):
            self._close_segment(over)
        if due:
            if self.sequence_length is not None:
                self.episodic_train(sess)
            else:
                self.flat_train(sess)
            self.after_train(sess)

    def sync(self, sess):
        pass        # replicas are identical after every all-reduce step: nothing to pull

    def setup_buffers(self):
        if self.exp_buffers is not None:
            self.exp = {k: Column() for k in self.exp_buffers}

    def clear_buffers(self):
        if self.on_policy and self.exp_buffers is not None:
            for col in self.exp.values():
                col.clear()
            if self.sequence_length is not None:
                self.sequence_length.clear()
                self.sequence_terminal.clear()
        self.sample_counter = 0

    # -- rollout segments for episodic training ----------------------------------------------------------
    def _open_rows(self):
        """rows collected since the last closed segment."""
        done = sum(self.sequence_length) if self.unroll_length is None \
            else len(self.sequence_length) * self.unroll_length
        return len(self.exp["state"]) - done

    def _close_segment(self, terminal):
        """A segment ends with the episode or when `unroll_length` rows are in; fixed-length segments are
        zero-padded at once (reference :166-176)."""
        n = self._open_rows()
        if self.unroll_length is not None:
            for col in self.exp.values():
                col.pad(self.unroll_length - n)
        self.sequence_length.append(n)
        self.sequence_terminal.append(terminal)

    def _pack_segments(self):
        """Variable-length segments -> the longest one's length, zeros behind the shorter (reference :178-187)."""
        if self.unroll_length is not None or not self.sequence_length:
            return
        T, starts = max(self.sequence_length), np.cumsum([0] + list(self.sequence_length))
        for col in self.exp.values():
            src = col.rows()
            out = np.zeros((len(self.sequence_length) * T,) + src.shape[1:], np.float32)
            for s, (lo, n) in enumerate(zip(starts, self.sequence_length)):
                out[s * T:s * T + n] = src[lo:lo + n]
            col.assign(out)

    def episodic_train(self, sess):
        assert self.on_policy
        self._pack_segments()
        for _ in range(self.opt_epochs):
            self.train(sess)

    def before_work(self, sess):
        self.sequence_length = [] if self.episodic_training else None
        self.sequence_terminal = []
        self.evaluator_counter = self.sample_counter = 0
        self.setup_buffers()
        for op in self.init_ops:
            op()
        self.sync(sess)

    def after_work(self, sess):
        pass

    def before_episode(self, sess):
        self.local_net.reset_running_state()
        if not self.evaluator_counter:
            self.total_reward, self.episode_step = 0.0, 0
        self.episode_buffer_stamp = 0
        return self.env.reset()

    def after_episode(self, sess, info):
        if not self.is_evaluator:
            return
        self.evaluator_counter += 1
        if self.evaluator_counter >= self.is_evaluator:
            k = self.evaluator_counter
            print("[PERFORM] Life Time: {}; Total Reward: {:.4f}; Avg Reward: {:.4f}; Step: {}; {}".format(
                self.episode_step / k, self.total_reward / k, self.total_reward / max(1, self.episode_step),
                self.global_step, time.strftime("%m-%d %H:%M:%S", time.localtime())))
            self.evaluator_counter = 0

    def sample_action(self, sess, state):
        return self.local_net.run(sess, state, self.update_ops)

    def interact_with_env(self, sess, state, action):
        """One environment step.  The environment sees the action clipped to its box, the experience keeps the raw
        one; an `episode_reward` reported by the environment is spread over the rows since the last report."""
        space = self.env.action_space
        applied = np.clip(action, space.low, space.high) if hasattr(space, "low") else action
        state_, reward, terminal, info = self.env.step(applied)
        self.sample_counter += 1
        self.episode_step += 1
        self.total_samples += 1
        for name, value in (("state", state), ("action", action), ("reward", reward)):
            self.exp[name].append(value)
        bonus = info.get("episode_reward")
        if bonus is not None:
            rows = len(self.exp["reward"])
            if bonus != 0:
                self.exp["reward"].shift(self.episode_buffer_stamp, bonus)
                self.total_reward += (rows - self.episode_buffer_stamp) * bonus
            self.episode_buffer_stamp = rows
        self.total_reward += reward
        return state_, reward, terminal, info

    def need_train(self, sess, terminal, last_state, info):
        if self.episodic_training:         # a round per `batch_size` closed segments (the one closing now included)
            return (terminal or self.episodic_terminal) and len(self.sequence_length) + 1 >= self.batch_size
        return terminal if self.unroll_length is None else self.sample_counter >= self.unroll_length

    @property
    def episodic_terminal(self):
        """the open segment has reached `unroll_length` rows."""
        return self.sequence_length is not None and self.unroll_length is not None \
            and self._open_rows() >= self.unroll_length

    def after_train(self, sess):
        self.sync(sess)
        self.clear_buffers()
        over = self.max_samples is not None and self.total_samples > self.max_samples
        self.request_stop = self.request_stop or self.agree(over)

    def agree(self, flag):
        """A stop decision every replica takes together (logical OR): each learner step holds a collective, so a
        rank that stopped alone would leave the others waiting in their next all-reduce."""
        try:
            import torch
            import torch.distributed as dist
        except ImportError:
            return bool(flag)
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return bool(flag)
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    # -- learner rounds --------------------------------------------------------------------------------
    def flat_train(self, sess):
        """One training round on the flat (non-episodic) experience: build the plan, hand it to the network."""
        n = len(self.exp["state"])
        if self.on_policy:
            plan = minibatch_plan(n, self.batch_size, self.opt_epochs, self.np_random)
            columns = {k: col.rows() for k, col in self.exp.items()}
        elif n >= self.batch_size:
            plan = replay_plan(n, self.batch_size, self.opt_epochs or self.sample_counter, self.np_random)
            columns = {k: ring.filled() for k, ring in self.exp.items()}
        else:
            return
        self.current_opt_epoch = 0
        per_epoch = max(1, len(plan) // max(1, self.opt_epochs or 1))
        staged = getattr(self.target_net, "train_round", None)
        if staged is not None and not self.request_stop:
            cols = {k: columns[k] for k in self.LEARNER_INPUT}
            self.last_losses = staged(self.opt_op, self.train_ops, list(self.LEARNER_INPUT), cols, plan)
            self.global_step = self.target_net.global_step
            self.current_opt_epoch = self.opt_epochs or 0
            self.report(self.last_losses)
            return
        whole = self.exp
        try:
            for i, rows in enumerate(plan):
                self.current_opt_epoch = i // per_epoch
                self.exp = {k: c[rows] for k, c in columns.items()}     # what `train_args` reads
                self.train(sess)
        finally:
            self.exp = whole
        self.current_opt_epoch = self.opt_epochs or 0

    def report(self, losses):
        pass

    def train(self, sess=None):
        """One learner step on `train_args` (the reference's `worker.train(sess)`, :372-385)."""
        if self.request_stop:
            return None, None
        loss, result = self.target_net.train(sess, self.opt_op, self.train_ops, *self.train_args)
        self.global_step = self.target_net.global_step
        return loss, result

    def step(self, batch=None):
        """Alias named by BASELINE.json's north_star (`learner step`); NEW, not in the reference.
        `batch`: optional dict of the worker's learner inputs by column name."""
        if batch is not None:
            self.exp = batch
        return self.train(None)


def init_process_group(backend=None, single_rank_group=False):
    """One process per GPU: RANK/WORLD_SIZE/LOCAL_RANK/MASTER_ADDR/MASTER_PORT from torchrun.
    Returns (rank, world_size, local_rank).  `single_rank_group`: create the group even for one rank."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, world, local_rank


def shutdown_process_group():
    """Orderly end of a data-parallel run: first the hipGraphs that captured collectives (RCCL's teardown waits for
    captured point-to-point plans a live graph still references -- pfpn_amd/graphs.py), then the process group."""
    import torch.distributed as dist
    from ..graphs import release_captured_collectives
    release_captured_collectives()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


class DistributedModel(object):
    """Builds the worker of THIS process (rank) and runs it: training replica, one-shot player (`test`) or the
    polling evaluator (`evaluate`).  Replaces reference :406-638 (TF cluster, parameter server, hooks)."""

    def __init__(self, worker_wrapper, env_wrapper, network_wrapper,
                 checkpoint_dir=None, save_checkpoint_interval=1000, max_iterations=None, device="GPU:0",
                 log_dir=None, debug="chief", max_samples=None, use_evaluator=True, seed=None):
        self.worker_wrapper = worker_wrapper
        self.env_wrapper = env_wrapper
        self.network_wrapper = network_wrapper
        self.checkpoint_dir = checkpoint_dir
        self.save_checkpoint_interval = save_checkpoint_interval
        self.max_iterations = max_iterations
        self.max_samples = max_samples
        self.log_dir = log_dir
        self.seed = seed

    def worker_wrapper_kwargs(self, world_size):
        return {}

    def build_worker(self, evaluator=0, render=False):
        rank, world, _ = init_process_group()
        seed = self.seed if (self.seed is None or evaluator) else self.seed + rank + 1      # reference :565-568
        worker = self.worker_wrapper(_job_name="evaluator" if evaluator else "worker", _task_id=rank,
                                     _is_chief=(rank == 0) and not evaluator, _is_evaluator=int(evaluator),
                                     _env_wrapper=self.env_wrapper, _network_wrapper=self.network_wrapper,
                                     _seed=seed, _log_dir=self.log_dir, _max_samples=self.max_samples,
                                     render=render, **self.worker_wrapper_kwargs(world))
        worker.init()
        return worker

    # -- checkpoints: TensorFlow tensor_bundle files under the reference's names (pfpn_amd/checkpoint.py) ----------
    def restore(self, worker, prefix=None):
        from .. import checkpoint
        if prefix is None:
            prefix = checkpoint.latest_checkpoint(self.checkpoint_dir) if self.checkpoint_dir else None
        if prefix is None or not os.path.exists(prefix + ".data-00000-of-00001"):
            return None
        opt = getattr(worker, "opt_op", None)
        rd = checkpoint.restore(prefix, worker.global_net, opt if hasattr(opt, "groups") else None)
        worker.global_step = worker.global_net.global_step
        key = worker.GLOBAL_NET_NAME + "/samples/samples"
        if key in rd.entries:
            worker.total_samples = int(rd.get(key))
        print("[RESTORE] {} (global step {})".format(prefix, worker.global_step))
        return prefix

    def save(self, worker):
        from .. import checkpoint
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        prefix = os.path.join(self.checkpoint_dir, "model.ckpt-%d" % worker.global_net.global_step)
        opt = getattr(worker, "opt_op", None)
        checkpoint.save(prefix, worker.global_net, opt if hasattr(opt, "groups") else None,
                        total_samples=worker.total_samples)
        return prefix

    def start(self, distributions=None):
        return self.test() if distributions is None else self.train(distributions)

    def train(self, distributions=None):
        worker = self.build_worker()
        self.restore(worker)
        saves = self.checkpoint_dir and worker.is_chief and self.save_checkpoint_interval
        if saves:
            every, last, plain_after_train = int(self.save_checkpoint_interval), [worker.global_step], worker.after_train

            def after_train_and_save(sess):
                plain_after_train(sess)
                if worker.global_step - last[0] >= every:
                    last[0] = worker.global_step
                    self.save(worker)
            worker.after_train = after_train_and_save
        worker.work(None, max_iterations=self.max_iterations)
        if saves and worker.global_step > 0:
            self.save(worker)
        return worker

    def test(self, episodes=None):
        """`main` without --train: play the latest checkpoint, episode after episode (reference :436-444 loops
        forever; `episodes` bounds it for scripted use)."""
        worker = self.build_worker(evaluator=1, render=True)
        self.restore(worker)
        self._bound_episodes(worker, episodes)
        worker.work(None)
        return worker

    def evaluate(self, n=10, evaluation_cond=lambda: True, poll_seconds=30.0, rounds=None):
        """The background evaluator of a training run (reference :446-496): wait for a checkpoint that is newer than
        the last one evaluated (polling every `poll_seconds` while `evaluation_cond()` holds), restore it, average
        `n` episodes, print the [PERFORM] line, repeat; stop when the condition fails without a new checkpoint.
        `rounds` bounds the number of checkpoints evaluated (None: unbounded)."""
        from .. import checkpoint
        worker = self.build_worker(evaluator=n)
        rng = np.random.RandomState(self.seed)
        seen, done = [None], [0]
        plain_before_episode = worker.before_episode

        def next_checkpoint():
            latest = checkpoint.latest_checkpoint(self.checkpoint_dir)
            while (latest is None or latest == seen[0]) and evaluation_cond():
                time.sleep(poll_seconds)
                latest = checkpoint.latest_checkpoint(self.checkpoint_dir)
            return None if latest == seen[0] else latest

        def before_episode(sess):
            if worker.evaluator_counter == 0:                 # between two averaged groups of n episodes
                fresh = None if (rounds is not None and done[0] >= rounds) else next_checkpoint()
                if fresh is None:
                    worker.request_stop = True
                else:
                    self.restore(worker, fresh)
                    seen[0] = fresh
                    done[0] += 1
                    if hasattr(worker.env, "seed"):
                        worker.env.seed(int(rng.randint(1 << 30)))
            return plain_before_episode(sess)
        worker.before_episode = before_episode
        worker.work(None)
        return worker

    @staticmethod
    def _bound_episodes(worker, episodes):
        if episodes is None:
            return
        left, plain = [int(episodes)], worker.after_episode

        def after_episode(sess, info):
            plain(sess, info)
            left[0] -= 1
            if left[0] <= 0:
                worker.request_stop = True
        worker.after_episode = after_episode
