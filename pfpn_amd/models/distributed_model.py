"""Worker runtime: rollout loop, experience buffers, minibatch scheduling, one-process-per-GPU launch.

Host-side mirror of reference models/distributed_model.py: `AbstractDistributedWorker.work` (:154-193),
`interact_with_env` (:264-290), `need_train` (:292-299), `flat_train` (:320-352), `train` (:372-385),
`DistributedModel.dispatch` (:548-625).  The TF cluster / parameter server / MonitoredTrainingSession /
evaluator process are replaced by `torch.distributed` (backend "nccl" = RCCL on ROCm, or "gloo" for the
CPU-side tests of the host logic) launched by torchrun: ranks read RANK / WORLD_SIZE / LOCAL_RANK /
MASTER_ADDR / MASTER_PORT from the environment.  PyBullet stays on the host (one env per worker).
"""
import os
import time

import numpy as np


class AbstractDistributedWorker(object):
    GLOBAL_NET_NAME = "global_net"

    def __init__(self, _env_wrapper=None, _network_wrapper=None, _job_name="worker", _task_id=0,
                 _is_chief=True, _is_evaluator=0, _seed=None, _on_policy=True, _log_dir=None,
                 _max_samples=None, render=False,
                 batch_size=None, unroll_length=None, opt_epochs=1, episodic_training=False, **kwargs):
        self.env_wrapper = _env_wrapper
        self.network_wrapper = _network_wrapper
        self.job_name, self.task_id = _job_name, _task_id
        self.name = "{}_{}".format(_job_name, _task_id)
        self.is_chief = _is_chief
        self.is_evaluator = _is_evaluator
        self.seed = _seed
        self.on_policy = _on_policy
        self.max_samples = _max_samples
        self.render = render
        self.batch_size = batch_size
        self.unroll_length = unroll_length
        self.opt_epochs = opt_epochs
        # episodic training: the learner is fed whole (zero-padded) rollout segments plus their lengths --
        # what the V-trace network needs (reference distributed_model.py:166-189, :293-310, :354-370)
        self.episodic_training = episodic_training
        self.logger = None
        self.request_stop = False
        self.global_step = 0
        self.total_samples = 0
        self.np_random = np.random.RandomState(_seed)
        self.update_ops = None
        self.sequence_length = None
        self.current_opt_epoch = 0

    # -- construction ----------------------------------------------------------------------------
    def init(self):
        """reference distributed_model.py:86-111: build env + global net."""
        self.env = self.env_wrapper(self.name, self.render) if self.env_wrapper else None
        if self.env is not None and self.seed is not None and hasattr(self.env, "seed"):
            self.env.seed(self.seed + self.task_id)
        self.global_net = self.network_wrapper(self.env, not self.is_evaluator)
        self.global_net.init()
        self.init_ops = list(self.global_net.init_ops)
        self.build_local_net(self.name)

    def build_local_net(self, name):
        raise NotImplementedError

    # -- rollout loop (reference :154-193) --------------------------------------------------------
    def work(self, sess=None, max_iterations=None):
        self.batch_size = None if self.batch_size is None else int(self.batch_size)
        self.unroll_length = None if self.unroll_length is None else int(self.unroll_length)
        self.opt_epochs = None if self.opt_epochs is None else int(self.opt_epochs)
        self.request_stop = False
        self.before_work(sess)
        while not self.request_stop:
            state = self.before_episode(sess)
            terminal = False
            info = {}
            while not terminal and not self.request_stop:
                action, *_ = self.sample_action(sess, state)
                state, _, terminal, info = self.interact_with_env(sess, state, action)
                if self.target_net.trainable:
                    need_train = self.need_train(sess, terminal, state, info)
                    if self.sequence_length is not None and (terminal or self.episodic_terminal):
                        self._close_segment(terminal)
                    if need_train:
                        if self.sequence_length is not None:
                            self.episodic_train(sess)
                        else:
                            self.flat_train(sess)
                        self.after_train(sess)
                if max_iterations is not None and self.global_step >= max_iterations:
                    self.request_stop = True
            self.after_episode(sess, info)
        self.after_work(sess)

    def sync(self, sess):
        pass        # replicas are identical after every all-reduce step: nothing to pull

    def setup_buffers(self):
        if self.exp_buffers is not None:
            self.exp = {k: [] for k in self.exp_buffers}

    def clear_buffers(self):
        if self.on_policy and self.exp_buffers is not None:
            for _, v in self.exp.items():
                v.clear()
            if self.sequence_length is not None:
                self.sequence_length.clear()
                self.sequence_terminal.clear()
        self.sample_counter = 0

    # -- rollout segments for episodic training ---------------------------------------------------
    def _open_rows(self):
        """rows collected since the last closed segment."""
        closed = len(self.sequence_length) * (self.unroll_length if self.unroll_length is not None else 0)
        if self.unroll_length is None:
            closed = sum(self.sequence_length)
        return len(self.exp["state"]) - closed

    def _close_segment(self, terminal):
        """A segment ends at an episode end or when `unroll_length` rows are in: record its length; with a
        fixed unroll length it is zero-padded to that length at once (reference :166-176)."""
        n = self._open_rows()
        if self.unroll_length is not None:
            for buf in self.exp.values():
                buf.extend(np.zeros_like(buf[0]) for _ in range(self.unroll_length - n))
        self.sequence_length.append(n)
        self.sequence_terminal.append(terminal)

    def _pack_segments(self):
        """Variable-length segments -> equal length: zero rows behind every shorter one (reference :178-187)."""
        if self.unroll_length is not None or not self.sequence_length:
            return
        T = max(self.sequence_length)
        for k, buf in self.exp.items():
            packed, pos = [], 0
            for n in self.sequence_length:
                seg = buf[pos:pos + n]
                packed.extend(seg)
                packed.extend(np.zeros_like(buf[0]) for _ in range(T - n))
                pos += n
            buf[:] = packed

    def episodic_train(self, sess):
        """reference :354-370 without the recurrent training state (no PFPN setting has one)."""
        assert self.on_policy
        self._pack_segments()
        for _ in range(self.opt_epochs):
            self.train(sess)

    def before_work(self, sess):
        self.sequence_length = [] if self.episodic_training else None
        self.sequence_terminal = []
        self.evaluator_counter = 0
        self.sample_counter = 0
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
        if self.is_evaluator:
            self.evaluator_counter += 1
            if self.evaluator_counter >= self.is_evaluator:
                print("[PERFORM] Life Time: {}; Total Reward: {:.4f}; Avg Reward: {:.4f}; Step: {}; {}".format(
                    self.episode_step / self.evaluator_counter, self.total_reward / self.evaluator_counter,
                    self.total_reward / max(1, self.episode_step), self.global_step,
                    time.strftime("%m-%d %H:%M:%S", time.localtime())))
                self.evaluator_counter = 0

    def sample_action(self, sess, state):
        return self.local_net.run(sess, state, self.update_ops)

    def interact_with_env(self, sess, state, action):
        """reference :264-290: the env sees the clipped action, the buffer stores the unclipped one."""
        self.sample_counter += 1
        self.episode_step += 1
        if hasattr(self.env.action_space, "low"):
            a = np.clip(action, self.env.action_space.low, self.env.action_space.high)
        else:
            a = action
        state_, reward, terminal, info = self.env.step(a)
        self.exp["state"].append(state)
        self.exp["action"].append(action)
        self.exp["reward"].append(reward)
        if "episode_reward" in info:
            if info["episode_reward"] != 0:
                for i in range(self.episode_buffer_stamp, len(self.exp["reward"])):
                    self.exp["reward"][i] += info["episode_reward"]
                self.total_reward += (len(self.exp["reward"]) - self.episode_buffer_stamp) * info["episode_reward"]
            self.episode_buffer_stamp = len(self.exp["reward"])
        self.total_reward += reward
        self.total_samples += 1
        return state_, reward, terminal, info

    def need_train(self, sess, terminal, last_state, info):
        if self.episodic_training:         # train once `batch_size` segments are in (reference :293-297)
            if terminal or self.episodic_terminal:
                return len(self.sequence_length) + 1 >= self.batch_size
            return False
        if self.unroll_length is None:
            return terminal
        return self.sample_counter >= self.unroll_length

    @property
    def episodic_terminal(self):
        """the open segment has reached `unroll_length` rows (reference :302-308)."""
        if self.sequence_length is not None and self.unroll_length is not None:
            return self._open_rows() >= self.unroll_length
        return False

    def after_train(self, sess):
        self.sync(sess)
        self.clear_buffers()
        if self.max_samples is not None and self.total_samples > self.max_samples:
            self.request_stop = True

    # -- minibatch scheduling (reference :320-352) -------------------------------------------------
    def flat_train(self, sess):
        n = len(self.exp["state"])
        exp_bak = self.exp
        if self.on_policy:      # disposable experience buffer
            exp = {k: np.asarray(v, dtype=np.float32) for k, v in exp_bak.items()}
            ids = np.arange(n)
            self.current_opt_epoch = 0
            while self.current_opt_epoch < self.opt_epochs:
                self.np_random.shuffle(ids)
                if self.batch_size:
                    for s in range(0, n, self.batch_size):
                        cand = ids[s:s + self.batch_size]
                        self.exp = {k: buf[cand] for k, buf in exp.items()}
                        self.train(sess)
                else:
                    self.exp = {k: buf[ids] for k, buf in exp.items()}
                    self.train(sess)
                self.current_opt_epoch += 1
        elif n >= self.batch_size:
            for _ in range(self.opt_epochs if self.opt_epochs else self.sample_counter):
                ids = self.np_random.choice(n, self.batch_size)          # with replacement (:347)
                self.exp = {k: np.asarray([v.data[i] for i in ids], dtype=np.float32) for k, v in exp_bak.items()}
                self.train(sess)
        self.exp = exp_bak

    def train(self, sess=None):
        """One learner step: the reference's `worker.train(sess)` (:372-385)."""
        if self.request_stop:
            return None, None
        ops = self.train_ops
        loss, result = self.target_net.train(sess, self.opt_op, ops, *self.train_args)
        self.global_step = self.target_net.global_step
        return loss, result

    def step(self, batch=None):
        """Alias named by BASELINE.json's north_star (`learner step`); NEW, not in the reference.
        `batch`: optional dict of the worker's train_args by buffer name."""
        if batch is not None:
            self.exp = batch
        return self.train(None)

    @property
    def exp_buffers(self):
        return ["state", "action", "reward"]

    @property
    def train_ops(self):
        return ["global_step"]

    @property
    def train_args(self):
        return [self.exp["state"], self.exp["action"], self.exp["reward"]]


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


class DistributedModel(object):
    """reference :406-638, minus the TF cluster: builds one worker in THIS process (rank) and runs it."""

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

    def build_worker(self, evaluator=False):
        rank, world, _ = init_process_group()
        worker = self.worker_wrapper(_job_name="evaluator" if evaluator else "worker", _task_id=rank,
                                     _is_chief=(rank == 0), _is_evaluator=1 if evaluator else 0,
                                     _env_wrapper=self.env_wrapper, _network_wrapper=self.network_wrapper,
                                     _seed=self.seed, _log_dir=self.log_dir, _max_samples=self.max_samples,
                                     **self.worker_wrapper_kwargs(world))
        worker.init()
        return worker

    # -- checkpoints: TensorFlow tensor_bundle files under the reference's names (pfpn_amd/checkpoint.py),
    #    so runs can be resumed / evaluated across the two code bases (reference :436-496, :608-619) -------
    def restore(self, worker):
        from .. import checkpoint
        prefix = checkpoint.latest_checkpoint(self.checkpoint_dir) if self.checkpoint_dir else None
        if prefix is None or not os.path.exists(prefix + ".data-00000-of-00001"):
            return None
        checkpoint.restore(prefix, worker.global_net, getattr(worker, "opt_op", None)
                           if hasattr(getattr(worker, "opt_op", None), "groups") else None)
        worker.global_step = worker.global_net.global_step
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
        worker = self.build_worker(evaluator=distributions is None)
        self.restore(worker)
        if self.checkpoint_dir and worker.is_chief and not worker.is_evaluator and self.save_checkpoint_interval:
            every, state = int(self.save_checkpoint_interval), {"last": worker.global_step}
            after = worker.after_train

            def after_train_and_save(sess):
                after(sess)
                if worker.global_step - state["last"] >= every:
                    state["last"] = worker.global_step
                    self.save(worker)
            worker.after_train = after_train_and_save
        worker.work(None, max_iterations=self.max_iterations)
        if self.checkpoint_dir and worker.is_chief and not worker.is_evaluator and worker.global_step > 0:
            self.save(worker)
        return worker
