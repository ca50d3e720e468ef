"""Off-policy (DDPG/SAC) worker: ring replay buffer, warm-up, per-step training.

Host-side mirror of reference models/workers/ddpg.py: `Buffer` (:11-27), `DDPGWorker` (:29-94) --
exp buffers (:44-46), train_args (:58-63), `interact_with_env` storing not_terminal / state_ (:70-76),
replay sampling with replacement in AbstractDistributedWorker.flat_train.
"""
from .base_worker import SyncActorCriticWorker

__all__ = ["SyncDDPGWorker", "Buffer"]


class Buffer(list):
    """ring buffer (reference ddpg.py:11-27)."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.data = [None] * self.capacity
        self.clear()

    def clear(self):
        self.pointer = 0
        self.size = 0

    def append(self, item):
        self.data[self.pointer] = item
        self.pointer = (self.pointer + 1) % self.capacity
        self.size = min(self.capacity, self.size + 1)

    def __len__(self):
        return self.size


def ddpg_worker_wrapper(base_actor_critic_worker):
    class DDPGWorker(base_actor_critic_worker):
        def __init__(self, buffer_capacity=int(1e6), observations=0, **kwargs):
            kwargs["_on_policy"] = False
            if "opt_epochs" not in kwargs:
                kwargs["opt_epochs"] = None          # train with steps equal to the unroll length
            super().__init__(**kwargs)
            self.buffer_capacity = buffer_capacity
            self.observations = observations

        def init(self):
            self.separate_optimizer = True           # ddpg.py:41
            super().init()

        @property
        def exp_buffers(self):
            return ["state", "action", "reward", "not_terminal", "state_"]

        @property
        def train_args(self):
            return [self.exp["state"], self.exp["action"], self.exp["reward"], self.exp["not_terminal"],
                    self.exp["state_"]]

        def sample_action(self, sess, state):
            if not self.is_evaluator and len(self.exp["state"]) < self.observations:
                return [self.env.action_space.sample()]
            return super().sample_action(sess, state)

        def interact_with_env(self, sess, state, action):
            state_, reward, terminal, info = super().interact_with_env(sess, state, action)
            self.exp["not_terminal"].append(0.0 if terminal and not self.overtime(info) else 1.0)
            self.exp["state_"].append(state_)
            if len(self.exp["state"]) == self.observations:
                terminal = True
            return state_, reward, terminal, info

        def need_train(self, sess, terminal, last_state, info):
            if not self.is_evaluator and len(self.exp["state"]) <= self.observations:
                if len(self.exp["state"]) == self.observations:
                    self.after_train(sess)
                return False
            return super().need_train(sess, terminal, last_state, info)

        def setup_buffers(self):
            self.exp = {k: Buffer(self.buffer_capacity) for k in self.exp_buffers}

    return DDPGWorker


SyncDDPGWorker = ddpg_worker_wrapper(SyncActorCriticWorker)
