"""Off-policy worker (DDPG / TD3 / SAC): replay ring, random warm-up, one learner step per environment step.

Behaviour of reference models/workers/ddpg.py: `Buffer` (:11-27) is a fixed-capacity ring per experience key;
`DDPGWorker` (:29-94) stores (state, action, reward, not_terminal, next state), treats a time-limit ending as not
terminal, draws uniform-random actions until `observations` transitions are in, forces separate actor / critic
optimisers, and trains from minibatches sampled WITH replacement (`replay_plan` in models/distributed_model.py).
Here a ring is one preallocated float32 array per key, so a minibatch is a single fancy-index gather instead of a
Python loop over a million-slot list.
"""
import numpy as np

from .base_worker import SyncActorCriticWorker

__all__ = ["SyncDDPGWorker", "Buffer"]


class Buffer(object):
    """fixed-capacity ring: `data` holds the slots (allocated at the first append, when the row shape is known),
    `pointer` is the next slot to overwrite, len() the fill."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.data = None
        self.pointer = self.size = 0

    def append(self, item):
        if self.data is None:
            self.data = np.zeros((self.capacity,) + np.shape(item), np.float32)
        self.data[self.pointer] = item
        self.pointer = (self.pointer + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def filled(self):
        """the valid slots as an array view (slot order, not age order: sampling is uniform over them)."""
        return self.data[:self.size] if self.data is not None else np.zeros((0,), np.float32)

    def __getitem__(self, i):
        return self.filled()[i]

    def shift(self, start, delta):
        """add `delta` to the slots [start, len()) -- the rows since the environment's last `episode_reward` report
        (distributed_model.py:277-282 loops `range(stamp, len(buffer))` over them).  Slot order, like the reference's
        indices; once the ring is full len() stops growing and the range is empty, as it is there."""
        if self.data is not None and start < self.size:
            self.data[max(0, int(start)):self.size] += delta

    def clear(self):
        self.pointer = self.size = 0

    def __len__(self):
        return self.size


class SyncDDPGWorker(SyncActorCriticWorker):
    BUFFERS = LEARNER_INPUT = ("state", "action", "reward", "not_terminal", "state_")

    def __init__(self, buffer_capacity=int(1e6), observations=0, **kwargs):
        kwargs["_on_policy"] = False
        kwargs.setdefault("opt_epochs", None)          # None: as many learner steps as environment steps per round
        super().__init__(**kwargs)
        self.buffer_capacity, self.observations = buffer_capacity, observations

    def init(self):
        self.separate_optimizer = True                 # critic and actor always get their own Adam (ddpg.py:41)
        super().init()

    def setup_buffers(self):
        self.exp = {k: Buffer(self.buffer_capacity) for k in self.BUFFERS}

    def _warming_up(self):
        return not self.is_evaluator and len(self.exp["state"]) < self.observations

    def sample_action(self, sess, state):
        if self._warming_up():
            return [self.env.action_space.sample()]
        return super().sample_action(sess, state)

    def interact_with_env(self, sess, state, action):
        state_, reward, terminal, info = super().interact_with_env(sess, state, action)
        ended = terminal and not self.overtime(info)
        self.exp["not_terminal"].append(0.0 if ended else 1.0)
        self.exp["state_"].append(state_)
        if len(self.exp["state"]) == self.observations:
            terminal = True                            # the warm-up ends with an episode boundary (ddpg.py:74-75)
        return state_, reward, terminal, info

    def need_train(self, sess, terminal, last_state, info):
        n = len(self.exp["state"])
        if not self.is_evaluator and n <= self.observations:
            if n == self.observations:
                self.after_train(sess)
            return False
        return super().need_train(sess, terminal, last_state, info)
