"""Off-policy worker (DDPG / SAC): replay ring, random warm-up, one learner step per environment step.

Behaviour of reference models/workers/ddpg.py: `Buffer` (:11-27) is a fixed-capacity ring per experience key;
`DDPGWorker` (:29-94) stores (state, action, reward, not_terminal, next state), treats a time-limit ending as not
terminal, draws uniform-random actions until `observations` transitions are in, forces separate actor / critic
optimisers, and trains from minibatches sampled WITH replacement (AbstractDistributedWorker.flat_train).
"""
from .base_worker import SyncActorCriticWorker

__all__ = ["SyncDDPGWorker", "Buffer"]


class Buffer(list):
    """fixed-capacity ring: `data` holds the slots, `pointer` the next slot to overwrite, len() the fill."""

    def __init__(self, capacity):
        super().__init__()
        self.capacity = int(capacity)
        self.data = [None] * self.capacity
        self.pointer = self.size = 0

    def append(self, item):
        self.data[self.pointer] = item
        self.pointer = (self.pointer + 1) % self.capacity
        if self.size < self.capacity:
            self.size += 1

    def clear(self):
        self.pointer = self.size = 0

    def __len__(self):
        return self.size


class SyncDDPGWorker(SyncActorCriticWorker):
    KEYS = ("state", "action", "reward", "not_terminal", "state_")

    def __init__(self, buffer_capacity=int(1e6), observations=0, **kwargs):
        kwargs["_on_policy"] = False
        kwargs.setdefault("opt_epochs", None)          # None: as many learner steps as environment steps per round
        super().__init__(**kwargs)
        self.buffer_capacity, self.observations = buffer_capacity, observations

    def init(self):
        self.separate_optimizer = True                 # critic and actor always get their own Adam (ddpg.py:41)
        super().init()

    exp_buffers = property(lambda self: list(self.KEYS))
    train_args = property(lambda self: [self.exp[k] for k in self.KEYS])

    def setup_buffers(self):
        self.exp = {k: Buffer(self.buffer_capacity) for k in self.KEYS}

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
