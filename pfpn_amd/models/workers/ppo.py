"""PPO worker: what is buffered per environment step, when a rollout round ends, how advantages are formed.

Behaviour of reference models/workers/ppo.py (`PPOWorker`, :11-82): the learner receives
[state, action, value, log_prob, advantage] (:23-30); `run()` returns (action, log-prob, value) and the last two
are buffered with the step (:43-48); when an episode ends, the unroll is full or a segment closes, the open
segment [buffer_stamp:] gets its GAE advantages, bootstrapped with 0 after a true terminal and with V(last state)
after a time limit or an unroll boundary (:50-73).
"""
import numpy as np

from .base_worker import SyncActorCriticWorker

__all__ = ["SyncPPOWorker", "DPPOWorker"]


class SyncPPOWorker(SyncActorCriticWorker):
    BUFFERS = ("state", "action", "reward", "value", "advantage", "log_prob")
    LEARNER_INPUT = ("state", "action", "value", "log_prob", "advantage")

    def __init__(self, **kwargs):
        super().__init__(**dict(kwargs, _on_policy=True))
        self.buffer_stamp = 0                       # first row of the segment whose advantages are still open

    def sample_action(self, sess, state):
        fetched = super().sample_action(sess, state)
        if self.target_net.trainable:
            _, log_prob, value = fetched[:3]
            self.exp["log_prob"].append(log_prob)
            self.exp["value"].append(value)
        return fetched

    def _close_advantages(self, sess, terminal, last_state, info):
        true_end = terminal and not self.overtime(info)
        bootstrap = 0.0 if true_end else self.local_net.evaluate(sess, last_state)
        lo = self.buffer_stamp
        values = np.append(self.exp["value"][lo:], np.float32(bootstrap))
        self.exp["advantage"].extend(self.local_net.generalized_advantage_estimate(self.exp["reward"][lo:], values))
        self.buffer_stamp = len(self.exp["value"])

    def need_train(self, sess, terminal, last_state, info):
        train_now = super().need_train(sess, terminal, last_state, info)
        if terminal or train_now or self.episodic_terminal:
            self._close_advantages(sess, terminal, last_state, info)
        return train_now

    def setup_buffers(self):
        self.buffer_stamp = 0
        return super().setup_buffers()

    def clear_buffers(self):
        super().clear_buffers()
        self.buffer_stamp = 0


DPPOWorker = SyncPPOWorker
