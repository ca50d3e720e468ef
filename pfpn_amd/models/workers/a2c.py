"""A2C worker (the A3C PFPN setting's learner side): buffers, bootstrap, GAE, value targets.

Host-side mirror of reference models/workers/a2c.py (:11-86): exp buffers, train_args order
[state, action, value_target, advantage], bootstrap value 0 at a true terminal / V(last state) at a time
limit or an unroll boundary, advantage = GAE over the open segment, value_target = value + advantage.
The reference's asynchronous parameter-server variant (`AsyncA2CWorker` under `AsyncModel`) has no counterpart
here: replicas are synchronous (one RCCL all-reduce per step), so the A3C setting runs as batched A2C.
"""
import numpy as np

from .base_worker import SyncActorCriticWorker

__all__ = ["SyncA2CWorker", "BatchedA2CWorker"]


class SyncA2CWorker(SyncActorCriticWorker):
    def __init__(self, **kwargs):
        kwargs["_on_policy"] = True
        super().__init__(**kwargs)

    BUFFERS = ("state", "action", "reward", "value", "value_target", "advantage")
    LEARNER_INPUT = ("state", "action", "value_target", "advantage")

    def sample_action(self, sess, state):
        out = super().sample_action(sess, state)
        if self.target_net.trainable:
            self.exp["value"].append(out[1])         # run() -> [action, value]
        return out

    def need_train(self, sess, terminal, last_state, info):
        go = super().need_train(sess, terminal, last_state, info)
        if terminal or go or self.episodic_terminal:
            lo = self.buffer_stamp
            boot = 0.0 if (terminal and not self.overtime(info)) else self.local_net.evaluate(sess, last_state)
            values = np.append(self.exp["value"][lo:], np.float32(boot))
            adv = self.local_net.generalized_advantage_estimate(self.exp["reward"][lo:], values)
            self.exp["advantage"].extend(adv)
            self.exp["value_target"].extend(np.add(self.exp["value"][lo:], adv))     # a2c.py:42-49
            self.buffer_stamp = len(self.exp["value"])
        return go

    def setup_buffers(self):
        self.buffer_stamp = 0
        return super().setup_buffers()

    def clear_buffers(self):
        super().clear_buffers()
        self.buffer_stamp = 0


BatchedA2CWorker = SyncA2CWorker
