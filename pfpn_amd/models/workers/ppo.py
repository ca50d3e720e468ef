"""PPO worker: which buffers, when to train, the GAE call.

Host-side mirror of reference models/workers/ppo.py: `PPOWorker` (:11-82) -- exp buffers (:18-21),
train_args order [state, action, value, log_prob, advantage] (:23-30), `sample_action` storing
log_prob/value (:43-48), `need_train` with bootstrap value and GAE over [buffer_stamp:] (:50-73).
"""
from .base_worker import SyncActorCriticWorker

__all__ = ["SyncPPOWorker", "DPPOWorker"]


def ppo_worker_wrapper(base_actor_critic_worker):
    class PPOWorker(base_actor_critic_worker):
        def __init__(self, **kwargs):
            kwargs["_on_policy"] = True
            super().__init__(**kwargs)

        @property
        def exp_buffers(self):
            return ["state", "action", "reward", "value", "advantage", "log_prob"]

        @property
        def train_args(self):
            return [self.exp["state"], self.exp["action"], self.exp["value"], self.exp["log_prob"],
                    self.exp["advantage"]]

        def sample_action(self, sess, state):
            result = super().sample_action(sess, state)
            if self.target_net.trainable:
                self.exp["log_prob"].append(result[1])
                self.exp["value"].append(result[2])
            return result

        def need_train(self, sess, terminal, last_state, info):
            need_train = super().need_train(sess, terminal, last_state, info)
            if terminal or need_train or self.episodic_terminal:
                if terminal and not self.overtime(info):
                    boostrap_value = 0.0
                else:
                    boostrap_value = self.local_net.evaluate(sess, last_state)
                self.exp["value"].append(boostrap_value)
                self.exp["advantage"].extend(self.local_net.generalized_advantage_estimate(
                    self.exp["reward"][self.buffer_stamp:], self.exp["value"][self.buffer_stamp:]))
                self.exp["value"].pop()
                self.buffer_stamp = len(self.exp["value"])
            return need_train

        def setup_buffers(self):
            self.buffer_stamp = 0
            return super().setup_buffers()

        def clear_buffers(self):
            super().clear_buffers()
            self.buffer_stamp = 0

    return PPOWorker


SyncPPOWorker = DPPOWorker = ppo_worker_wrapper(SyncActorCriticWorker)
