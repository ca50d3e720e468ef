"""V-trace worker (the IMPALA PFPN setting's learner side).

Host-side mirror of reference models/workers/vtrace.py (:11-57): episodic training is forced on, the buffers
hold [state, action, reward, behaviour log_prob, not_terminal] and the learner also receives the segment
lengths; a time-limit ending counts as not terminal.  The reference's actor/learner queue model
(`IMPALAWorker` under `LearnerModel`) is replaced by synchronous replicas: every rank is actor and learner.
"""
from .base_worker import SyncActorCriticWorker

__all__ = ["SyncVTraceWorker", "IMPALAWorker", "VTraceLearner"]


class SyncVTraceWorker(SyncActorCriticWorker):
    def __init__(self, **kwargs):
        kwargs["_on_policy"] = True
        kwargs["episodic_training"] = True
        super().__init__(**kwargs)

    BUFFERS = LEARNER_INPUT = ("state", "action", "reward", "log_prob", "not_terminal")
    # the learner also gets the segment lengths
    train_args = property(lambda self: [self.exp[k] for k in self.LEARNER_INPUT] + [self.sequence_length])

    def sample_action(self, sess, state):
        out = super().sample_action(sess, state)
        if self.target_net.trainable:
            self.exp["log_prob"].append(out[1])      # run() -> [action, action_log_prob]
        return out

    def interact_with_env(self, sess, state, action):
        state_, reward, terminal, info = super().interact_with_env(sess, state, action)
        self.exp["not_terminal"].append(0.0 if (terminal and not self.overtime(info)) else 1.0)
        return state_, reward, terminal, info


IMPALAWorker = VTraceLearner = SyncVTraceWorker
