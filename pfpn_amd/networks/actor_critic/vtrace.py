"""V-trace (IMPALA) network for the MI355X build.

Host-side mirror of reference networks/actor_critic/vtrace.py: `v_trace_network_wrapper` (:14-207) over the
particle-filtering A2C network.  The value targets and policy-gradient advantages come from the CURRENT value
and target log-prob of the training batch (stop-gradient), computed on the device by `pfpn_vtrace` between the
forward and the backward pass; the loss is the A2C loss (-mean(logp * pg_adv) - beta * entropy + coef * value MSE,
:181-182, a2c.py:53-65) over the valid rows of zero-padded episodes.
"""
import numpy as np
import torch

from ... import kernels
from .a2c import ParticleFilteringA2CNetwork

__all__ = ["ParticleFilteringVTraceNetwork", "v_trace_network_wrapper"]


def v_trace_network_wrapper(base_a2c_class):
    class VTraceNetwork(base_a2c_class):
        def __init__(self, rho_clip=1.0, pg_rho_clip=1.0, **kwargs):
            # data are fed as (E*T, x): E episodes of equal (zero-padded) length T, episode-major
            # (reference vtrace.py:16-37)
            super().__init__(**kwargs)
            self.rho_clip = rho_clip
            self.pg_rho_clip = pg_rho_clip
            self.reward, self.running_log_prob, self.not_terminal = "reward", "pi_running", "not_terminal"
            self.batch_sequence_length = "sequence_length"

        def run(self, sess, state, ops=None):
            """reference vtrace.py:184-192: [action, action_log_prob] for one state."""
            action, logp, _ = self.run_one(state)
            if not self.trainable:
                return [action[0].cpu().numpy()]
            if logp is None:
                logp = self.policy.log_prob(action)
            packed = torch.cat([action.view(-1), logp.view(-1)]).cpu().numpy()
            A = self.n_actions
            return [packed[:A].copy(), float(packed[A])]

        def vtrace_targets(self, logp, value, reward, log_prob, not_terminal, seq_len):
            """-> (value_target vs, pg advantage, valid mask or None); episode count from seq_len."""
            E = 1 if seq_len is None else int(seq_len.shape[0])
            if seq_len is None:
                vs, pg = kernels.vtrace(logp, log_prob, value, reward, not_terminal, E, self.gamma, self.rho_clip,
                                        self.pg_rho_clip)
                return vs, pg, None
            vs, pg, mask = kernels.vtrace(logp, log_prob, value, reward, not_terminal, E, self.gamma, self.rho_clip,
                                          self.pg_rho_clip, seq_len=seq_len, want_mask=True)
            return vs, pg, mask

        def train(self, sess, optimizer, ops, state, action, reward, log_prob, not_terminal,
                  batch_sequence_length=None):
            """reference vtrace.py:194-206.  Returns ((loss, entropy-or-None, policy_loss, value_loss), extra)."""
            st, ac, rw, lp, nt = [self._to_dev(a) for a in (state, action, reward, log_prob, not_terminal)]
            sl = None
            if batch_sequence_length is not None:
                sl = torch.as_tensor(np.asarray(batch_sequence_length, dtype=np.int32)).to(self.device)
            loss4 = self._pg_gradients(
                st.view(-1, self.S), ac.view(-1, self.n_actions),
                lambda logp, v: self.vtrace_targets(logp, v, rw.view(-1), lp.view(-1), nt.view(-1), sl))
            optimizer.apply(self)
            return self._report(loss4, ops)

    return VTraceNetwork


ParticleFilteringVTraceNetwork = v_trace_network_wrapper(ParticleFilteringA2CNetwork)
