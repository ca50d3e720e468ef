"""Clipped-PPO wrapper over the particle-filtering A2C network (the DPPO learner step).

Host-side mirror of reference networks/actor_critic/ppo.py: `clip_ppo_network_wrapper` (:14-74) --
`value_target = advantage + running_value` (:31-34), value loss (:39-42), clipped surrogate (:44-54),
`run()` -> [action, log_prob, value] (:56-62), `train()` feed order (:64-72) -- on top of
actor_critic.py's loss assembly (:132-184: value_loss*coef, advantage normalisation, loss sum).

One learner step = state-norm(+EMA) -> actor & critic MLP forward (PyTorch-ROCm fp32 GEMMs) ->
fused HIP kernel [mixture log-prob of action_hist + PPO ratio/clip/value loss + full backward of
the mixture] -> MLP backward -> optimizer (clip, all-reduce, TF-Adam, resample tick).
"""
import numpy as np
import torch

from ... import kernels
from ..utils import MixtureGaussianDistribution
from .a2c import ParticleFilteringA2CNetwork

__all__ = ["ParticleFilteringClipPPONetwork"]


def clip_ppo_network_wrapper(a2c_base_class):
    class ClipPPONetwork(a2c_base_class):
        def __init__(self, epsilon=0.2, **kwargs):
            super().__init__(**kwargs)
            self.epsilon = epsilon
            self.running_log_prob = "pi_running"
            self.running_value = "value_running"

        def _alloc_policy_buffers(self, B):
            super()._alloc_policy_buffers(B)
            dev, AN = self.device, self.n_actions * self.particles
            hc = self.critic_net_shape[-1]
            G = 2 if self.trunk is not None else 1
            self.train_out = {
                "logp": torch.empty(B, dtype=torch.float32, device=dev),
                "dlogits": torch.empty(B, AN, dtype=torch.float32, device=dev),
                "dloc": self.store.g("actor/samples"),
                "dlogstd": self.store.g("actor/samples_std"),
                "dvalue": torch.empty(B, dtype=torch.float32, device=dev),
                "loss3": torch.empty(3, dtype=torch.float32, device=dev),
            }
            self.adv_stats = torch.empty(2, dtype=torch.float32, device=dev)
            if self.trunk is not None:
                self.dh = torch.empty(2, B, hc, dtype=torch.float32, device=dev)
            else:
                self.dh_a = torch.empty(1, B, self.actor_net_shape[-1], dtype=torch.float32, device=dev)
                self.dh_c = torch.empty(1, B, hc, dtype=torch.float32, device=dev)
            self.entropy_buf = None

        # -- rollout ---------------------------------------------------------------------------
        def run(self, sess, state, ops=None):
            """reference ppo.py:56-62: [action, action_log_prob, value] for one state (row 0)."""
            action, logp, value = self.run_batch(np.asarray(state, dtype=np.float32)[None])
            if not self.trainable:
                return [action[0].cpu().numpy()]
            if logp is None:       # deterministic action: log-prob of the mean action
                logp = self.policy.log_prob(action)
            packed = torch.cat([action.view(-1), logp.view(-1), value.view(-1)]).cpu().numpy()
            A = self.n_actions
            return [packed[:A].copy(), float(packed[A]), float(packed[A + 1])]

        # -- learner step ------------------------------------------------------------------------
        def compute_gradients(self, state, action, value, log_prob, advantage):
            """Forward + backward of loss = policy_loss + value_loss_coef*value_loss on device tensors.
            Gradients land in self.store.grad; returns loss3 (device): {loss, policy_loss, value_loss*coef}.
            No host synchronisation: capturable in a hipGraph."""
            B = state.shape[0]
            st = self.store
            self._ensure_buffers(B)
            x = self._normalize(state)
            self.update_state_normalizer(state)        # side stream, joined in the optimiser
            with self.aux():                           # independent of the forward pass
                kernels.adv_stats(advantage, self.normalize_advantage, out=self.adv_stats)
            ha, hc = self._trunks_forward(x)
            with self.aux():                           # critic head beside the policy-head GEMM
                v = self._value(hc)
            policy = MixtureGaussianDistribution(self._policy_logits(ha), self.loc, self.logstd,
                                                 self.normalize_policy_output_)
            self.aux_join()
            # without an entropy term dlogits is final when the fused kernel ends: it then also emits the
            # fc_policy bias gradient (column sums of dlogits), saving a 20 MB pass
            fused_db = not self.entropy_beta
            out = kernels.ppo_mixture_fwd_bwd(policy.logits, action, self.loc, self.logstd, log_prob, advantage, v,
                                              value, self.adv_stats, self.epsilon, self.value_loss_coef,
                                              out=self.train_out,
                                              dbias=st.g("actor/fc_policy/bias") if fused_db else None)
            dlogits = out["dlogits"]
            if self.entropy_beta:
                # policy_loss -= beta * mean_b sum_a H  (actor_critic.py:172-176, a2c.py:62-65)
                if self.entropy_buf is None or self.entropy_buf.shape[0] != B:
                    self.entropy_buf = torch.full((B, self.n_actions), -float(self.entropy_beta) / B,
                                                  dtype=torch.float32, device=self.device)
                self.last_entropy = policy.entropy()
                policy.entropy_backward(self.entropy_buf, dlogits)
            if self.fixed_sigma:
                out["dlogstd"].zero_()
            w3 = st["critic/fc3/weight"].view(-1)
            dh_a, dh_c = (self.dh[0], self.dh[1]) if self.trunk is not None else (self.dh_a[0], self.dh_c[0])
            # critic-head backward and the fc_policy bias gradient (column sums of dlogits) are short
            # latency-bound launches: beside the two fc_policy GEMMs
            with self.aux():
                kernels.value_head_bwd(hc, w3, out["dvalue"], dh_c, st.g("critic/fc3/weight").view(-1),
                                       st.g("critic/fc3/bias"), relu6=True)
                if not fused_db:
                    kernels.act_bwd_bias_grad(None, dlogits, st.g("actor/fc_policy/bias"), relu6=False)
            dws = self.dw_stream if self.overlap_dw else None
            torch.mm(ha.t(), dlogits, out=st.g("actor/fc_policy/weight"))
            torch.mm(dlogits, self.policy_weight.t(), out=dh_a)
            self.aux_join()
            if self.trunk is not None:
                self.trunk.backward(self.dh, dw_stream=dws)
            else:
                self.actor_trunk.backward(self.dh_a)
                self.critic_trunk.backward(self.dh_c)
            self._dw_pending = dws is not None
            return out["loss3"]

        def train_device(self, optimizer, state, action, value, log_prob, advantage):
            """One learner step on device tensors; returns loss3 (device tensor, no sync)."""
            loss3 = self.compute_gradients(state, action, value, log_prob, advantage)
            optimizer.apply(self)
            return loss3

        def train(self, sess, optimizer, ops, state, action, value, log_prob, advantage):
            """reference ppo.py:64-72 + actor_critic.py:382-404.
            Returns ((loss, entropy-or-None, policy_loss, value_loss), extra_fetches)."""
            args = [self._to_dev(a) for a in (state, action, value, log_prob, advantage)]
            args[0] = args[0].view(-1, self.S)
            args[1] = args[1].view(-1, self.n_actions)
            loss3 = self.train_device(optimizer, *args).cpu().numpy()
            ent = None
            if self.entropy_beta:
                ent = float(self.last_entropy.sum(1).mean().item())
                loss3 = loss3.copy()
                loss3[1] -= self.entropy_beta * ent
                loss3[0] -= self.entropy_beta * ent
            loss = (float(loss3[0]), ent, float(loss3[1]), float(loss3[2]))
            extra = []
            for o in (ops if hasattr(ops, "__len__") else ([ops] if ops else [])):
                extra.append(self.global_step if o in ("global_step", "step") else (o() if callable(o) else None))
            return loss, extra

    return ClipPPONetwork


ParticleFilteringClipPPONetwork = clip_ppo_network_wrapper(ParticleFilteringA2CNetwork)
