"""TD3 / DDPG wrapper over the particle-filtering policy (deterministic particle action, twin critics).

Host-side mirror of reference networks/actor_critic/ddpg.py: `ddpg_network_wrapper` (:11-243) --
* the actor's output is the DETERMINISTIC particle action `tanh(loc)[argmax softmax(logits)]` with the custom gradient
  of `MixtureGaussianDistribution.mean()` (utils.py:205-221); exploration adds `act_noise * N(0,1)` and clips to the
  action box (:136-170); `act_noise` also fixes the particle width (`fixed_sigma`, `init_sigma`, :48-50);
* critics `q` and `twin_q` on concat(normalised state, action); the eval net returns [Q(s,a_hist), twinQ(s,a_hist),
  Q(s, actor(s))] (:174-197);
* the target net is a FULL copy (actor, critics, state-normaliser statistics) built under `target_net/` (:31-36,
  :201-219): target action = clip(actor'(s') + target_noise * N(0,1)), value = min(Q', twinQ')(s', a');
  `q_target = r + gamma * not_terminal * value'` (:216-219);
* value loss = mean_b((Q - y)^2 + (twinQ - y)^2) * coef w.r.t. the critic variables (:221-224, a2c.py:51-56); policy loss
  = -mean(Q(s, actor(s))) w.r.t. the actor variables, its gradient only on steps with global_step % policy_delay == 0
  (:199-207); on those steps every target variable moves by tau towards its eval twin (:97-109); hard copy at init
  (:110-121); `run` -> [action] (:226-230), `train` feed (:232-243).
Device work: the dense stacks (PyTorch-ROCm GEMMs + HIP epilogues), `pfpn_mean_action`, the custom gradient through
`pfpn_rsample_bwd` with zero noise (the reparameterised sample degenerates to the deterministic action: softmax(l + 0),
loc + scale*0), the loss head of `pfpn_sac_loss` with alpha = 0 and both policy-Q rows equal, `pfpn_soft_update`.
Not built (no shipped setting uses them): Ornstein-Uhlenbeck noise, `twin_q=False`, learnable noise (`act_noise=None`).
"""
import numpy as np
import torch

from ... import kernels
from .. import ops
from ..ops import DenseStack, ParamStore
from ..utils import MixtureGaussianDistribution
from .a2c import ParticleFilteringA2CNetwork

__all__ = ["ParticleFilteringDDPGNetwork"]


def ddpg_network_wrapper(base_class):
    class DDPGNetwork(base_class):
        def __init__(self, twin_q=True, policy_delay=2, tau=0.005, ou_noise=False, ou_sigma=0.3, ou_theta=0.15,
                     ou_dt=0.01, act_noise=0.1, target_noise=0.2, **kwargs):
            if ou_noise:
                raise NotImplementedError("Ornstein-Uhlenbeck action noise is not built (TD3's Gaussian noise is)")
            if not twin_q:
                raise NotImplementedError("twin_q=False (plain DDPG) is not built: the reference default is TD3")
            if act_noise is None or target_noise is None:
                raise NotImplementedError("learnable exploration / smoothing noise (act_noise=None) is not built")
            kwargs.setdefault("normalize_policy_output", True)              # ddpg.py:25-26
            kwargs["fixed_sigma"], kwargs["init_sigma"] = True, act_noise   # ddpg.py:48-50
            super().__init__(**kwargs)
            self.twin_q, self.policy_delay, self.tau = twin_q, int(policy_delay), tau
            self.act_noise, self.target_noise = float(act_noise), float(target_noise)
            self.reward, self.not_terminal = "reward", "not_terminal"

        # -- parameters: critic tensors first, then actor tensors (the two Adam groups) ------------------
        def _declare(self, store):
            A = int(self.action_shape[0])
            self.trunk = self.critic_trunk = None
            hc = self.critic_net_shape[-1]
            self.q_stack = DenseStack(store, ["critic/q", "critic/twin_q"], self.S + A, self.critic_net_shape)
            store.declare("critic/q+critic/twin_q|fc3/weight", (2, hc))
            store.declare("critic/q+critic/twin_q|fc3/bias", (2, 4))        # [g,0] is the bias, rest padding
            self.actor_trunk = DenseStack(store, ["actor"], self.S, self.actor_net_shape)
            self.build_policy_params(store, self.actor_net_shape[-1])

        def _init_weights(self):
            self.q_stack.init_weights(self.rng, self.weight_std)
            ops.truncated_normal_(self.store["critic/q+critic/twin_q|fc3/weight"], self.weight_std, self.rng)
            self.store["critic/q+critic/twin_q|fc3/bias"].zero_()
            self.actor_trunk.init_weights(self.rng, self.weight_std)
            self.init_policy_params()

        def _post_init(self):
            super()._post_init()
            lo = self.action_lower_bound if self.action_lower_bound is not None else -np.ones(self.n_actions)
            hi = self.action_upper_bound if self.action_upper_bound is not None else np.ones(self.n_actions)
            self.a_lo = torch.as_tensor(np.asarray(lo, np.float32), device=self.device)
            self.a_hi = torch.as_tensor(np.asarray(hi, np.float32), device=self.device)
            if self.trainable:
                # target net: its own copy of EVERY variable of the same layout + its own normaliser statistics
                self.target_store = ParamStore(self.device)
                for name, shape, _ in self.store.specs:
                    self.target_store.declare(name, shape)
                self.target_store.allocate(with_optimizer=False)
                self.target_norm = (torch.zeros(self.S, device=self.device), torch.ones(self.S, device=self.device))
                self.target_net = self
                self.init_ops.append(self.sync_target_net_hard)
                self.train_ops.insert(0, self.sync_target_net_soft)        # before the resample tick
                self._target_initialised, self._sync_now = False, True
                self.zero_log_alpha = torch.zeros(1, device=self.device)

        def critic_param_names(self):
            return [n for n in self.store.names() if "critic" in n]

        def actor_param_names(self):
            return [n for n in self.store.names() if "critic" not in n]

        def _named(self, src):
            out = {}
            for ref, (name, sel) in self.q_stack.reference_names().items():
                out[ref] = sel(src[name])
            for g, q in enumerate(("q", "twin_q")):
                out["critic/%s/fc3/weight" % q] = src["critic/q+critic/twin_q|fc3/weight"][g]
                out["critic/%s/fc3/bias" % q] = src["critic/q+critic/twin_q|fc3/bias"][g, :1]
            for ref, (name, sel) in self.actor_trunk.reference_names().items():
                out[ref] = sel(src[name])
            for n in ("actor/samples", "actor/samples_std", "actor/fc_policy/weight", "actor/fc_policy/bias"):
                out[n] = src[n]
            return out

        def variables(self, views=None):
            return self._named(self.store if views is None else views)

        def gradients(self):
            return self._named(self.store.views_of(self.store.grad))

        def target_variables(self):
            return self._named(self.target_store)

        # -- target network maintenance -----------------------------------------------------------------------
        def sync_target_net_hard(self):
            """init op (ddpg.py:110-121): copy once."""
            if not self._target_initialised:
                n = self.target_store.flat.numel()
                self.target_store.flat.copy_(self.store.flat[:n])
                self.target_norm[0].copy_(self.state_normalizer.mean)
                self.target_norm[1].copy_(self.state_normalizer.std)
                self._target_initialised = True

        def sync_target_net_soft(self):
            """train op (ddpg.py:97-109): on policy-delay steps theta' <- (1 - tau) theta' + tau theta, for every variable
            of the target net (the normaliser statistics included)."""
            if not self._sync_now:
                return
            n = self.target_store.flat.numel()
            kernels.soft_update(self.target_store.flat, self.store.flat[:n], self.tau)
            kernels.soft_update(self.target_norm[0], self.state_normalizer.mean, self.tau)
            kernels.soft_update(self.target_norm[1], self.state_normalizer.std, self.tau)

        # -- buffers ------------------------------------------------------------------------------------------
        def _alloc_policy_buffers(self, B):
            dev, A, AN = self.device, self.n_actions, self.n_actions * self.particles
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
            self.logits, self.logits2, self.xn2 = f(B, AN), f(B, AN), f(B, self.S)
            self.xa, self.xh, self.xa2 = f(B, self.S + A), f(B, self.S + A), f(B, self.S + A)
            self.q_pi, self.q_run, self.q_tgt = f(2, B), f(2, B), f(2, B)
            hc = self.critic_net_shape[-1]
            self.dh_run, self.dh_pi = f(2, B, hc), z(2, B, hc)
            self.dha = f(1, B, self.actor_net_shape[-1])
            self.zero_noise, self.zero_ba, self.zero_b = z(B, AN), z(B, A), z(B)
            self.dlogits, self.scr_ls = f(B, AN), f(A, self.particles)
            self.td3_out = {"dq_pi": f(2, B), "dq_run": f(2, B), "dlogp": f(B), "dlog_alpha": f(1), "loss3": f(3)}
            self.scratch_w3, self.scratch_b3 = f(hc), f(1)

        # -- rollout ------------------------------------------------------------------------------------------
        def _actor_logits(self, x, slot, out, params=None):
            ha = self.actor_trunk.forward(x, slot=slot, params=params)[0]
            src = self.store if params is None else params
            torch.addmm(src["actor/fc_policy/bias"], ha, src["actor/fc_policy/weight"], out=out)
            return ha

        def explore(self, a, sigma, noise=None):
            """a + sigma * N(0,1), clipped to the action box (ddpg.py:150-170)."""
            if noise is None:
                noise = torch.randn(a.shape, device=self.device, generator=self.rng)
            return torch.minimum(torch.maximum(a + sigma * noise, self.a_lo), self.a_hi)

        def run_batch(self, states, noise=None, running_update=True):
            st = self._to_dev(states).view(-1, self.S)
            self._ensure_buffers(st.shape[0])
            x = self._normalize(st)
            self._actor_logits(x, 0, self.logits)
            policy = MixtureGaussianDistribution(self.logits, self.loc, self.logstd, True)
            a = policy.mean()
            if self.trainable and running_update and self.max_active is not None:
                kernels.active_stats(policy.logits, self.max_active, self.sum_active)
            if self.random_action:
                a = self.explore(a, self.act_noise, noise)
            self.policy = policy
            return a, None, None

        def run(self, sess, state, ops=None):
            """reference ddpg.py:226-230: [action]."""
            action, _, _ = self.run_batch(np.asarray(state, dtype=np.float32)[None])
            return [action[0].cpu().numpy()]

        def evaluate(self, sess, state):
            raise NotImplementedError("TD3 has no state-value head; the DDPG worker never calls evaluate()")

        # -- learner step ---------------------------------------------------------------------------------------
        def _q_forward(self, xin, slot, out, target=False):
            src = self.target_store if target else self.store
            w3, b3 = src["critic/q+critic/twin_q|fc3/weight"], src["critic/q+critic/twin_q|fc3/bias"]
            h = self.q_stack.forward(xin, slot=slot, params=self.target_store if target else None)
            for g in range(2):
                kernels.value_head_fwd(h[g], w3[g], b3[g], out=out[g])
            return h

        def make_train_noise(self, B):
            return {"target": torch.randn(B, self.n_actions, device=self.device, generator=self.rng)}

        def compute_gradients(self, state, action, reward, not_terminal, state_, noise=None, sync=None):
            B, S, A = state.shape[0], self.S, self.n_actions
            st = self.store
            self._ensure_buffers(B)
            if not self._target_initialised:
                self.sync_target_net_hard()
            if sync is None:
                sync = self._policy_step()
            self._sync_now = bool(sync)
            noise = noise or self.make_train_noise(B)
            x = self._normalize(state)
            x2 = kernels.state_norm(state_, self.target_norm[0], self.target_norm[1], self.clip_state or 0.0, self.xn2)
            self.update_state_normalizer(state)
            ha = self._actor_logits(x, 0, self.logits)
            k, a_opt = kernels.mean_action(self.logits, self.loc, True)
            self._actor_logits(x2, 1, self.logits2, params=self.target_store)
            _, a2 = kernels.mean_action(self.logits2, self.target_store["actor/samples"], True)
            a2 = self.explore(a2, self.target_noise, noise["target"])
            self.last = {"a_opt": a_opt, "k": k, "a2": a2}
            kernels.sac_q_inputs(x, x2, a_opt, action, a2, self.xa, self.xh, self.xa2)     # three concats, one launch
            hq_pi = self._q_forward(self.xa, 0, self.q_pi)
            self.q_pi[1].copy_(self.q_pi[0])              # the policy loss uses Q alone: min(Q, Q) routes to row 0
            hq_run = self._q_forward(self.xh, 1, self.q_run)
            self._q_forward(self.xa2, 2, self.q_tgt, target=True)
            out = kernels.sac_loss(self.q_pi, self.q_run, self.q_tgt, self.zero_b, self.zero_b, reward, not_terminal,
                                   self.zero_log_alpha, self.gamma, self.value_loss_coef, 0.0, self.td3_out)
            # critic gradients: value loss through Q / twinQ (s, action_hist)
            gw3, gb3 = st.g("critic/q+critic/twin_q|fc3/weight"), st.g("critic/q+critic/twin_q|fc3/bias")
            gb3.zero_()
            w3 = st["critic/q+critic/twin_q|fc3/weight"]
            for g in range(2):
                kernels.value_head_bwd(hq_run[g], w3[g], out["dq_run"][g], self.dh_run[g], gw3[g], gb3[g, :1], relu6=True)
            self.q_stack.backward(self.dh_run, slot=1, need_dw=True)
            off, ln = st.span(self.actor_param_names())
            if not sync:                                  # policy-delay step: the actor's gradient is zero
                st.grad[off:off + ln].zero_()
                return out["loss3"]
            # actor gradients: through Q(s, a) to the action only (the critic's parameters are skipped)
            kernels.value_head_bwd(hq_pi[0], w3[0], out["dq_pi"][0], self.dh_pi[0], self.scratch_w3, self.scratch_b3,
                                   relu6=True)
            self.dh_pi[1].zero_()
            da = self.q_stack.backward(self.dh_pi, slot=0, need_dw=False, need_dx=(S, S + A))    # action columns only
            dloc = st.g("actor/samples")
            self.dlogits.zero_(); dloc.zero_(); self.scr_ls.zero_()
            kernels.rsample_bwd(self.logits, self.loc, self.logstd, self.zero_noise, self.zero_noise, da, self.zero_ba,
                                self.dlogits, dloc, self.scr_ls)
            st.g("actor/samples_std").zero_()             # fixed_sigma: not a trainable variable
            torch.mm(ha.t(), self.dlogits, out=st.g("actor/fc_policy/weight"))
            kernels.act_bwd_bias_grad(None, self.dlogits, st.g("actor/fc_policy/bias"), relu6=False)
            torch.mm(self.dlogits, self.policy_weight.t(), out=self.dha[0])
            self.actor_trunk.backward(self.dha, slot=0)
            return out["loss3"]

        def train_device(self, optimizer, state, action, reward, not_terminal, state_, noise=None):
            loss3 = self.compute_gradients(state, action, reward, not_terminal, state_, noise)
            optimizer.apply(self)
            return loss3

        def _policy_step(self):
            return self.global_step % self.policy_delay == 0 if self.policy_delay > 1 else True

        def _graph_phase(self):
            return (self._policy_step(),)

        def train(self, sess, optimizer, ops, state, action, reward, not_terminal, state_):
            """reference ddpg.py:232-243 + actor_critic.py:382-404.  Consecutive steps differ in structure (policy delay):
            after a few eager steps each of the two kinds is captured into a hipGraph of its own and replayed
            (graphed_train keys its graphs by _graph_phase)."""
            args = [self._to_dev(a) for a in (state, action, reward, not_terminal, state_)]
            args[0], args[1], args[4] = args[0].view(-1, self.S), args[1].view(-1, self.n_actions), args[4].view(-1, self.S)
            B = args[0].shape[0]
            self._sync_now = bool(self._policy_step())     # read by the delayed soft update after a replay
            loss3 = self.graphed_train(optimizer, args, lambda: self.make_train_noise(B)).cpu().numpy()
            loss = (float(loss3[0]), None, float(loss3[1]), float(loss3[2]))
            extra = []
            for o in (ops if hasattr(ops, "__len__") else ([ops] if ops else [])):
                extra.append(self.global_step if o in ("global_step", "step") else (o() if callable(o) else None))
            return loss, extra

    return DDPGNetwork


ParticleFilteringDDPGNetwork = ddpg_network_wrapper(ParticleFilteringA2CNetwork)
