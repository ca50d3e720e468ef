"""A2C base + the particle-filtering policy head on MI355X.

Host-side mirror of reference networks/actor_critic/a2c.py: `A2CNetwork` (:16-89: GAE :30-40,
run/evaluate :67-78), the sampler/log-prob plumbing of `ContinuousA2CNetwork` (:205-243) and
`ParticleFilteringA2CNetwork` (:310-559: particle grid :476-534, fc_policy :541-558, activation
statistics :346-365, resample tick :367-383, resample ops :385-474).
"""
import os

import numpy as np
import torch

from ... import kernels
from .. import ops
from ..utils import MixtureGaussianDistribution
from .actor_critic import ActorCriticNetwork

__all__ = ["ParticleFilteringA2CNetwork", "particle_grid"]


def particle_grid(n_actions, n, tanh_mode=False, init_sigma=None):
    """Initial particle locations / log-stds, reference a2c.py:476-523 (bounds forced to [-1, 1],
    :479-480).  float64 on the host like the reference's NumPy, then fp32 variables."""
    u = np.ones((n_actions, n))
    l = -u
    ar = np.expand_dims(np.arange(n), 0)
    if tanh_mode:
        loc = l + (u - l) / n * (ar + 0.5)
    else:
        loc = l + (u - l) / (n - 1) * ar
    if init_sigma:
        std = np.zeros_like(loc) + np.reshape(np.asarray(init_sigma, dtype=np.float64), (-1, 1) if np.ndim(init_sigma) else ())
        if tanh_mode:
            loc_ = loc
            loc = np.arctanh(loc)
            std = np.maximum(loc - np.arctanh(np.maximum(1e-6 - 1, np.subtract(loc_, std))),
                             np.arctanh(np.minimum(1 - 1e-6, np.add(loc_, std))) - loc)
    else:
        std = (u - l) / (n - 1)
        if tanh_mode:
            assert n > 3
            loc = np.arctanh(loc)
            left = loc - loc[:, np.maximum(0, np.arange(n) - 1)]
            right = loc[:, np.minimum(n - 1, np.arange(n) + 1)] - loc
            std = np.maximum(left, right)
    return loc.astype(np.float32), np.log(std).astype(np.float32)


class A2CNetwork(ActorCriticNetwork):
    def __init__(self, gamma=0.99, lambd=0.95, **kwargs):
        super().__init__(**kwargs)
        self.gamma = gamma
        self.gae_gamma = None if lambd is None else self.gamma * lambd

    def generalized_advantage_estimate(self, reward, value, normalize=False):
        """reference a2c.py:30-40 + utils.discount :5-15 on the device: fp32 deltas, float64 reverse
        scan, fp32 result (bit-exact against the reference's NumPy loop)."""
        assert len(value) == len(reward) + 1
        if normalize:
            raise NotImplementedError("normalize=True is never used by the PFPN workers")
        T = len(reward)
        if T == 0:
            return []
        r = torch.from_numpy(np.asarray(reward, dtype=np.float32)).to(self.device)
        v = torch.from_numpy(np.asarray(value, dtype=np.float32)).to(self.device)
        offs = torch.tensor([0, T], dtype=torch.int64, device=self.device)
        adv = kernels.gae(r, v, offs, self.gamma, self.gae_gamma if self.gae_gamma else 0.0)
        return list(adv.cpu().numpy())

    def gae_batch(self, reward, value, offsets):
        """Device-resident ragged GAE: reward [Ttot], value [Ttot+E], offsets [E+1] (all tensors)."""
        return kernels.gae(reward, value, offsets, self.gamma, self.gae_gamma if self.gae_gamma else 0.0)


class ParticleFilteringA2CNetwork(A2CNetwork):
    def __init__(self, action_lower_bound, action_upper_bound, init_sigma=None, fixed_sigma=False,
                 particles=50, resample=3, resample_interval=2000, resample_threshold=None,
                 normalize_policy_output=False, **kwargs):
        self.action_lower_bound = None if action_lower_bound is None else np.asarray(action_lower_bound, dtype=np.float64)
        self.action_upper_bound = None if action_upper_bound is None else np.asarray(action_upper_bound, dtype=np.float64)
        super().__init__(**kwargs)
        if len(self.action_shape) != 1:
            raise ValueError("Particle Filtering Policy Network only supports continuous action space.")
        self.n_actions = int(self.action_shape[0])
        self.particles = int(particles)
        self._defer_ok = {}
        self.dis_action_shape = [self.particles] * self.n_actions
        if resample and resample < 0 and resample != -1:
            raise ValueError("resample is -1 (categorical source draw) or k > 0 (top-k), a2c.py:399-408")
        if resample and resample > self.particles:
            raise ValueError("resample=k needs k <= particles (a2c.py:390)")
        self.resample = resample
        self.resample_interval = resample_interval
        self.resample_threshold = resample_threshold
        self.init_sigma = fixed_sigma if fixed_sigma and fixed_sigma is not True else init_sigma
        self.fixed_sigma = True if fixed_sigma else False
        if self.fixed_sigma and resample and not self.init_sigma:
            raise ValueError("fixed_sigma=True with resampling needs init_sigma (a2c.py:438-439 jitters with it)")
        self.normalize_policy_output = False                      # tanh squash after the mixture (unused)
        self.normalize_policy_output_ = bool(normalize_policy_output)   # tanh squash before the mixture
        self.train_flag = 0

    # -- parameters ------------------------------------------------------------------------------
    def build_policy_params(self, store, in_dim):
        A, N = self.n_actions, self.particles
        self.policy_in = in_dim
        store.declare("actor/samples", (A, N))
        store.declare("actor/samples_std", (A, N))
        store.declare("actor/fc_policy/weight", (in_dim, A * N))
        store.declare("actor/fc_policy/bias", (A * N,))

    def init_policy_params(self):
        loc, logstd = particle_grid(self.n_actions, self.particles, self.normalize_policy_output_, self.init_sigma)
        self.store["actor/samples"].copy_(torch.from_numpy(loc))
        self.store["actor/samples_std"].copy_(torch.from_numpy(logstd))
        ops.truncated_normal_(self.store["actor/fc_policy/weight"], self.weight_std, self.rng)
        self.store["actor/fc_policy/bias"].zero_()

    def _bucket_extra(self):
        extra = super()._bucket_extra()
        if self.trainable and self.resample:
            extra += 2 * self.n_actions * self.particles
        return extra

    def _post_init(self):
        A, N = self.n_actions, self.particles
        self.loc = self.store["actor/samples"]
        self.logstd = self.store["actor/samples_std"]
        self.policy_weight = self.store["actor/fc_policy/weight"]
        self.policy_bias = self.store["actor/fc_policy/bias"]
        self.resample_rng = torch.Generator(device=self.device)      # identical on every DP rank
        self.resample_rng.manual_seed(int(self.seed) if self.seed is not None else 0)
        if self.trainable and self.resample:
            self.max_active = torch.zeros(A, N, dtype=torch.float32, device=self.device)
            self.sum_active = torch.zeros(A, N, dtype=torch.float32, device=self.device)
            self.local_update_variables += ["max_active", "sum_active"]
            self.running_update_ops += ["update_max_active", "update_sum_active"]
            self.train_ops.append(self.resample_tick)
        else:
            self.max_active = self.sum_active = None
        if self.normalize_state and self.trainable:
            self.local_update_variables = ["state_normalizer/mean", "state_normalizer/std"] + self.local_update_variables

    def _alloc_policy_buffers(self, B):
        dev, AN = self.device, self.n_actions * self.particles
        if not self.trainable:
            self.logits = torch.empty(B, AN, dtype=torch.float32, device=dev)
            return
        # (large shapes: logits and dlogits at the relative placement the streaming backward runs best at)
        self.logits, dlogits = kernels.stream_pair(B, AN, dev)
        hc = self.critic_net_shape[-1]
        self.train_out = {
            "logp": torch.empty(B, dtype=torch.float32, device=dev),
            "dlogits": dlogits,
            "dloc": self.store.g("actor/samples"),
            "dlogstd": self.store.g("actor/samples_std"),
            "dvalue": torch.empty(B, dtype=torch.float32, device=dev),
            "loss3": torch.empty(3, dtype=torch.float32, device=dev),
        }
        self.adv_stats = kernels.new_adv_stats(dev, self.clip_advantage)     # {mean, rstd | clip bounds}
        self.ws_mix = kernels.mixture_workspace(B, self.n_actions, self.particles, dev)   # private: deferred finalize
        self._mix_slabs = {}
        self._pbias_ws = {}
        if self.trunk is not None:
            self.dh = torch.empty(2, B, hc, dtype=torch.float32, device=dev)
        else:
            self.dh_a = torch.empty(1, B, self.actor_net_shape[-1], dtype=torch.float32, device=dev)
            self.dh_c = torch.empty(1, B, hc, dtype=torch.float32, device=dev)
        # per-size buffer like the rest (a captured learner step keeps pointing at it)
        self.entropy_buf = torch.full((B, self.n_actions), -float(self.entropy_beta) / B, dtype=torch.float32,
                                      device=dev) if self.entropy_beta else None

    def stat_slots(self):
        slots = super().stat_slots()
        if self.max_active is not None:
            AN = self.n_actions * self.particles
            base = 2 * self.S if (self.normalize_state and self.trainable) else 0
            slots += [(self.max_active, base, AN, False), (self.sum_active, base + AN, AN, False)]
        return slots

    def _stat_attrs(self):
        attrs = super()._stat_attrs()
        if self.max_active is not None:
            attrs += [(self, "max_active"), (self, "sum_active")]
        return attrs

    # -- forward -----------------------------------------------------------------------------------
    def _policy_logits(self, ha):
        """fc_policy: logits [B, A*N] = h W + b (a2c.py:541-554)."""
        # bias through the GEMM epilogue (hipBLASLt): no separate pass over the 20 MB of logits
        torch.addmm(self.policy_bias, ha, self.policy_weight, out=self.logits)
        return self.logits

    def _forward(self, state, x=None, with_value=False):
        """state [B,S] device tensor -> (policy distribution, actor hidden, critic hidden).
        `x`: the already normalised state (when the caller launched the normaliser itself).
        `with_value`: also leave the critic's value in `self.value_buf`, evaluated inside the merged trunk's last epilogue
        (same bits as `_value`, one launch less); `self._value_folded` says whether that happened."""
        B = state.shape[0]
        self._ensure_buffers(B)
        if x is None:
            x = self._normalize(state)
        self._value_folded = bool(with_value and self.trunk is not None)      # (B % 4 != 0: the stack launches the head itself)
        if self._value_folded:
            ha, hc = self._trunks_forward(x, value_head=(self.store["critic/fc3/weight"].view(-1),
                                                          self.store["critic/fc3/bias"], self.value_buf))
        else:
            ha, hc = self._trunks_forward(x)
        logits = self._policy_logits(ha)
        policy = MixtureGaussianDistribution(logits, self.loc, self.logstd, self.normalize_policy_output_)
        return policy, ha, hc

    def _value(self, hc):
        return kernels.value_head_fwd(hc, self.store["critic/fc3/weight"].view(-1), self.store["critic/fc3/bias"],
                                      out=self.value_buf)

    # -- learner step pieces shared by the PPO / A2C / V-trace heads -----------------------------------
    def _entropy_term(self, policy, dlogits, B, mask=None, n_valid=None):
        """policy_loss -= beta * mean_valid(sum_a H)  (actor_critic.py:172-176, a2c.py:62-65): adds the
        entropy gradient into dlogits; keeps the per-row entropies for the loss report."""
        if mask is None:
            dH = self.entropy_buf
        else:
            dH = ((-float(self.entropy_beta)) * mask / n_valid).view(B, 1).expand(B, self.n_actions).contiguous()
        self.last_entropy = policy.entropy()
        self.last_entropy_mask = mask
        policy.entropy_backward(dH, dlogits)

    def _entropy_value(self):
        H = self.last_entropy.sum(1)
        if self.last_entropy_mask is None:
            return float(H.mean().item())
        m = self.last_entropy_mask
        return float(((H * m).sum() / m.sum()).item())

    def _backward_heads(self, ha, hc, dlogits, dvalue, fused_db, deferred=None):
        """fc_policy and critic-head backward, then the trunks; gradients land in the flat gradient buffer.
        `deferred` (merged trunk only): the critic head's backward runs inside the trunk's last-layer pass and every
        bias-gradient reduction becomes a job of the caller's single `kernels.reduce_jobs` launch."""
        st = self.store
        w3 = st["critic/fc3/weight"].view(-1)
        dh_a, dh_c = (self.dh[0], self.dh[1]) if self.trunk is not None else (self.dh_a[0], self.dh_c[0])
        B = dlogits.shape[0]
        if deferred is not None:
            if not fused_db:
                # the head's bias gradient (column sums of dlogits) as per-chunk partials + a job of the reduction launch
                AN = dlogits.shape[1]
                ws = self._pbias_ws.get(B)
                if ws is None:
                    ws = self._pbias_ws[B] = kernels.learner_workspace(B, AN, dlogits.device)
                cpg = kernels.act_bwd_partial(None, dlogits, 0, ws, relu6=False)
                deferred.append(kernels.job_colsum(ws, 0, cpg, AN, AN, st.g("actor/fc_policy/bias").view(-1)))
            torch.mm(ha.t(), dlogits, out=st.g("actor/fc_policy/weight"))
            torch.mm(dlogits, self.policy_weight.t(), out=dh_a)
            self.trunk.backward(self.dh, deferred=deferred,
                                value=(1, w3, dvalue, st.g("critic/fc3/weight"), st.g("critic/fc3/bias")))
            return
        # launch-per-piece form (steps that cannot defer: separate trunks, row counts off the float4 grid)
        kernels.value_head_bwd(hc, w3, dvalue, dh_c, st.g("critic/fc3/weight").view(-1), st.g("critic/fc3/bias"), relu6=True)
        if not fused_db:
            kernels.act_bwd_bias_grad(None, dlogits, st.g("actor/fc_policy/bias"), relu6=False)
        torch.mm(ha.t(), dlogits, out=st.g("actor/fc_policy/weight"))
        torch.mm(dlogits, self.policy_weight.t(), out=dh_a)
        if self.trunk is not None:
            self.trunk.backward(self.dh)
        else:
            self.actor_trunk.backward(self.dh_a)
            self.critic_trunk.backward(self.dh_c)

    def _deferred_jobs(self, B, fused_mixture=False):
        """[] when this step can leave its small reductions (bias gradients, critic head, normaliser EMA and -- PPO --
        the fused mixture kernel's slabs) to ONE launch at the end of the backward pass: merged trunk, fused prologue
        (and the packed mixture kernel for `fused_mixture`).  Else None (`self.defer_reduce = False`: never)."""
        key = (B, fused_mixture)
        ok = self._defer_ok.get(key)
        if ok is None:
            ok = self.trunk is not None and self.trunk.L > 1 and B % 4 == 0 and self.state_normalizer is not None and \
                self.normalize_state and self.defer_reduce
            if ok and fused_mixture:
                try:
                    for want_db in (False, True):
                        self._mix_slabs[want_db] = kernels.ppo_mixture_slabs(
                            B, self.n_actions, self.particles, want_db, self.logits, self.train_out["dlogits"])
                    self._mix_slabs["composed"] = False
                except Exception:
                    # a particle grid outside the one-launch kernel (N > 64): the two-launch composition finishes its own
                    # sums; everything else of the step is deferred all the same
                    self._mix_slabs["composed"] = True
            self._defer_ok[key] = ok
        return [] if ok else None

    def _pg_gradients(self, state, action, targets):
        """Policy-gradient (A2C) loss: forward, `targets(logp, value) -> (value_target, advantage, mask)`,
        loss head, backward.  Returns loss4 (device): {loss, policy_loss, value_loss*coef, n_valid}."""
        B = state.shape[0]
        self._ensure_buffers(B)
        deferred = self._deferred_jobs(B)
        x = self.prologue(state, deferred=deferred)
        if deferred is not None:
            st = self.store
            ha, hc = self._trunks_forward(x, value_head=(st["critic/fc3/weight"], st["critic/fc3/bias"], self.value_buf),
                                          masks=True)
            v = self.value_buf
        else:
            ha, hc = self._trunks_forward(x)
            v = self._value(hc)
        policy = MixtureGaussianDistribution(self._policy_logits(ha), self.loc, self.logstd,
                                             self.normalize_policy_output_)
        logp = policy.log_prob(action)
        value_target, advantage, mask = targets(logp, v)
        if mask is not None and self.normalize_advantage:
            raise NotImplementedError("normalize_advantage over a valid_data_mask is not built "
                                      "(no shipped V-trace setting uses it)")
        kernels.adv_stats(advantage, self.normalize_advantage, out=self.adv_stats)
        dlogp, dvalue, loss4 = kernels.a2c_loss(logp, advantage, v, value_target, self.adv_stats,
                                                self.value_loss_coef, mask)
        dlogits, dloc, dlogstd = policy.log_prob_backward(action, dlogp)
        dlogits = dlogits.view(B, -1)
        self.store.g("actor/samples").copy_(dloc)
        self.store.g("actor/samples_std").copy_(dlogstd)
        if self.entropy_beta:
            self._entropy_term(policy, dlogits, B, mask, loss4[3] if mask is not None else None)
        if self.fixed_sigma:
            self.store.g("actor/samples_std").zero_()
        self._backward_heads(ha, hc, dlogits, dvalue, fused_db=False, deferred=deferred)
        if deferred is not None:
            self._finish_deferred(deferred, allow_norm=not self.fixed_sigma)
        self.last_targets = (value_target, advantage, mask, logp, v)
        return loss4

    def _report(self, loss4, ops):
        loss4 = loss4.cpu().numpy()
        kernels.check_faults()      # the host has just synchronised: a recorded kernel fault surfaces here, not as a NaN log line
        ent = None
        if self.entropy_beta:
            ent = self._entropy_value()
            loss4 = loss4.copy()
            loss4[1] -= self.entropy_beta * ent
            loss4[0] -= self.entropy_beta * ent
        loss = (float(loss4[0]), ent, float(loss4[1]), float(loss4[2]))
        extra = []
        for o in (ops if hasattr(ops, "__len__") else ([ops] if ops else [])):
            extra.append(self.global_step if o in ("global_step", "step") else (o() if callable(o) else None))
        return loss, extra

    def train(self, sess, optimizer, ops, state, action, value_target, advantage):
        """reference a2c.py:80-88 (A2C / A3C): policy-gradient loss on host-computed targets.
        Returns ((loss, entropy-or-None, policy_loss, value_loss), extra_fetches)."""
        st, ac, vt, adv = [self._to_dev(a) for a in (state, action, value_target, advantage)]
        loss4 = self._pg_gradients(st.view(-1, self.S), ac.view(-1, self.n_actions),
                                   lambda logp, v: (vt.view(-1), adv.view(-1), None))
        optimizer.apply(self)
        return self._report(loss4, ops)

    def denormalize_action(self, a):
        """reference a2c.py:246-261 scales only when `self.normalize_policy_output` is set, and the
        particle network forces that flag to False (a2c.py:327; its tanh squash is the
        `normalize_policy_output_` flag): the tanh action in (-1, 1) goes to the env unscaled.
        Reference behaviour kept as is."""
        return a

    def run_batch(self, states, noise=None, running_update=True):
        """Batched rollout inference for E environments at once: one launch sequence for
        sample + log-prob + value + activation statistics.  Returns device tensors
        (action [E,A], log_prob [E] or None, value [E] or None)."""
        st = self._to_dev(states).view(-1, self.S)
        policy, ha, hc = self._forward(st, with_value=self.trainable)
        upd = running_update and self.trainable and self.max_active is not None
        mx, sm = (self.max_active, self.sum_active) if upd else (None, None)
        logp = None
        if self.random_action:
            if self.normalize_policy_output_:
                a, s_ = policy.sample(1, noise, self.rng, mx, sm)
                action = self.denormalize_action(a)
                self._last_sample = (a, s_)
            else:
                action, logp = policy.sample(1, noise, self.rng, mx, sm, want_logp=True)
        else:
            action = self.denormalize_action(policy.mean())
            if upd:
                kernels.active_stats(policy.logits, mx, sm)
        self.policy = policy
        value = (self.value_buf if self._value_folded else self._value(hc)) if self.trainable else None
        return action, logp, value

    def run_one(self, state):
        """run_batch on ONE state, the reference worker's per-environment-step call.  After a few eager calls the
        launch sequence is captured once and replayed (pfpn_amd/graphs.py: GraphedRollout, ~95 vs ~200 us per call;
        bit-identical results); PFPN_GRAPH_RUN=0 keeps it eager."""
        st = np.asarray(state, dtype=np.float32)[None]
        g = getattr(self, "_run_graph", None)
        if g is None:
            self._run_calls = getattr(self, "_run_calls", 0) + 1
            if self._run_calls > 3 and self.device.type == "cuda" and os.environ.get("PFPN_GRAPH_RUN", "1") != "0":
                from ...graphs import GraphedRollout
                g = self._run_graph = GraphedRollout(self, 1)
        return g(st) if g is not None else self.run_batch(st)

    def run(self, sess, state, ops=None):
        """reference a2c.py:67-73 / actor_critic.py:368-380: one state -> [action, value]."""
        action, _, value = self.run_one(state)
        out = [action[0].cpu().numpy()]
        if self.trainable:
            out.append(float(value[0].item()))
        return out

    def evaluate(self, sess, state):
        """reference a2c.py:75-78: V(state) (no statistics update beyond what _run does)."""
        st = self._to_dev(np.asarray(state, dtype=np.float32)[None])
        _, _, hc = self._forward(st)
        if self.trainable and self.max_active is not None:
            # _run appends running_update_ops to every fetch (actor_critic.py:369-371)
            kernels.active_stats(self.logits, self.max_active, self.sum_active)
        return float(self._value(hc)[0].item())

    def evaluate_batch(self, states):
        """V(states [n,S]) as a device tensor: the bootstrap values of all rollout segments that end in one
        batched critic call (the per-state `evaluate` above, batched; same statistics side effect)."""
        st = self._to_dev(np.asarray(states, dtype=np.float32)).view(-1, self.S)
        if st.shape[0] == 0:
            return torch.empty(0, dtype=torch.float32, device=self.device)
        _, _, hc = self._forward(st)
        if self.trainable and self.max_active is not None:
            kernels.active_stats(self.logits, self.max_active, self.sum_active)
        return self._value(hc).clone()

    # -- resample ----------------------------------------------------------------------------------
    def steps_until_tick(self):
        """optimiser applies left until (and including) the one whose train op resamples (a2c.py:367-383)."""
        if not (self.trainable and self.resample):
            return 1 << 30
        return max(1, self.resample_interval - self.train_flag)

    def resample_tick(self):
        """train op run after every optimiser apply (a2c.py:367-383)."""
        self.train_flag += 1
        if self.train_flag >= self.resample_interval:
            self.resample_now()
            self.max_active.zero_()
            self.sum_active.zero_()
            self.train_flag = 0
            return True
        return False

    def resample_now(self, u_resample=None, u_noise=None):
        """build_resample_ops (a2c.py:385-474) on the device.  Adam slots are not touched."""
        A, N = self.n_actions, self.particles
        if u_resample is None:
            u_resample = torch.rand(A, N, dtype=torch.float64, device=self.device, generator=self.resample_rng)
            u_noise = torch.rand(A, N, dtype=torch.float32, device=self.device, generator=self.resample_rng) * 2 - 1
        thr = float(self.resample_threshold) if self.resample_threshold else 0.0
        self.last_resample = kernels.resample(self.loc, self.logstd, self.policy_bias, self.policy_weight,
                                              self.max_active, self.sum_active, u_resample, u_noise, thr,
                                              self.normalize_policy_output_,
                                              topk=self.resample if self.resample and self.resample > 0 else 0,
                                              fixed_sigma=float(self.init_sigma) if self.fixed_sigma else 0.0)
        return self.last_resample
