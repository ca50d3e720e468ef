"""Soft actor-critic wrapper over the particle-filtering policy (off-policy resample path).

Host-side mirror of reference networks/actor_critic/sac.py: `sac_network_wrapper` (:12-175) -- twin Q
critics on concat(normalised state, action) (:107-121), target network sharing the actor and the state
normaliser (:141-153), learnable log_alpha with alpha = stop_gradient(exp(log_alpha)) (:36-39),
`vf_target = min(Q1,Q2) - alpha*logp` (:123-126), `q_target = r + gamma*not_terminal*V'(s')` (:132-139),
Q loss (:160-164), policy loss with target_entropy = -|A| (:166-173), soft target update tau (:67-73),
hard sync at init (:75-86), `run` -> [action] (:91-95), `train` feed (:97-105).
The policy sample is the Gumbel-softmax rsample branch of MixtureGaussianDistribution (utils.py:156-186).
"""

import numpy as np
import torch

from ... import kernels
from .. import ops
from ..ops import DenseStack, ParamStore
from ..utils import MixtureGaussianDistribution
from .a2c import ParticleFilteringA2CNetwork

__all__ = ["ParticleFilteringSACNetwork"]


class _TrainNoise(dict):
    """the SAC step's noise tensors by name; `refill(buf)` (set by the network) redraws all of them in one launch"""
    refill = None


def sac_network_wrapper(base_class):
    class AbstractSACNetwork(base_class):
        def __init__(self, alpha=0.2, tau=0.005, **kwargs):
            if "normalize_policy_output" not in kwargs:
                kwargs["normalize_policy_output"] = True          # sac.py:15-16
            super().__init__(**kwargs)
            self.alpha = alpha       # replaced by exp(log_alpha) for the trainable net (sac.py:36-39)
            self.batch_actor = True        # (tests flip these on an instance: the one-pass-per-input reference form)
            self.batch_critic = True
            self.fuse_actor = True         # sample + log-prob + critic inputs | their backward as ONE launch each
            self.ride_loss = True          # the loss head inside the first launch of the critics' backward pass
            self.pair_critics = True       # live and target critics layer by layer, one epilogue launch for both
            self.tau = tau
            self.reward, self.not_terminal = "reward", "not_terminal"

        # -- parameters: critic tensors first, then actor tensors (the two Adam groups) ------------
        def _declare(self, store):
            A = int(self.action_shape[0])
            self.trunk = None
            self.critic_trunk = None
            hc = self.critic_net_shape[-1]
            self.q_stack = DenseStack(store, ["critic/q1", "critic/q2"], self.S + A, self.critic_net_shape)
            store.declare("critic/q1+critic/q2|fc3/weight", (2, hc))
            store.declare("critic/q1+critic/q2|fc3/bias", (2, 4))      # [g,0] is the bias, rest padding
            self.actor_trunk = DenseStack(store, ["actor"], self.S, self.actor_net_shape)
            self.build_policy_params(store, self.actor_net_shape[-1])
            store.declare("alpha/log_alpha", (1,))

        def _init_weights(self):
            self.q_stack.init_weights(self.rng, self.weight_std)
            ops.truncated_normal_(self.store["critic/q1+critic/q2|fc3/weight"], self.weight_std, self.rng)
            self.store["critic/q1+critic/q2|fc3/bias"].zero_()
            self.actor_trunk.init_weights(self.rng, self.weight_std)
            self.init_policy_params()
            self.store["alpha/log_alpha"].zero_()                      # sac.py:37

        def _post_init(self):
            super()._post_init()
            st = self.store
            self.q_w3 = st["critic/q1+critic/q2|fc3/weight"]
            self.q_b3 = st["critic/q1+critic/q2|fc3/bias"]
            self.log_alpha = st["alpha/log_alpha"]
            # the noise draws' counter exists from the start (a stepper's state snapshot lists it; checkpoint.restore resets it)
            self.noise_epoch = torch.full((1,), int(self.global_step), dtype=torch.int64, device=self.device)
            if self.trainable:
                # target net = own copies of the Q tensors only; actor + normaliser are shared (sac.py:141-153)
                self.target_store = ParamStore(self.device)
                for name, shape, _ in st.specs:
                    if "critic" in name:
                        self.target_store.declare(name, shape)
                self.target_store.allocate(with_optimizer=False)
                self.critic_span = st.span(self.critic_param_names())
                self.target_net = self                                  # attribute read by the workers
                self.init_ops.append(self.sync_target_net_hard)
                self.train_ops.insert(0, self.sync_target_net_soft)    # before the resample tick
                self._target_initialised = False

        def critic_param_names(self):
            return [n for n in self.store.names() if "critic" in n]

        def actor_param_names(self):
            return [n for n in self.store.names() if "critic" not in n]

        def variables(self, views=None):
            src = self.store if views is None else views
            out = {}
            for ref, (name, sel) in self.q_stack.reference_names().items():
                out[ref] = sel(src[name])
            for g, q in enumerate(("q1", "q2")):
                out["critic/%s/fc3/weight" % q] = src["critic/q1+critic/q2|fc3/weight"][g]
                out["critic/%s/fc3/bias" % q] = src["critic/q1+critic/q2|fc3/bias"][g, :1]
            for ref, (name, sel) in self.actor_trunk.reference_names().items():
                out[ref] = sel(src[name])
            for n in ("actor/samples", "actor/samples_std", "actor/fc_policy/weight", "actor/fc_policy/bias",
                      "alpha/log_alpha"):
                out[n] = src[n]
            return out

        def gradients(self):
            st = self.store
            out = {}
            for ref, (name, sel) in self.q_stack.reference_names().items():
                out[ref] = sel(st.g(name))
            for g, q in enumerate(("q1", "q2")):
                out["critic/%s/fc3/weight" % q] = st.g("critic/q1+critic/q2|fc3/weight")[g]
                out["critic/%s/fc3/bias" % q] = st.g("critic/q1+critic/q2|fc3/bias")[g, :1]
            for ref, (name, sel) in self.actor_trunk.reference_names().items():
                out[ref] = sel(st.g(name))
            for n in ("actor/samples", "actor/samples_std", "actor/fc_policy/weight", "actor/fc_policy/bias",
                      "alpha/log_alpha"):
                out[n] = st.g(n)
            return out

        def target_variables(self):
            ts = self.target_store
            out = {}
            for ref, (name, sel) in self.q_stack.reference_names().items():
                out[ref] = sel(ts[name])
            for g, q in enumerate(("q1", "q2")):
                out["critic/%s/fc3/weight" % q] = ts["critic/q1+critic/q2|fc3/weight"][g]
                out["critic/%s/fc3/bias" % q] = ts["critic/q1+critic/q2|fc3/bias"][g, :1]
            return out

        # -- target network maintenance ---------------------------------------------------------------
        def sync_target_net_hard(self):
            """init op (sac.py:75-86): copy once."""
            if not self._target_initialised:
                off, ln = self.critic_span
                self.target_store.flat[:ln].copy_(self.store.flat[off:off + ln])
                self._target_initialised = True

        def sync_target_net_soft(self):
            """train op (sac.py:67-73): theta' <- (1-tau) theta' + tau theta."""
            off, ln = self.critic_span
            kernels.soft_update(self.target_store.flat[:ln], self.store.flat[off:off + ln], self.tau)

        def soft_update_args(self):
            """(target values, offset, length, tau) of the soft update over the flat parameter buffer: what the optimiser's
            Adam launch needs to do the update itself (kernels.clip_adam / adam_groups `soft=`)."""
            off, ln = self.critic_span
            return self.target_store.flat[:ln], off, ln, self.tau

        # -- buffers ----------------------------------------------------------------------------------
        def _alloc_policy_buffers(self, B):
            dev, A, AN = self.device, self.n_actions, self.n_actions * self.particles
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            # s and s' share one buffer each ([2B, .]: first half s, second half s'): the actor is evaluated on both in
            # ONE pass of 2B rows (compute_gradients); the halves are what the rest of the step names x / x2, logits / logits2
            self.xn_both, self.logits_both = f(2 * B, self.S), f(2 * B, AN)
            self.xn, self.xn2 = self.xn_both[:B], self.xn_both[B:]
            self.logits, self.logits2 = self.logits_both[:B], self.logits_both[B:]
            self.dlogits = f(B, AN)
            # Q(s, a_pi) and Q(s, action_hist) use the same (live) critic: their inputs, Q values and dL/dQ are the two
            # halves of [2B]-row buffers, so that the twin trunks can run ONCE over both (compute_gradients)
            self.xq_both, self.xa2 = f(2 * B, self.S + A), f(B, self.S + A)
            self.xa, self.xh = self.xq_both[:B], self.xq_both[B:]
            self.q_both, self.q_tgt = f(2, 2 * B), f(2, B)
            self.q_pi, self.q_run = self.q_both[:, :B], self.q_both[:, B:]
            hc = self.critic_net_shape[-1]
            self.dh_both = f(2, 2 * B, hc)
            self.dh_run, self.dh_pi = f(2, B, hc), f(2, B, hc)
            self.dha = f(1, B, self.actor_net_shape[-1])
            self.dq_both = f(2, 2 * B)
            self.sac_out = {"dq_pi": self.dq_both[:, :B], "dq_run": self.dq_both[:, B:], "dlogp": f(B),
                            "dlog_alpha": self.store.g("alpha/log_alpha"), "loss3": f(3)}
            # the two-pass form (`batch_critic = False`, undeferred steps) keeps contiguous [2, B] buffers of its own
            self.q_pi_c, self.q_run_c = f(2, B), f(2, B)
            self.sac_out_c = dict(self.sac_out, dq_pi=f(2, B), dq_run=f(2, B))
            self.scratch_w3, self.scratch_b3 = f(hc), f(1)
            self.ws_pbias = kernels.learner_workspace(B, AN, dev)      # private: deferred fc_policy bias gradient
            # private: the particle-grid partial sums of the log-prob / reparameterised-sample backward wait here for the
            # step's reduction launch
            self.ws_mixbwd = kernels.mixture_workspace(B, A, self.particles, dev)
            self.ws_rsbwd = kernels.rollout_workspace(B, A, self.particles, dev)
            self.ws_actor = kernels.sac_actor_workspace(B, A, self.particles, dev) \
                if kernels.sac_actor_supported(A, self.particles) else None

        # -- rollout ----------------------------------------------------------------------------------
        def _actor_logits(self, x, slot, out, masks=False):
            ha = self.actor_trunk.forward(x, slot=slot, masks=masks)[0]
            torch.addmm(self.policy_bias, ha, self.policy_weight, out=out)     # bias in the GEMM epilogue
            return ha

        def run_batch(self, states, noise=None, running_update=True):
            st = self._to_dev(states).view(-1, self.S)
            B = st.shape[0]
            self._ensure_buffers(B)
            x = self._normalize(st)
            self._actor_logits(x, 0, self.logits)
            policy = MixtureGaussianDistribution(self.logits, self.loc, self.logstd, True)
            upd = running_update and self.trainable and self.max_active is not None
            mx, sm = (self.max_active, self.sum_active) if upd else (None, None)
            if self.random_action:
                a, s_ = policy.sample(1, noise, self.rng, mx, sm)
            else:
                a = policy.mean()
                if upd:
                    kernels.active_stats(policy.logits, mx, sm)
            self.policy = policy
            return self.denormalize_action(a), None, None

        def run(self, sess, state, ops=None):
            """reference sac.py:91-95: [action]."""
            action, _, _ = self.run_batch(np.asarray(state, dtype=np.float32)[None])
            return [action[0].cpu().numpy()]

        def evaluate(self, sess, state):
            raise NotImplementedError("SAC has no state-value head; the SAC worker never calls evaluate()")

        # -- learner step -----------------------------------------------------------------------------
        def _q_forward(self, xin, slot, out, target=False, fold=False):
            params = self.target_store if target else None
            w3 = (self.target_store if target else self.store)["critic/q1+critic/q2|fc3/weight"]
            b3 = (self.target_store if target else self.store)["critic/q1+critic/q2|fc3/bias"]
            if fold:       # both Q heads inside the last trunk layer's epilogue pass; mask bits where a backward follows
                return self.q_stack.forward(xin, slot=slot, params=params, value_head=(-1, w3, b3, out),
                                            masks=not target)
            h = self.q_stack.forward(xin, slot=slot, params=params)
            for g in range(2):
                kernels.value_head_fwd(h[g], w3[g], b3[g], out=out[g])
            return h

        def _deferred_jobs(self, B, fused_mixture=False):
            """[] when the step's small reductions (bias gradients, both Q heads, the normaliser EMA) can wait for ONE
            launch at the end of the backward pass (`self.defer_reduce = False`: never), else None."""
            ok = self._defer_ok.get(B)
            if ok is None:
                ok = B % 4 == 0 and self.state_normalizer is not None and self.normalize_state and self.q_stack.L > 1 and \
                    self.defer_reduce
                self._defer_ok[B] = ok
            return [] if ok else None

        def make_train_noise(self, B):
            """the step's four draws (Gumbel + normal for s and for s') from ONE launch; `_buf` is the backing buffer; the
            dict's `refill` attribute redraws into it (graphs.refill_noise)."""
            AN = self.n_actions * self.particles
            buf = self._draw_train_noise(torch.empty(4, B, AN, dtype=torch.float32, device=self.device))
            noise = _TrainNoise({"gumbel": buf[0], "gumbel_": buf[1], "eps": buf[2], "eps_": buf[3], "_buf": buf})
            noise.refill = self._draw_train_noise
            return noise

        def _noise_counter(self):
            """the step's draw counter: an int64 device scalar the loss head advances by one per learner step and every
            noise draw reads when its launch RUNS -- so a captured draw (a rider of the optimiser launch, graphs.py) is
            fresh on every replay.  It starts at the learner step count and `checkpoint.restore` sets it to the restored
            count (a run resumed from a checkpoint does not replay the draws of the run's beginning)."""
            return self.noise_epoch

        def _draw_train_noise(self, buf):
            """fresh draws into buf [4, B, A*N] in ONE launch: a Philox block per element, keyed by the network generator's
            seed and the step's draw counter (no uniform fill + transform pass)."""
            return kernels.sac_noise_fill_(buf, self.rng.initial_seed(), 0, epoch=self._noise_counter())

        def stage_next_prologue(self, state, action, reward, not_terminal, state_):
            """captured steps (graphs.py): the NEXT step's prologue (s and s' normalised, the moments of s) as a rider of
            this step's optimiser launch"""
            B = state.shape[0]
            if self._deferred_jobs(B) is None:
                return
            self._next_prologue = self.prologue_rider(state, None, state2=state_, out2=self.xn2)

        def stage_next_noise(self, noise):
            """captured steps: the NEXT step's four noise tensors drawn by extra blocks of this step's optimiser launch"""
            buf = noise.get("_buf") if isinstance(noise, dict) else None
            if buf is None or not buf.is_contiguous() or buf.dim() != 3:
                return
            self._next_prologue = kernels.noise_rider(getattr(self, "_next_prologue", None), buf, self.rng.initial_seed(),
                                                      self._noise_counter())

        def compute_gradients(self, state, action, reward, not_terminal, state_, noise=None):
            B, S, A = state.shape[0], self.S, self.n_actions
            st = self.store
            self._ensure_buffers(B)
            if not self._target_initialised:
                self.sync_target_net_hard()
            noise = noise or self.make_train_noise(B)
            deferred = self._deferred_jobs(B)
            if deferred is not None:
                # normalise s AND s' + the moments of s in one launch, the EMA as a job
                x = self.prologue(state, deferred=deferred, state2=state_, out2=self.xn2)
                x2 = self.xn2
            else:
                x = self._normalize(state)
                x2 = self.state_normalizer.normalize(state_, self.clip_state or 0.0, out=self.xn2)
                self.update_state_normalizer(state)
            fold = deferred is not None
            loc, logstd = self.loc, self.logstd
            nb = noise.get("_buf")
            both = fold and self.batch_actor and nb is not None and nb.is_contiguous() and \
                tuple(nb.shape) == (4, B, A * self.particles) and x.data_ptr() == self.xn_both.data_ptr()
            actor_slot = 0
            fused = both and self.fuse_actor and self.ws_actor is not None
            if both:
                # eval actor on s and the shared actor on s' (sac.py:141-153) are the same network: ONE pass over the
                # 2B rows [s; s'] -- trunk, head, reparameterised sample and log-prob are 7 launches instead of 14; the
                # backward runs over the first B rows of that pass's activations
                hb = self._actor_logits(self.xn_both, 0, self.logits_both, masks=True)
                self.actor_trunk.share_prefix(2 * B, B, 0, 3)
                actor_slot, ha = 3, hb[:B]
                if fused:
                    # sample, log-prob at the sample and the three critic inputs: one launch (csrc/sac_actor_kernels.hip)
                    ab, sb, kb, lb = kernels.sac_actor_fwd(self.logits_both, loc, logstd, nb[0:2].view(2 * B, -1),
                                                           nb[2:4].view(2 * B, -1), x=self.xn_both, a_hist=action,
                                                           xq_both=self.xq_both, xq_next=self.xa2)
                else:
                    ab, sb, kb = kernels.rsample_fwd(self.logits_both, loc, logstd, nb[0:2].view(2 * B, -1),
                                                     nb[2:4].view(2 * B, -1))
                    lb = kernels.logprob_fwd(self.logits_both, sb, loc, logstd, True)
                a, s_, k, logp, a2, logp2 = ab[:B], sb[:B], kb[:B], lb[:B], ab[B:], lb[B:]
            else:
                ha = self._actor_logits(x, 0, self.logits, masks=fold)
                self._actor_logits(x2, 1, self.logits2)
                a, s_, k = kernels.rsample_fwd(self.logits, loc, logstd, noise["gumbel"], noise["eps"])
                logp = kernels.logprob_fwd(self.logits, s_, loc, logstd, True)
                a2, s2, _ = kernels.rsample_fwd(self.logits2, loc, logstd, noise["gumbel_"], noise["eps_"])
                logp2 = kernels.logprob_fwd(self.logits2, s2, loc, logstd, True)
            self.last = {"a": a, "s_": s_, "k": k, "logp": logp}
            # Q inputs
            if not fused:
                kernels.sac_q_inputs(x, x2, a, action, a2, self.xa, self.xh, self.xa2)
            # (the second half's rows must start on a row chunk of the partial sums: 16-row chunks up to 8192 rows)
            both_q = fold and self.batch_critic and B % 16 == 0 and 2 * B <= 8192
            if both_q:
                # the live critic on [(s, a_pi); (s, action_hist)] in ONE pass of 2B rows (4 launches instead of 8)
                # ... next to the target critics on (s', a'): GEMM, GEMM, ONE epilogue launch for both, per layer
                hw, hb = "critic/q1+critic/q2|fc3/weight", "critic/q1+critic/q2|fc3/bias"
                paired = self.pair_critics and self.q_stack.forward_pair(
                    self.xq_both, 0, (st[hw], st[hb], self.q_both),
                    self.xa2, 2, self.target_store, (self.target_store[hw], self.target_store[hb], self.q_tgt))
                if not paired:
                    self._q_forward(self.xq_both, 0, self.q_both, fold=True)
                q_pi, q_run, sout = self.q_pi, self.q_run, self.sac_out
            else:
                q_pi, q_run, sout = self.q_pi_c, self.q_run_c, self.sac_out_c
                hq_pi = self._q_forward(self.xa, 0, q_pi, fold=fold)
                hq_run = self._q_forward(self.xh, 1, q_run, fold=fold)
            if not (both_q and paired):
                self._q_forward(self.xa2, 2, self.q_tgt, target=True, fold=fold)
            self.last["q_pi"] = q_pi
            rider = ()
            if both_q and self.ride_loss:
                # the loss head rides in the first launch of the critics' backward pass (every block forms the dQ of its rows)
                out = sout
                rider = (kernels.sac_loss_rider(self.q_both, self.q_tgt, logp, logp2, reward, not_terminal, self.log_alpha,
                                                self.gamma, self.value_loss_coef, -float(A), self.dq_both, sout,
                                                noise_epoch=self._noise_counter()),)
            else:
                out = kernels.sac_loss(q_pi, q_run, self.q_tgt, logp, logp2, reward, not_terminal,
                                       self.log_alpha, self.gamma, self.value_loss_coef, -float(A), sout,
                                       noise_epoch=self._noise_counter())
            # critic gradients (value loss through Q(s, action_hist))
            gw3, gb3 = st.g("critic/q1+critic/q2|fc3/weight"), st.g("critic/q1+critic/q2|fc3/bias")
            if not fold:
                gb3.zero_()        # (folded: the reduction jobs assign [g, 0]; the padding of the fresh buffer stays zero)
            if both_q:
                # ONE backward pass over the 2B rows: the value loss reaches the critic's parameters through the second
                # half (weight / bias / head gradients from rows [B, 2B) only), the policy loss passes through the first
                # half to the action alone (Q parameters are skipped, sac.py:167): 6 launches instead of 9
                da = self.q_stack.backward(self.dh_both, slot=0, need_dw=True, need_dx=(S, S + A), deferred=deferred,
                                           value=(-1, self.q_w3, self.dq_both, gw3, gb3) + rider,
                                           dw_rows=(B, 2 * B), dx_rows=(0, B))
            elif fold:
                # both heads' backward inside the trunks' last-layer passes; bias-gradient sums become jobs
                self.q_stack.backward(self.dh_run, slot=1, need_dw=True, deferred=deferred,
                                      value=(-1, self.q_w3, out["dq_run"], gw3, gb3))
                da = self.q_stack.backward(self.dh_pi, slot=0, need_dw=False, need_dx=(S, S + A), deferred=deferred,
                                           value=(-1, self.q_w3, out["dq_pi"], None, None))
            else:
                for g in range(2):
                    kernels.value_head_bwd(hq_run[g], self.q_w3[g], out["dq_run"][g], self.dh_run[g], gw3[g],
                                           gb3[g, :1], relu6=True)
                self.q_stack.backward(self.dh_run, slot=1, need_dw=True)
                # actor gradients: through Q(s, a_pi) to the action only (Q parameters are skipped, sac.py:167)
                for g in range(2):
                    kernels.value_head_bwd(hq_pi[g], self.q_w3[g], out["dq_pi"][g], self.dh_pi[g], self.scratch_w3,
                                           self.scratch_b3, relu6=True)
                da = self.q_stack.backward(self.dh_pi, slot=0, need_dw=False, need_dx=(S, S + A))
            # the particle-grid gradients land straight in the flat gradient buffer: log-prob backward writes them, the
            # reparameterised-sample backward adds its part in place
            dloc, dlogstd = st.g("actor/samples"), st.g("actor/samples_std")
            gpb = st.g("actor/fc_policy/bias")
            if fused:
                # log-prob backward + sample backward + the particle-grid and head-bias partial sums: one launch; the three
                # slab sums are jobs of the step's reduction launch
                AN = A * self.particles
                dlogits = kernels.sac_actor_bwd(self.logits, loc, logstd, nb[0], nb[2], s_, out["dlogp"], da,
                                                self.dlogits, self.ws_actor)
                ns, ss = kernels.sac_actor_bwd_slabs(B, A, self.particles)
                deferred.append(kernels.job_colsum(self.ws_actor, 0, ns, ss, AN, dloc.view(-1)))
                deferred.append(kernels.job_colsum(self.ws_actor, AN, ns, ss, AN, dlogstd.view(-1)))
                deferred.append(kernels.job_colsum(self.ws_actor, 2 * AN, ns, ss, AN, gpb))
                torch.mm(ha.t(), dlogits, out=st.g("actor/fc_policy/weight"))
            else:
                wm, wr = (self.ws_mixbwd, self.ws_rsbwd) if fold else (None, None)
                dlogits, _, _, dxs = kernels.logprob_bwd(self.logits, s_, loc, logstd, out["dlogp"], True, True,
                                                         out={"dlogits": self.dlogits, "dloc": dloc, "dlogstd": dlogstd},
                                                         slabs_ws=wm)
                dlogits = dlogits.view(B, -1)
                kernels.rsample_bwd(self.logits, loc, logstd, noise["gumbel"], noise["eps"], da, dxs,
                                    dlogits, dloc, dlogstd, slabs_ws=wr)
                if fold:
                    # dloc / dlogstd = (slab sum of the log-prob backward) + (slab sum of the sample backward): two jobs of
                    # the reduction launch instead of two finalize launches (same sums, same order)
                    AN = dlogits.shape[1]
                    n1, s1 = kernels.logprob_bwd_slabs(self.logits, self.dlogits, B, A, self.particles)
                    n2, s2, o2 = kernels.rsample_bwd_slabs(B, A, self.particles)
                    deferred.append(kernels.job_colsum(wm, 0, n1, s1, AN, dloc.view(-1), second=(wr, o2, n2, s2)))
                    deferred.append(kernels.job_colsum(wm, AN, n1, s1, AN, dlogstd.view(-1), second=(wr, o2 + AN, n2, s2)))
                torch.mm(ha.t(), dlogits, out=st.g("actor/fc_policy/weight"))
                if fold:
                    cpg = kernels.act_bwd_partial(None, dlogits, 0, self.ws_pbias, relu6=False)
                    deferred.append(kernels.job_colsum(self.ws_pbias, 0, cpg, dlogits.shape[1], dlogits.shape[1], gpb))
                else:
                    kernels.act_bwd_bias_grad(None, dlogits, gpb, relu6=False)
            torch.mm(dlogits, self.policy_weight.t(), out=self.dha[0])
            self.actor_trunk.backward(self.dha, slot=actor_slot, deferred=deferred)
            if fold:
                self._finish_deferred(deferred, allow_norm=not self.fixed_sigma)
            return out["loss3"]

        def train_device(self, optimizer, state, action, reward, not_terminal, state_, noise=None):
            loss3 = self.compute_gradients(state, action, reward, not_terminal, state_, noise)
            optimizer.apply(self)
            return loss3

        def train(self, sess, optimizer, ops, state, action, reward, not_terminal, state_):
            """reference sac.py:97-105 + actor_critic.py:382-404."""
            args = [self._to_dev(a) for a in (state, action, reward, not_terminal, state_)]
            args[0] = args[0].view(-1, self.S)
            args[1] = args[1].view(-1, self.n_actions)
            args[4] = args[4].view(-1, self.S)
            B = args[0].shape[0]
            loss3 = self.graphed_train(optimizer, args, lambda: self.make_train_noise(B)).cpu().numpy()
            loss = (float(loss3[0]), None, float(loss3[1]), float(loss3[2]))
            extra = []
            for o in (ops if hasattr(ops, "__len__") else ([ops] if ops else [])):
                extra.append(self.global_step if o in ("global_step", "step") else (o() if callable(o) else None))
            return loss, extra

    return AbstractSACNetwork


ParticleFilteringSACNetwork = sac_network_wrapper(ParticleFilteringA2CNetwork)
