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

        # -- rollout ---------------------------------------------------------------------------
        def run(self, sess, state, ops=None):
            """reference ppo.py:56-62: [action, action_log_prob, value] for one state (row 0)."""
            action, logp, value = self.run_one(state)
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
            deferred = self._deferred_jobs(B, fused_mixture=True)
            x = self.prologue(state, advantage, deferred=deferred)   # normalised state, state-norm EMA, advantage moments
            w3, b3 = st["critic/fc3/weight"], st["critic/fc3/bias"]
            if deferred is not None:
                ha, hc = self._trunks_forward(x, value_head=(w3, b3, self.value_buf), masks=True)
                v = self.value_buf
            else:
                ha, hc = self._trunks_forward(x)
                v = self._value(hc)
            policy = MixtureGaussianDistribution(self._policy_logits(ha), self.loc, self.logstd,
                                                 self.normalize_policy_output_)
            # without an entropy term dlogits is final when the fused kernel ends: it then also emits the
            # fc_policy bias gradient (column sums of dlogits), saving a 20 MB pass
            # (N > 64: the two-launch composition leaves dloc / dlogstd / dbias / loss3 final, nothing of it is deferred)
            composed = deferred is not None and self._mix_slabs.get("composed", False)
            fused_db = not self.entropy_beta
            gbias = st.g("actor/fc_policy/bias")
            slabs = deferred is not None and not composed
            out = kernels.ppo_mixture_fwd_bwd(policy.logits, action, self.loc, self.logstd, log_prob, advantage, v,
                                              value, self.adv_stats, self.epsilon, self.value_loss_coef,
                                              out=self.train_out, dbias=gbias if fused_db else None,
                                              finalize=not slabs, ws=self.ws_mix if slabs else None)
            dlogits = out["dlogits"]
            if slabs:
                # dloc / dlogstd / dbias / loss3 stay as per-block slabs until the step's single reduce launch
                AN = self.n_actions * self.particles
                nslab, stride = self._mix_slabs[fused_db]
                deferred.append(kernels.job_colsum(self.ws_mix, 0, nslab, stride, AN, out["dloc"].view(-1)))
                deferred.append(kernels.job_colsum(self.ws_mix, AN, nslab, stride, AN, out["dlogstd"].view(-1)))
                if fused_db:
                    deferred.append(kernels.job_colsum(self.ws_mix, 2 * AN, nslab, stride, AN, gbias.view(-1)))
                deferred.append(kernels.job_ppo_loss(self.ws_mix, stride - 4, nslab, stride, 1.0 / B,
                                                     self.value_loss_coef, out["loss3"]))
            if self.entropy_beta:
                self._entropy_term(policy, dlogits, B)
            self._backward_heads(ha, hc, dlogits, out["dvalue"], fused_db, deferred=deferred)
            if deferred is not None:
                self._finish_deferred(deferred, allow_norm=not self.fixed_sigma)
            if self.fixed_sigma:
                out["dlogstd"].zero_()
            return out["loss3"]

        def stage_next_prologue(self, state, action, value, log_prob, advantage):
            """multi-step captures: hand the NEXT step's prologue to this step's optimiser launch (graphs.py)"""
            self._next_prologue = self.prologue_rider(state, advantage)

        def train_device(self, optimizer, state, action, value, log_prob, advantage):
            """One learner step on device tensors; returns loss3 (device tensor, no sync)."""
            loss3 = self.compute_gradients(state, action, value, log_prob, advantage)
            optimizer.apply(self)
            return loss3

        def train(self, sess, optimizer, ops, state, action, value, log_prob, advantage):
            """reference ppo.py:64-72 + actor_critic.py:382-404.
            Returns ((loss, entropy-or-None, policy_loss, value_loss), extra_fetches)."""
            host = (state, action, value, log_prob, advantage)
            if not any(isinstance(a, torch.Tensor) for a in host):
                # the step of this minibatch size is already a hipGraph: the host arrays go STRAIGHT into its static inputs
                # (no staging tensors on the device, no device-to-device copies: five host-to-device copies and a replay)
                B = len(advantage)
                g = self._train_graphs.get((B, id(optimizer)) + tuple(self._graph_phase()))
                if g is not None and g.K == 1 and g.noise is None:
                    for dst, src in zip(g.inputs, host):
                        dst.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(src, dtype=np.float32))).view(dst.shape))
                    return self._report(g(), ops)
            args = [self._to_dev(a) for a in host]
            args[0] = args[0].view(-1, self.S)
            args[1] = args[1].view(-1, self.n_actions)
            loss3 = self.graphed_train(optimizer, args)
            return self._report(loss3, ops)

        ROUND_STEPS_PER_REPLAY = 8

        def _round_graph(self, optimizer, cols, B, K):
            """K consecutive minibatch steps of a training round as ONE hipGraph (pfpn_amd/graphs.py; consecutive
            one-step replays leave ~10 us of idle GPU between them).  Built once the one-step graph of this minibatch
            size exists (the first steps of a run stay eager, as in graphed_train); the staged rollout lives in static
            device columns the graph's gathers read, so a round of another size gets its own graph (at most two are
            kept).  PFPN_GRAPH_TRAIN=0 (no captured steps at all) keeps rounds eager too."""
            if (B, id(optimizer)) not in self._train_graphs:
                return None
            in_graph = False
            if getattr(optimizer, "exchange", optimizer.world > 1):
                # a K-step replay needs the collective INSIDE the graph (the default over RCCL); where the one-step graph of
                # this size fell back to two graphs around an eager all-reduce, DP rounds run one-step replays
                in_graph = getattr(self._train_graphs[(B, id(optimizer))], "in_graph_exchange", False)
                if not in_graph:
                    return None
            n = cols[0].shape[0]
            key = (B, n, K, id(optimizer))
            rg = self._round_graphs.get(key) if hasattr(self, "_round_graphs") else None
            if rg is None:
                if not hasattr(self, "_round_graphs"):
                    self._round_graphs = {}
                if len(self._round_graphs) >= 2:
                    return None
                from ...graphs import GraphedLearnerStep
                stage = [torch.empty_like(c) for c in cols]
                rows = torch.zeros(K, B, dtype=torch.int64, device=self.device)
                losses = torch.zeros(K, 3, dtype=torch.float32, device=self.device)
                sets = [tuple(c[:B].clone() for c in cols) for _ in range(K)]      # sane inputs for the warm-up steps
                rg = self._round_graphs[key] = GraphedLearnerStep(self, optimizer, steps_per_replay=K, input_sets=sets,
                                                                  gather_from=(stage, rows, losses),
                                                                  capture_exchange=in_graph)
            for dst, src in zip(rg.gather_from[0], cols):
                dst.copy_(src, non_blocking=True)
            return rg

        def train_round(self, sess_or_optimizer, ops, names, columns, plan):
            """A whole training round in one call (pfpn_amd/models/distributed_model.py: flat_train): `columns` are the
            rollout's host arrays by name (`names` = the learner's argument order), `plan` the row ids of every learner
            step.  ONE host-to-device copy per column, device-side gathers into the captured step's inputs, ONE read
            of all losses at the end.  Returns [((loss, entropy, policy_loss, value_loss), extra_fetches)] per step."""
            optimizer = sess_or_optimizer
            if self.entropy_beta or not plan:           # the entropy report reads the device every step anyway
                return [self.train(None, optimizer, ops, *[np.asarray(columns[k])[rows] for k in names])
                        for rows in plan]
            B0, K = len(plan[0]), self.ROUND_STEPS_PER_REPLAY
            n_rows = len(columns[names[0]])
            rg = getattr(self, "_round_graphs", {}).get((B0, n_rows, K, id(optimizer))) if len(plan) >= K else None
            if rg is not None and all(isinstance(columns[k], np.ndarray) for k in names):
                # the round's graph exists: the host columns go straight into the static columns its gathers read
                cols = rg.gather_from[0]
                for dst, k in zip(cols, names):
                    dst.copy_(torch.from_numpy(np.ascontiguousarray(columns[k], dtype=np.float32)).view(dst.shape))
            else:
                cols = [self._to_dev(columns[k]) for k in names]
                cols[0], cols[1] = cols[0].view(-1, self.S), cols[1].view(-1, self.n_actions)
                cols[2:] = [c.view(-1) for c in cols[2:]]
                rg = self._round_graph(optimizer, cols, B0, K) if len(plan) >= K else None
                if rg is not None:
                    cols = rg.gather_from[0]                # the staged rollout the graph's gathers read
            rows = torch.from_numpy(np.concatenate(plan).astype(np.int64)).to(self.device)
            losses = torch.empty(len(plan), 3, dtype=torch.float32, device=self.device)
            steps, pos, i = [], 0, 0
            while i < len(plan):
                if rg is not None and i + K <= len(plan) and all(len(p) == B0 for p in plan[i:i + K]):
                    if rg.can_replay():
                        # K minibatch steps per replay: their gathers read the staged rollout inside the graph
                        rg.gather_from[1].copy_(rows[pos:pos + K * B0].view(K, B0))
                        rg()
                        losses[i:i + K].copy_(rg.gather_from[2])
                        steps.extend(range(self.global_step - K + 1, self.global_step + 1))
                        pos += K * B0
                        i += K
                        continue
                ids = plan[i]
                losses[i].copy_(self.graphed_train(optimizer, cols, rows=rows[pos:pos + len(ids)]))
                pos += len(ids)
                steps.append(self.global_step)
                i += 1
            host = losses.cpu().numpy()
            kernels.check_faults()
            wanted = list(ops) if hasattr(ops, "__len__") else ([ops] if ops else [])
            return [((float(l[0]), None, float(l[1]), float(l[2])),
                     [st if o in ("global_step", "step") else (o() if callable(o) else None) for o in wanted])
                    for l, st in zip(host, steps)]

    return ClipPPONetwork


ParticleFilteringClipPPONetwork = clip_ppo_network_wrapper(ParticleFilteringA2CNetwork)
