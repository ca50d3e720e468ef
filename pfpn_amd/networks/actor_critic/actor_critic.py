"""Actor-critic assembler for the MI355X build.

Host-side mirror of reference networks/actor_critic/actor_critic.py: `ActorCriticNetwork.__init__`
(:7-42), `init` (:74-184: state-norm -> actor trunk -> policy -> sampler -> critic -> losses),
`build_state_normalizer_op` (:223-244), `_run` (:368-380), `_train` (:382-404).

The TF graph becomes explicit forward/backward code over preallocated device buffers: GEMMs through
PyTorch-ROCm, everything else through the HIP kernels.  `sess` arguments are accepted and ignored.
Out-of-scope options of the reference (value normaliser, conv/lstm layers, regularisers, a shared
`common_net`) raise NotImplementedError instead of being silently ignored.
"""
import os

import numpy as np
import torch

from ... import kernels
from .. import ops
from ..ops import DenseStack, ParamStore
from ..utils import OnlineNormalizer


class ActorCriticNetwork(object):
    def __init__(self, trainable, state_shape, action_shape,
                 common_net_shape, critic_net_shape, actor_net_shape,
                 weight_initializer=None, activator="relu6",
                 normalize_state=False, clip_state=False,
                 normalize_value=False, clip_value=False,
                 normalize_advantage=False, clip_advantage=False,
                 critic_regularizer=None, actor_regularizer=None,
                 entropy_beta=None, value_loss_coef=0.5,
                 log=True, device=None, seed=None, **kwargs):
        self.trainable = trainable
        self.random_action = self.trainable
        self.state_shape = list(state_shape) if hasattr(state_shape, "__len__") else [state_shape]
        self.action_shape = list(action_shape) if hasattr(action_shape, "__len__") else [action_shape]
        if common_net_shape:
            raise NotImplementedError("common_net_shape: no shipped PFPN setting uses a shared trunk")
        if normalize_value or clip_value:
            raise NotImplementedError("value normaliser: normalize_value=False in every DeepMimic setting")
        if clip_advantage and hasattr(clip_advantage, "__len__") and len(clip_advantage) != 2:
            raise ValueError("clip_advantage is a bound c (-> [-c, c]) or a pair (actor_critic.py:160-168)")
        if critic_regularizer or actor_regularizer:
            raise NotImplementedError("L2 regularisers are unset in every DeepMimic setting")
        act_name = getattr(activator, "__name__", activator)
        if act_name not in ("relu6", None):
            raise NotImplementedError("only the ReLU6 activator of deepmimic_base.py:8 is built")
        for shp in (critic_net_shape, actor_net_shape):
            if any(hasattr(s, "__len__") for s in shp):
                raise NotImplementedError("conv layers are never instantiated by the PFPN settings")
        self.common_net_shape = []
        self.critic_net_shape = list(critic_net_shape)
        self.actor_net_shape = list(actor_net_shape)
        self.weight_initializer = weight_initializer      # None -> truncated_normal(0, 0.01)
        self.weight_std = 0.01
        self.normalize_state = normalize_state
        self.clip_state = clip_state
        self.normalize_advantage = normalize_advantage
        self.clip_advantage = clip_advantage or None
        self.entropy_beta = entropy_beta
        self.value_loss_coef = value_loss_coef
        self.init_ops, self.train_ops, self.running_update_ops = [], [], []
        self.local_update_variables = []
        self.summaries = {}
        self.log = log
        self.device = torch.device(device) if device is not None else None
        self.seed = seed
        self.global_step = 0

    # compatible with the reference's LSTM hooks
    def reset(self):
        pass

    def reset_running_state(self):
        pass

    def reset_training_state(self):
        pass

    def add_summary(self, key, val, val_wrapper=None):
        if self.log:
            self.summaries[key] = val

    # ------------------------------------------------------------------------------------------
    def init(self):
        if self.device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("pfpn_amd needs an MI355X (HIP) device; there is no CPU execution path")
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.S = int(np.prod(self.state_shape))
        self.rng = torch.Generator(device=self.device)
        self.rng.manual_seed(int(self.seed) if self.seed is not None else 0)
        self.store = ParamStore(self.device)
        self._declare(self.store)
        self.store.allocate(extra=self._bucket_extra(), with_optimizer=self.trainable)
        self._init_weights()
        self.state_normalizer = OnlineNormalizer(self.S, self.device) if (self.normalize_state or self.clip_state) else None
        self.step_dev = torch.zeros(1, dtype=torch.float32, device=self.device)   # float(global_step)
        self.dp_world = 1                      # set by the optimiser (SyncReplicasApplier)
        # ONE stream: every helper-stream overlap that was built (critic head / statistics beside the policy GEMMs, the
        # weight-gradient GEMMs, the state-norm EMA) measured 3-8 % SLOWER on MI355X and is gone (LAB_NOTEBOOK.md).
        # Structural switches of the learner step, fixed in the product; tests flip them on an instance to hold the fused
        # forms against their launch-per-piece reference arithmetic bit for bit:
        self.defer_reduce = True       # the step's small reductions wait for ONE launch (kernels.reduce_jobs)
        self.ride_prologue = True      # multi-step captures: the next step's prologue rides in the optimiser launch
        self._B = None
        self._buf_cache, self._buf_pinned = {}, set()
        self._train_graphs, self._train_calls = {}, {}
        # names of the reference's placeholders/tensors, kept for workers that look them up
        self.state, self.action, self.action_hist = "state", "action", "action_hist"
        self.advantage, self.value = "advantage", "value"
        self.loss = self.policy_loss = self.value_loss = None
        self.policy_entropy = None
        self._post_init()

    # hooks for subclasses ---------------------------------------------------------------------
    def _declare(self, store):
        """Declare parameters: twin actor/critic trunks when their shapes agree (all shipped
        settings: [1024, 512] both, deepmimic_base.py:4-6)."""
        if self.actor_net_shape == self.critic_net_shape:
            self.trunk = DenseStack(store, ["actor", "critic"], self.S, self.actor_net_shape)
            self.actor_trunk = self.critic_trunk = None
        else:
            self.trunk = None
            self.actor_trunk = DenseStack(store, ["actor"], self.S, self.actor_net_shape)
            self.critic_trunk = DenseStack(store, ["critic"], self.S, self.critic_net_shape)
        self.build_policy_params(store, self.actor_net_shape[-1] if self.actor_net_shape else self.S)
        hc = self.critic_net_shape[-1] if self.critic_net_shape else self.S
        store.declare("critic/fc3/weight", (hc, 1))
        store.declare("critic/fc3/bias", (1,))

    def _bucket_extra(self):
        return 2 * self.S if self.normalize_state else 0

    def _init_weights(self):
        for t in (self.trunk, self.actor_trunk, self.critic_trunk):
            if t is not None:
                t.init_weights(self.rng, self.weight_std)
        ops.truncated_normal_(self.store["critic/fc3/weight"], self.weight_std, self.rng)
        self.store["critic/fc3/bias"].zero_()
        self.init_policy_params()

    def _post_init(self):
        pass

    def build_policy_params(self, store, in_dim):
        raise NotImplementedError

    def init_policy_params(self):
        raise NotImplementedError

    # reference-named parameter access (checkpoint / tests) ---------------------------------------
    def variables(self, views=None):
        """{reference variable name: tensor view} for the trainables of the saved graph.
        `views`: named views of another flat buffer of the same layout (`store.views_of(store.m)`: Adam slots)."""
        src = self.store if views is None else views
        out = {}
        for t in (self.trunk, self.actor_trunk, self.critic_trunk):
            if t is not None:
                for ref, (name, sel) in t.reference_names().items():
                    out[ref] = sel(src[name])
        for n in self.store.names():
            if "|" not in n:
                out[n] = src[n]
        return out

    def gradients(self):
        out = {}
        for t in (self.trunk, self.actor_trunk, self.critic_trunk):
            if t is not None:
                for ref, (name, sel) in t.reference_names().items():
                    out[ref] = sel(self.store.g(name))
        for n in self.store.names():
            if "|" not in n:
                out[n] = self.store.g(n)
        return out

    def load_variables(self, values, views=None):
        v = self.variables(views)
        with torch.no_grad():
            for k, arr in values.items():
                v[k].copy_(torch.as_tensor(np.asarray(arr, dtype=np.float32)).to(self.device).view(v[k].shape))

    # forward pieces -------------------------------------------------------------------------------
    def _share_pins(self):
        for name in ("trunk", "actor_trunk", "critic_trunk", "q_stack"):
            st = getattr(self, name, None)
            if st is not None and hasattr(st, "pinned"):
                st.pinned = self._buf_pinned

    def _ensure_buffers(self, B, pin=False):
        """Activation / gradient buffers for a batch of B rows.  A worker alternates between sizes (B = 1 or E in
        the rollout, the minibatch size in the learner), so each size's buffers are kept: nothing is allocated
        in steady state, and a captured hipGraph (`pin=True`) keeps pointing at live memory."""
        if self._B == B:
            if pin:
                self._buf_pinned.add(B)
                self._share_pins()
            return
        cache = self._buf_cache
        if B in cache:
            self.__dict__.update(cache[B])
        else:
            before = dict(self.__dict__)
            dev = self.device
            self.xn = torch.empty(B, self.S, dtype=torch.float32, device=dev)
            self.value_buf = torch.empty(B, dtype=torch.float32, device=dev)
            if self.trainable and self.normalize_state and self.state_normalizer is not None:
                self.ws_moments = kernels.learner_workspace(B, self.S, dev)     # private: deferred normaliser EMA
            self._alloc_policy_buffers(B)
            cache[B] = {k: v for k, v in self.__dict__.items()
                        if k not in ("_B", "_buf_cache", "_buf_pinned") and (k not in before or before[k] is not v)}
            if len(cache) > 8:                       # bounded: drop the oldest size no graph depends on
                for old in list(cache):
                    if old != B and old not in self._buf_pinned:
                        del cache[old]
                        break
        if pin:
            self._buf_pinned.add(B)
            self._share_pins()
        self._B = B

    def _alloc_policy_buffers(self, B):
        pass

    def _normalize(self, state):
        """x = stop_gradient(clip((s - mean)/std, +-clip))  (actor_critic.py:76-80, :223-244)."""
        if self.state_normalizer is None:
            return state
        if self.normalize_state:
            return self.state_normalizer.normalize(state, self.clip_state or 0.0, out=self.xn)
        return torch.clamp(state, -self.clip_state, self.clip_state, out=self.xn)

    def _trunks_forward(self, x, value_head=None, masks=False):
        """-> (actor hidden [B,Ha], critic hidden [B,Hc]) (views of the stacks' activation buffers).
        `value_head` = (w, b, out): also evaluate the critic head inside the merged trunk's last epilogue;
        `masks`: hidden layers leave their ReLU6 masks as bits for the deferred backward."""
        if self.trunk is not None:
            h = self.trunk.forward(x, value_head=(1,) + tuple(value_head) if value_head is not None else None,
                                   masks=masks)
            return h[0], h[1]
        if value_head is not None:
            raise ValueError("the folded critic head needs the merged actor+critic trunk")
        return self.actor_trunk.forward(x)[0], self.critic_trunk.forward(x)[0]

    def _finish_deferred(self, deferred, allow_norm=True):
        """The step's single reduction launch.  When the optimiser left a norm hook (one replica, global-norm clip:
        SyncReplicasApplier), the same launch also produces the partial sums of the global gradient norm -- job outputs
        that are gradients are squared where they are produced, every other part of the flat gradient buffer is final by
        now (the last GEMM precedes this launch) and is covered by square-sum jobs -- and rotates the Adam powers, so the
        optimiser needs ONE launch (pfpn_clip_adam_prenorm) instead of two.  A step that takes this path MUST be followed
        by the optimiser's apply (the powers have moved)."""
        hook = getattr(self, "_norm_hook", None)
        self._norm_ready = 0
        if hook is None or not allow_norm:
            kernels.reduce_jobs(deferred)
            return
        st = self.store
        base, end = st.grad.data_ptr(), st.grad.data_ptr() + 4 * st.size
        cover = []
        for j in deferred:
            if j.kind in (kernels.JOB_COLSUM, kernels.JOB_VECSUM) and j.out is not None and base <= j.out < end:
                n = j.n if j.kind == kernels.JOB_COLSUM else 1
                if j.out % 16:                       # cannot express the complement in float4s: plain launch
                    kernels.reduce_jobs(deferred)
                    return
                j.norm = 1
                cover.append((j.out, min(end, (j.out + 4 * n + 15) // 16 * 16)))     # the padding behind a tensor is zero
        cover.sort()
        jobs, pos = list(deferred), base
        for lo, hi in cover + [(end, end)]:
            if lo > pos:
                nfl = (lo - pos) // 4
                # (one block per 16384 floats, at most 512: a 16.8-M-float head gradient -- 128 x 256 particles -- behind 64
                # blocks read its 67 MB at a quarter of the chip: 28 us for the launch; every segment up to 1 M floats keeps
                # the block count, hence the bits, it had)
                jobs.append(kernels.job_sqsum(pos, nfl, max(1, min(512, nfl // 16384))))
            pos = max(pos, hi)
        if len(jobs) > kernels.MAX_REDUCE_JOBS:
            for j in deferred:
                j.norm = 0
            kernels.reduce_jobs(deferred)
            return
        self._norm_ready = kernels.reduce_jobs(jobs, norm=hook)

    def _to_dev(self, a, dtype=torch.float32):
        if isinstance(a, torch.Tensor):
            return a.to(self.device, dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(self.device, dtype)

    def valid_data_mask(self, x):
        return x

    # -- data-parallel statistics (reference sync_model.py:37-46 push_weights) -----------------------
    def stat_slots(self):
        """[(tensor, offset in the bucket tail, length, computed_in_tail)] of the tensors that are
        averaged over the data-parallel replicas together with the gradients."""
        slots = []
        if self.normalize_state and self.trainable:
            nz = self.state_normalizer
            slots += [(nz.mean, 0, self.S, True), (nz.std, self.S, self.S, True)]
        return slots

    def _stat_attrs(self):
        """(owner, attribute name) of every tensor of stat_slots(), same order."""
        return [(self.state_normalizer, "mean"), (self.state_normalizer, "std")] \
            if (self.normalize_state and self.trainable) else []

    def adopt_tail_statistics(self, home=None):
        """Data-parallel runs: the replicated statistics LIVE in the all-reduce bucket's tail (the attributes become
        views of it), so nothing is packed before the all-reduce and one scale by 1/W afterwards turns the sums
        into the replica mean for all of them.
        `home` (sharded exchange): they live behind the PARAMETERS instead (store.flat_tail) -- the all-gather of the
        parameter slices hands every replica the owners' means -- while a replica's local contribution still travels in
        the bucket tail (the EMA writes it there, pack_local_stats copies the activation statistics)."""
        tail = self.store.tail if home is None else home
        for (t, off, n, _), (owner, attr) in zip(self.stat_slots(), self._stat_attrs()):
            view = tail[off:off + n].view(t.shape)
            view.copy_(t)
            setattr(owner, attr, view)
        self._stats_in_tail = home is None
        self._stats_in_flat = home is not None

    def pack_local_stats(self, tail):
        if getattr(self, "_stats_in_tail", False):
            return
        for t, off, n, in_tail in self.stat_slots():
            if not in_tail:
                tail[off:off + n].copy_(t.view(-1))

    def unpack_synced_stats(self, tail, inv_world):
        """after the all-reduce: statistics <- mean over replicas.  With one replica the state-normaliser
        EMA was written in place (update_state_normalizer) and nothing is left to do."""
        if inv_world == 1.0:
            return
        if getattr(self, "_stats_in_tail", False):
            tail.mul_(inv_world)
            return
        for t, off, n, in_tail in self.stat_slots():
            torch.mul(tail[off:off + n], inv_world, out=t.view(-1))

    def graphed_train(self, optimizer, args, make_noise=None, rows=None):
        """One learner step on the device tensors `args` (-> loss3, device); with `rows` (int64 device tensor) on
        those rows of `args` (the staged rollout of a training round: the gather lands in the graph's static inputs).  A worker trains on a fixed minibatch
        size (32 rows in the DPPO setting: ~40 launches of almost no work each), so after a few eager steps at
        one size the step is captured into a hipGraph (pfpn_amd/graphs.py) and replayed: identical results,
        a third of the time.  Data-parallel steps are captured with the RCCL all-reduce inside (or as two graphs around
        it: pfpn_amd/graphs.py).  PFPN_GRAPH_TRAIN=0 keeps the eager path."""
        B = args[0].shape[0] if rows is None else rows.shape[0]
        key = (B, id(optimizer)) + tuple(self._graph_phase())
        g = self._train_graphs.get(key)
        if rows is not None:
            if g is not None and make_noise is None:
                g.load_rows(args, rows)
                return g()
            args = [a.index_select(0, rows) for a in args]
        if g is None:
            n = self._train_calls[key] = self._train_calls.get(key, 0) + 1
            if n > 4 and self.device.type == "cuda" and len(self._train_graphs) < 4 and \
                    os.environ.get("PFPN_GRAPH_TRAIN", "1") != "0":
                from ...graphs import GraphedLearnerStep
                g = self._train_graphs[key] = GraphedLearnerStep(
                    self, optimizer, *[a.clone() for a in args], noise=make_noise() if make_noise else None)
        if g is not None:
            return g(batch=args)
        loss3 = self.compute_gradients(*args) if make_noise is None else self.compute_gradients(*args, make_noise())
        optimizer.apply(self)
        return loss3

    def _graph_phase(self):
        """what distinguishes learner steps of different launch structure at one batch size (each gets its own captured
        graph): nothing for most networks; TD3: whether this step updates the policy."""
        return ()

    def prologue(self, state, advantage=None, deferred=None, state2=None, out2=None):
        """First launches of a learner step: x = normalised state, the state-normaliser EMA on this minibatch
        (UPDATE_OPS, reference sync_model.py:123-138) and -- when `advantage` is given -- its normalisation
        moments into self.adv_stats.  With the normaliser on this is two launches (fused prologue + EMA
        finalisation) instead of four or more.  `state2` -> `out2` (deferred steps): a second batch normalised with the
        same moments in the same launch (SAC's next states)."""
        fusable = self.state_normalizer is not None and self.normalize_state and self.trainable
        if state2 is not None and not (fusable and deferred is not None):
            self.state_normalizer.normalize(state2, self.clip_state or 0.0, out=out2)
            state2 = None
        if not fusable:
            x = self._normalize(state)
            self.update_state_normalizer(state)
            if advantage is not None:
                kernels.adv_stats(advantage, self.normalize_advantage, out=self.adv_stats)
            return x
        nz, tail, S = self.state_normalizer, self.store.tail, self.S
        B = state.shape[0]
        # (sharded exchange: the statistics live behind the parameters, a replica's contribution goes to the bucket tail)
        in_place = (self.dp_world == 1 and not getattr(self, "_stats_in_flat", False)) or \
            getattr(self, "_stats_in_tail", False)
        if deferred is not None:
            # the moments stay in a private workspace; the EMA itself is one job of the step's single reduce launch
            # (its result is first read by the NEXT step's prologue)
            staged = getattr(self, "_prologue_staged", None)
            self._prologue_staged = None
            if staged is not None and staged == self._prologue_key(state, advantage, state2):
                pass        # this launch already ran, as a rider of the previous captured step's optimiser launch
            else:
                kernels.state_prologue(state, nz.mean, nz.std, self.clip_state or 0.0, self.xn, True, advantage,
                                       self.normalize_advantage, self.adv_stats if advantage is not None else None,
                                       ws=self.ws_moments, s2=state2, out_x2=out2)
            mo, so = (nz.mean, nz.std) if in_place else (tail[:S], tail[S:2 * S])
            deferred.append(kernels.job_norm_ema(self.ws_moments, B, S, self.step_dev, nz.mean, nz.std, mo, so))
            return self.xn
        kernels.state_prologue(state, nz.mean, nz.std, self.clip_state or 0.0, self.xn, True, advantage,
                               self.normalize_advantage, self.adv_stats if advantage is not None else None)
        if in_place:
            # in place: visible to the NEXT step's forward (x above used the old moments); with the statistics
            # resident in the bucket tail this IS the write into the tail
            kernels.state_norm_ema(B, S, self.step_dev, nz.mean, nz.std, nz.mean, nz.std)
        else:                           # into the bucket tail that the all-reduce averages
            kernels.state_norm_ema(B, S, self.step_dev, nz.mean, nz.std, tail[:S], tail[S:2 * S])
        return self.xn

    @staticmethod
    def _prologue_key(state, advantage, state2=None):
        return (state.data_ptr(), tuple(state.shape), advantage.data_ptr() if advantage is not None else 0,
                state2.data_ptr() if state2 is not None else 0)

    def prologue_rider(self, state, advantage=None, state2=None, out2=None):
        """Multi-step captures (pfpn_amd/graphs.py): the arguments of the NEXT step's deferred fused prologue, for the
        optimiser launch of THIS step to carry (kernels.clip_adam(rider=...)), or None where that prologue would not be the
        single fused launch.  The optimiser calls prologue_staged() once it has taken the rider; the next prologue()
        call on the same tensors then launches nothing (it still queues its EMA job)."""
        fusable = self.state_normalizer is not None and self.normalize_state and self.trainable and self.ride_prologue
        if not fusable or getattr(self, "ws_moments", None) is None or self.xn.shape[0] != state.shape[0]:
            return None
        nz = self.state_normalizer
        r = kernels.state_prologue_rider(state, nz.mean, nz.std, self.clip_state or 0.0, self.xn, advantage,
                                         self.normalize_advantage, self.adv_stats if advantage is not None else None,
                                         self.ws_moments, s2=state2, out_x2=out2)
        r._key = self._prologue_key(state, advantage, state2)
        return r

    def prologue_staged(self, rider):
        if rider._key is not None:
            self._prologue_staged = rider._key
        if rider.noise:
            # (SAC: the optimiser launch drew the next step's noise; graphs.py requires one taker per captured step)
            self._noise_rode = getattr(self, "_noise_rode", 0) + 1

    def update_state_normalizer(self, state):
        """UPDATE_OPS on the training minibatch (reference sync_model.py:123-138, utils.py:60-68), the launch-per-piece form
        (steps that cannot defer their reductions).  One replica: in place (visible to the NEXT step's forward, the
        deterministic choice of DESIGN.md section 5); W replicas: into the bucket tail that is averaged by the all-reduce."""
        if not (self.normalize_state and self.trainable):
            return
        nz, tail, S = self.state_normalizer, self.store.tail, self.S
        if (self.dp_world == 1 and not getattr(self, "_stats_in_flat", False)) or getattr(self, "_stats_in_tail", False):
            nz.updated(state, self.step_dev, nz.mean, nz.std)
        else:
            nz.updated(state, self.step_dev, tail[:S], tail[S:2 * S])

    def actor_param_names(self):
        return [n for n in self.store.names() if n.startswith("actor")]

    def critic_param_names(self):
        return [n for n in self.store.names() if n.startswith("critic")]
