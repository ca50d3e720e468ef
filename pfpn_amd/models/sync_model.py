"""Synchronous data-parallel strategy on MI355X: one process per GPU, RCCL all-reduce over xGMI.

Host-side mirror of reference models/sync_model.py: `SyncWorker.build_local_net` (:16-157) wraps the
optimiser in `tf.train.SyncReplicasOptimizer(replicas_to_aggregate=n_workers)` (:92-95), i.e. the
parameter server takes the MEAN of the per-worker gradients that were each clipped by global norm
on the worker (:64), the mean of the pushed statistic tensors (`push_weights`, :37-46: state-normaliser
mean/std, max/sum activation), applies Adam once, runs `train_ops` (:77-84) and the workers pull the
result (:27-32).

Here every rank keeps a replica.  One flat bucket [clipped grads | local stat tensors] is summed with a
single `torch.distributed.all_reduce` (backend "nccl" == RCCL) and divided by W inside the Adam kernel
(`grad_mul`); every rank then applies the identical TF-Adam update and resample tick, so replicas stay
bit-identical and no pull step is needed.  There is no parameter server and no gRPC.
"""
import os

import torch
import torch.distributed as dist

from .. import kernels
from .distributed_model import AbstractDistributedWorker, DistributedModel

__all__ = ["SyncModel", "SyncReplicasApplier"]


def _host_staged(t, group):
    """gloo moves all-to-all / all-gather through host memory only: GPU tensors are staged (the multi-rank tests that share
    one GPU run over gloo; RCCL takes device tensors as they are)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_to_all(out, inp, group):
    if _host_staged(inp, group):
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_to_all_single(o, i, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, group=group)


def _all_gather(out, inp, group):
    if _host_staged(inp, group):
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_gather_into_tensor(o, i, group=group)
        out.copy_(o)
    elif dist.get_backend(group) == "gloo":
        dist.all_gather_into_tensor(out, inp.clone(), group=group)      # (`inp` may be a slice of `out`: RCCL gathers in place)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


class SyncReplicasApplier(object):
    """clip (per replica) -> mean over replicas -> TF-Adam -> statistics -> train_ops.

    `groups`: list of dict(names=[...] or None for the whole buffer, lr=float): one Adam (own
    beta powers) per group, applied in order (SAC: critic list first, then actor list --
    reference base_worker.py:33-45)."""

    def __init__(self, net, groups, norm_clip=None, beta1=0.9, beta2=0.999, epsilon=1e-8,
                 process_group=None, world_size=None, force_exchange=False, exchange_mode=None, fused_norm=True):
        """`fused_norm=False` (tests): the global norm's partial sums come from a launch of their own instead of the
        step's reduction launch -- the same arithmetic in another grouping, the reference form of the bitwise tests."""
        self.net = net
        st = net.store
        self.norm_clip = norm_clip
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon
        self.pg = process_group
        if world_size is None:
            world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.world = int(world_size)
        # the exchange path (clip in place, bucket all-reduce, Adam with 1/W, statistics in the bucket tail) is taken
        # for W > 1; `force_exchange` takes it with ONE replica too (a world-size-1 group: the RCCL call, its stream
        # ordering against the kernels and the tail-resident statistics can then be checked on a single GPU)
        self.exchange = self.world > 1 or bool(force_exchange)
        net.dp_world = self.world
        self.norm_scale = torch.zeros(2, dtype=torch.float32, device=st.device)
        # beta powers of all groups in one [G,4] tensor: {b1^t, b2^t as TF's beta*_power hold them | the pair the
        # last fused apply used}; a group's "beta_pow" is the view of the first pair
        self.beta_pow = torch.tensor([[beta1, beta2, beta1, beta2]] * len(groups), dtype=torch.float32,
                                     device=st.device)
        self.groups = []
        for gi, g in enumerate(groups):
            if g.get("names") is None:
                off, ln = 0, st.size
            else:
                off, ln = st.span(g["names"])
            self.groups.append({"off": off, "len": ln, "lr": float(g["lr"]), "beta_pow": self.beta_pow[gi, :2]})
        import ctypes
        G = len(self.groups)
        self._g_off = (ctypes.c_longlong * G)(*[g["off"] for g in self.groups])
        self._g_len = (ctypes.c_longlong * G)(*[g["len"] for g in self.groups])
        self._g_lr = (ctypes.c_float * G)(*[g["lr"] for g in self.groups])
        self.fused = True
        # global-norm clip: the step's reduction launch can also produce the norm's partial sums and rotate the powers
        # (ActorCriticNetwork._finish_deferred), leaving ONE launch for clip + Adam (one replica) or for the in-place clip
        # (exchange path).
        net._norm_hook = None
        if self.norm_clip and len(self.groups) <= 4 and st.device.type == "cuda" and fused_norm:
            net._norm_hook = {"partial": torch.zeros(4096, dtype=torch.float32, device=st.device),
                              "beta_pow": self.beta_pow, "beta1": beta1, "beta2": beta2}
        # SAC: the soft target update (a train op after the apply, sac.py:67-73) becomes part of the optimiser's device
        # work -- inside the Adam launch where that is one launch, else right behind it -- instead of a launch of its own
        # after every graph replay
        self._soft_net = hasattr(net, "soft_update_args") and (net.sync_target_net_soft in net.train_ops or
                                                               getattr(net, "_soft_by_optimizer", False))
        if self._soft_net and not getattr(net, "_soft_by_optimizer", False):
            net.train_ops.remove(net.sync_target_net_soft)
            net._soft_by_optimizer = True
        # Exchange form (PFPN_EXCHANGE): "allreduce" (default) = ONE all-reduce of the bucket of per-replica clipped
        # gradients + statistics, every replica applies Adam to the whole buffer; "sharded" = all-to-all of UNSCALED bucket
        # slices to their owners (the bytes do not wait for the norm: they can leave while the rest of the backward runs),
        # all-gather of the W clip factors, each owner forms sum_w s_w g_w / W in replica order and runs Adam on 1/W of
        # the buffer, in-place all-gather of the parameter + statistics slices.  Same semantics
        # (sync_model.py:64-95), same results up to the order of W additions; replicas stay bit-identical.
        mode = os.environ.get("PFPN_EXCHANGE", "allreduce") if exchange_mode is None else exchange_mode
        if mode not in ("allreduce", "sharded"):
            raise ValueError("PFPN_EXCHANGE is 'allreduce' or 'sharded'")
        self.sharded = bool(self.exchange and mode == "sharded")
        if self.sharded:
            W = self.world
            if st.npad % (4 * W) or not self._fused_ok() or self._soft_net or st.device.type != "cuda":
                raise ValueError("the sharded exchange needs a replica count that divides the padded buffer into 16-byte "
                                 "slices, a global-norm clip, <= 4 Adam groups and no soft target network")
            self.shard_len = st.npad // W
            self.recv = torch.zeros(st.npad, dtype=torch.float32, device=st.device)
            self.scales = torch.ones(W, dtype=torch.float32, device=st.device)
            self.rank = dist.get_rank(process_group) if dist.is_available() and dist.is_initialized() else 0
            if hasattr(net, "adopt_tail_statistics"):
                net.adopt_tail_statistics(home=st.flat_tail)
        elif self.exchange and hasattr(net, "adopt_tail_statistics"):
            net.adopt_tail_statistics()        # replicated statistics live in the all-reduce bucket from now on

    # -- device part (no host sync) ------------------------------------------------------------------------
    # One replica: everything is ONE capturable sequence (apply_device).  Data parallel: the step splits at the
    # all-reduce into `pre_reduce` (per-replica clip, in place) | `reduce` (ONE RCCL all-reduce of the bucket) |
    # `post_reduce` (Adam on the sum with 1/W folded in, statistics <- replica mean); the two halves are captured
    # into hipGraphs of their own (pfpn_amd/graphs.py) and the collective is enqueued between their replays.
    def _fused_ok(self):
        return bool(self.norm_clip) and self.fused and len(self.groups) <= 4

    def _prenorm(self, net):
        ready = getattr(net, "_norm_ready", 0)
        net._norm_ready = 0
        return (net._norm_hook["partial"], ready) if ready and net._norm_hook else None

    def apply_device(self, net):
        st = net.store
        if self.exchange:
            self.pre_reduce(net)
            self.reduce(net)
            self.post_reduce(net)
            self.finish_reduce(net)
            return
        soft = net.soft_update_args() if self._soft_net else None
        if self._fused_ok():
            # clip + every group's Adam + power / step-counter advance in one launch (two without the norm hook)
            pre = self._prenorm(net)
            # (multi-step captures: the next step's prologue as extra blocks of this launch -- graphs.py stages it)
            rider = getattr(net, "_next_prologue", None)
            net._next_prologue = None
            if rider is not None and pre is None:
                rider = None
            kernels.clip_adam(st.flat, st.grad, st.m, st.v, self._g_off, self._g_len, self._g_lr, self.beta_pow,
                              self.norm_clip, self.norm_scale, self.beta1, self.beta2, self.epsilon,
                              step_counter=net.step_dev, prenorm=pre, soft=soft if pre is not None else None, rider=rider)
            if rider is not None:
                net.prologue_staged(rider)
            if soft is not None and pre is None:
                net.sync_target_net_soft()
            net.unpack_synced_stats(st.tail, 1.0)
            return
        fused_scale = None
        if self.norm_clip:
            # tf.clip_by_global_norm over ALL gradients jointly (base_worker.py:97-102); scale folded into Adam
            kernels.clip_by_global_norm(st.grad, self.norm_clip, self.norm_scale, apply=False)
            fused_scale = self.norm_scale
        self._adam_groups_unfused(net, fused_scale, 1.0)
        if soft is not None:
            net.sync_target_net_soft()
        net.unpack_synced_stats(st.tail, 1.0)

    def pre_reduce(self, net):
        """per-replica global-norm clip of the local gradients, in place (the bucket then holds what
        `AccumulatorApplyGradient_i` receives in the reference's graph).  Sharded exchange: only the clip FACTOR is
        formed (norm_scale[1]); the gradients stay as they are and the activation statistics join them in the tail."""
        st = net.store
        if self.sharded:
            kernels.clip_scale_only(st.grad, self.norm_clip, self.norm_scale, self.beta_pow, self.beta1, self.beta2,
                                    prenorm=self._prenorm(net))
            net.pack_local_stats(st.tail)
            return
        if self._fused_ok():
            kernels.clip_scale_inplace(st.grad, self.norm_clip, self.norm_scale, self.beta_pow, self.beta1, self.beta2,
                                       prenorm=self._prenorm(net))
        elif self.norm_clip:
            kernels.clip_by_global_norm(st.grad, self.norm_clip, self.norm_scale, apply=True)

    def post_reduce(self, net):
        """Adam on the all-reduced sum with 1/W folded in; the statistics tail becomes the replica mean."""
        st, W = net.store, self.world
        if self.sharded:
            L = self.shard_len
            kernels.shard_adam(st.flatx, st.m, st.v, self.recv, self.scales, W, self.rank * L, L, st.size,
                               st.size + st.extra, self._g_off, self._g_len, self._g_lr, self.beta_pow, self.beta1,
                               self.beta2, self.epsilon, step_counter=net.step_dev)
            return
        soft = net.soft_update_args() if self._soft_net else None
        if self._fused_ok():
            in_tail = getattr(net, "_stats_in_tail", False) and st.tail.numel() > 0
            kernels.adam_groups(st.flat, st.grad, st.m, st.v, self._g_off, self._g_len, self._g_lr, self.beta_pow,
                                1.0 / W, self.beta1, self.beta2, self.epsilon, step_counter=net.step_dev,
                                scale=st.tail if in_tail else None, scale_mul=1.0 / W, soft=soft)
            if not in_tail:
                net.unpack_synced_stats(st.tail, 1.0 / W)
            return
        self._adam_groups_unfused(net, None, 1.0 / W)
        if soft is not None:
            net.sync_target_net_soft()
        net.unpack_synced_stats(st.tail, 1.0 / W)

    def _adam_groups_unfused(self, net, fused_scale, grad_mul):
        st = net.store
        for gi, g in enumerate(self.groups):
            sl = slice(g["off"], g["off"] + g["len"])
            last = gi == len(self.groups) - 1
            kernels.tf_adam(st.flat[sl], st.grad[sl], st.m[sl], st.v[sl], g["beta_pow"], g["lr"], self.beta1,
                            self.beta2, self.epsilon, advance=True, norm_scale=fused_scale, grad_mul=grad_mul,
                            step_counter=net.step_dev if last else None)

    def reduce(self, net):
        """The exchange step: ONE all-reduce (sum) of the flat bucket [clipped grads | local statistics]
        (reference sync_model.py:92-95 accumulators 0-13 for the gradients, 14-17 for push_weights).
        Plain torch ops only, so the same code runs over gloo on CPU tensors in the tests."""
        st = net.store
        if getattr(self, "sharded", False):
            # slice j of every replica's bucket [unscaled grads | local statistics] to replica j; the W clip factors to all
            _all_to_all(self.recv, st.bucketx, self.pg)
            _all_gather(self.scales, self.norm_scale[1:2], self.pg)
            return
        net.pack_local_stats(st.tail)
        dist.all_reduce(st.bucket, op=dist.ReduceOp.SUM, group=self.pg)

    def finish_reduce(self, net):
        """sharded exchange: every replica receives the owners' parameter + statistics slices (in place: its own slice is
        already where it belongs).  No-op for the all-reduce form."""
        if getattr(self, "sharded", False):
            st, L = net.store, self.shard_len
            _all_gather(st.flatx, st.flatx[self.rank * L:(self.rank + 1) * L], self.pg)

    def gather_optimizer_state(self, net):
        """sharded exchange: a replica's Adam slots are current on ITS slice only; all-gather both (in place) so that every
        replica holds the complete slots (checkpoint writer, tests).  No-op otherwise."""
        if getattr(self, "sharded", False):
            st, L = net.store, self.shard_len
            for buf in (st.mx, st.vx):
                _all_gather(buf, buf[self.rank * L:(self.rank + 1) * L], self.pg)

    # -- host part -----------------------------------------------------------------------------------
    def apply_host(self, net):
        net.global_step += 1
        for op in net.train_ops:
            op()

    def apply(self, net):
        self.apply_device(net)
        self.apply_host(net)

    def state_dict(self):
        return {"beta_pow": [g["beta_pow"].clone() for g in self.groups]}


class SyncWorker(AbstractDistributedWorker):
    """reference sync_model.py:9-14: synchronous worker; `n_workers` = replicas to aggregate."""

    def __init__(self, n_workers=1, **kwargs):
        super().__init__(**kwargs)
        self.n_workers = n_workers
        self.n_aggregate = self.n_workers

    def build_local_net(self, name=None):
        """reference sync_model.py:16-157.  With all-reduce DP the local net IS the global net
        replica; build the optimiser around it."""
        self.target_net = self.global_net
        self.local_net = self.global_net
        if not self.global_net.trainable:
            return
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # replicas start identical (the reference's initial pull_weights, distributed_model.py:195-197)
            dist.broadcast(self.global_net.store.flat, src=0)
        self.opt_op = self.build_optimizer(self.global_net)


class SyncModel(DistributedModel):
    def worker_wrapper_kwargs(self, world_size):
        return {"n_workers": world_size}
