"""hipGraph capture of the launch-bound learner step.

One DPPO learner step is ~40 short launches (GEMMs + HIP kernels); at a 4096-row batch the host
dispatch cost is comparable to the device time, so the whole device part of the step
(compute_gradients + clip + TF-Adam + statistics) is captured once into a hipGraph and replayed.
Host-side bookkeeping (global_step, the resample tick every `resample_interval` steps) stays outside
the graph: the resample kernels run eagerly between replays.
"""
import gc
import os
import sys
import time
import weakref

import torch

from . import kernels

# Steppers whose hipGraphs hold captured RCCL collectives.  RCCL's communicator teardown waits for captured point-to-point
# plans (all_to_all_single) that a LIVE graph still references: `destroy_process_group()` then never returns
# (tools/capture_collective_repro.py, profiles/r04_capture_collective_repro.txt: only all-to-all, only with the graph alive;
# deleting the graph first cures it).  So the graphs go first: `release_captured_collectives()`, called by
# `models.distributed_model.shutdown_process_group()` and by bench.py before they destroy the group.
_with_collectives = weakref.WeakSet()
# tests only: the next captured-exchange attempt fails INSIDE the capture region (rehearses the fall-back) -- None / False: off;
# True: on every replica; an int r: only on the replica whose optimiser carries `rank_for_tests == r` (the others learn of it
# through the flag).  "invalidated": the failure carries the runtime's invalidated-capture signature (CaptureInvalidated path).
inject_capture_failure = None


def release_captured_collectives():
    """destroy every hipGraph that captured a collective (their steppers become unusable); returns how many"""
    n = 0
    for g in list(_with_collectives):
        g.close()
        n += 1
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return n


def _capture_mode(force=False):
    """Called right in front of a capture: the keyword arguments of torch.cuda.graph() -- thread-local error mode while a
    torch.distributed group lives in the process -- after letting the group's watchdog thread go quiet.  The watchdog polls the
    completion events of every collective that was enqueued eagerly (every 100 ms, until it has seen them complete); a capture
    that starts right behind the warm-up's collectives runs WHILE it is still querying them, and that combination is what the two
    invalidated captures of round 5 had in common (one in ~15 full test runs each: the watchdog's event query failing, a
    hipBLASLt call refusing inside the capture -- CaptureInvalidated).  Collectives issued inside a capture are not handed to the
    watchdog, so once its list is empty it stays empty for the capture's duration: synchronise, then give it three of its
    polling periods."""
    try:
        import torch.distributed as dist
        alive = dist.is_available() and dist.is_initialized()
    except Exception:
        alive = False
    if alive and torch.cuda.is_available():
        torch.cuda.synchronize()
        time.sleep(float(os.environ.get("PFPN_CAPTURE_QUIET_S", "0.3")))
    return {"capture_error_mode": "thread_local"} if (force or alive) else {}


class CaptureInvalidated(RuntimeError):
    """The HIP runtime invalidated a stream capture: some call inside it was not capturable (seen once in ~30 full test runs: a
    hipBLASLt matmul of a shape it had run eagerly just before returned HIPBLAS_STATUS_INTERNAL_ERROR inside the capture,
    `operation not permitted when stream is capturing`).  There is no falling back from THAT in this process: on this runtime
    (ROCm 7.0) hipStreamEndCapture of an invalidated capture leaves the capturing thread refusing every later kernel launch on
    the default stream (tools/capture_recover_probe.py) -- so the failure is reported as what it is instead of walking into the
    replicas' agreement all-reduce and dying there.  (A capture refused BEFORE anything illegal happened -- RCCL declining, a
    Python exception -- ends cleanly and does fall back: the two-graph form.)"""

    def __init__(self, cause):
        super().__init__("the runtime invalidated the stream capture (%s): this process cannot launch on its default stream any "
                         "more and must be restarted" % str(cause).split("\n")[0])


def capture_invalidated(e):
    t = str(e)
    return "stream is capturing" in t or "during capture" in t or "StreamCapture" in t


class ReplicaLost(RuntimeError):
    """Another replica published that it cannot go on (its stream capture was invalidated: CaptureInvalidated there)."""


_agreements = {}
_FATAL = b"X"


def any_replica(flag, group=None, device="cpu", fatal=False, timeout_s=300.0):
    """Logical OR of `flag` over the replicas of `group`, agreed through the job's rendezvous STORE (host side: it works for a
    replica whose GPU stream is unusable, and a replica waiting here is not parked inside a device collective).  Every replica
    of the group must reach the call the same number of times (agreement k pairs up with the others' agreement k).
    `fatal=True`: publish that this replica is lost and return at once (the caller raises); the others get ReplicaLost from
    their own call instead of waiting minutes for a collective's timeout.  Falls back to an all-reduce(MAX) on `device` where
    the store cannot be reached (never seen; the private accessor is torch's own)."""
    import datetime
    import torch.distributed as dist
    pg = group if group is not None else dist.group.WORLD
    try:
        ranks = list(dist.get_process_group_ranks(pg))
        store = dist.distributed_c10d._get_default_store()
    except Exception:
        store = None
    if store is None:
        if fatal:
            return True
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return bool(int(t.item()))
    me = dist.get_rank()
    tag = "-".join(str(r) for r in ranks)
    last, of_store = _agreements.get(tag, (0, None))
    seq = 1 if of_store is not store else last + 1           # (a process group created anew starts its count anew, everywhere)
    _agreements[tag] = (seq, store)
    key = lambda r: "pfpn/agree/%s/%d/%d" % (tag, seq, r)
    store.set(key(me), _FATAL if fatal else (b"1" if flag else b"0"))
    if fatal:
        return True
    store.wait([key(r) for r in ranks], datetime.timedelta(seconds=float(timeout_s)))
    votes = {r: bytes(store.get(key(r))) for r in ranks}
    lost = [r for r, v in votes.items() if v == _FATAL]
    if lost:
        raise ReplicaLost("replica(s) %s lost their stream capture to the runtime and cannot go on: the job must be restarted "
                          "(launch the ranks under a launcher that ends the job when one rank fails, e.g. torch.distributed.run)"
                          % lost)
    return any(v != b"0" for v in votes.values())


def refill_noise(noise, generator):
    """fresh draws into static noise buffers, same distributions as MixtureGaussianDistribution.make_noise /
    the SAC network's make_train_noise (u_cat: U[0,1) float64; gumbel*: -log(-log U(tiny,1)); others: N(0,1))."""
    if getattr(noise, "refill", None) is not None:      # SAC learner step: one launch draws all four tensors
        noise.refill(noise["_buf"])
        return
    if "_buf" in noise:                       # (a buffer of the older kind: one uniform fill + one transform launch)
        kernels.sac_noise_from_uniform_(noise["_buf"].uniform_(generator=generator))
        return
    for k, t in noise.items():
        if k == "u_cat":
            t.uniform_(generator=generator)
        elif k.startswith("gumbel"):
            kernels.gumbel_from_uniform_(t.uniform_(generator=generator))
        else:
            t.normal_(generator=generator)


class GraphedLearnerStep(object):
    """`net.compute_gradients(*inputs[, noise])` + the optimiser's device part as one hipGraph.
    `inputs` are static device tensors (a call with `batch` copies into them); `noise` (SAC: the
    reparameterisation draws) is a dict of static tensors refilled from the network's generator before every
    replay -- random draws stay outside the graph.

    `steps_per_replay = K > 1` captures K consecutive learner steps (each on its own static input set, `input_sets`;
    default: the same `inputs` K times) in ONE graph: consecutive replays of a one-step graph leave the GPU idle for
    ~10 us between them, K steps per replay pay that once.  The host part of every step (step counter, resample tick)
    runs after the replay, so a K-step replay is only legal while no resample falls inside it: `can_replay()`."""

    def __init__(self, net, optimizer, *inputs, warmup=3, noise=None, steps_per_replay=1, input_sets=None,
                 gather_from=None, capture_exchange=None, debug_nodes=False):
        """`gather_from` = (columns, rows [K, B] int64, losses [K, 3]), all static device tensors: step k of a replay
        first gathers rows[k] of the staged rollout `columns` into its input set (the minibatch slicing of a training
        round, inside the graph) and leaves its loss3 in losses[k].
        `capture_exchange` (default: env PFPN_GRAPH_EXCHANGE, ON over RCCL with the all-reduce exchange): data-parallel step
        with the all-reduce INSIDE the capture (one graph, K steps per replay allowed); False: two graphs around an eagerly
        enqueued collective.  The default falls back to the two-graph form by itself (see below); an explicit True raises."""
        self.net, self.opt = net, optimizer
        self.gather_from = gather_from
        self.debug_nodes = bool(debug_nodes)               # keep the graph's node list (kernel_nodes(): launches per replay)
        # Data parallel, two forms.  DEFAULT over RCCL (round 4; measured on a world-size-1 group: 134 us per 512-row step
        # against 150, bench.py `strong_shard`): the bucket all-reduce is captured INSIDE the step's graph -- RCCL's kernels
        # are captured like any other once the communicator exists (one tiny eager all-reduce below, which every replica
        # runs because every replica builds its stepper) -- so a data-parallel step is ONE graph and K steps share a replay
        # as on one replica.  FALLBACK, taken by every replica together when the capture fails on ANY of them (agreed
        # through an all-reduce(MAX)), over gloo (its collectives run on the host), for the sharded exchange (its
        # all-to-all does not belong in a long-lived graph, see `_with_collectives`) and with PFPN_GRAPH_EXCHANGE=0 /
        # `capture_exchange=False`: TWO graphs around the eagerly enqueued collective (everything up to the per-replica
        # clip | Adam + statistics); the host enqueues 3 things per step instead of ~25.
        exchanging = bool(getattr(optimizer, "exchange", optimizer.world > 1))
        explicit_arg = capture_exchange is not None
        if capture_exchange is None:
            capture_exchange = os.environ.get("PFPN_GRAPH_EXCHANGE", "1") != "0"
        want = bool(exchanging and capture_exchange)
        if want:
            import torch.distributed as dist
            if getattr(optimizer, "sharded", False):
                if explicit_arg:
                    raise ValueError("capture_exchange covers the all-reduce exchange only")
                want = False
            elif dist.get_backend(optimizer.pg) != "nccl":
                if explicit_arg:
                    raise ValueError("capture_exchange needs the RCCL backend (gloo collectives run on the host)")
                want = False
        self.K = int(steps_per_replay)
        # (K > 1 with noise: every step's draw must happen INSIDE the replay -- as a rider of the previous optimiser launch)
        assert self.K >= 1 and (self.K == 1 or noise is None or hasattr(net, "stage_next_noise")), \
            "several captured steps need a network that draws its noise inside the graph"
        self.input_sets = [tuple(inputs)] * self.K if input_sets is None else [tuple(t) for t in input_sets]
        assert len(self.input_sets) == self.K
        self.inputs = self.input_sets[0]
        self.noise = noise
        self.exchange_fallback = None
        net._ensure_buffers(self.inputs[0].shape[0], pin=True)    # the graph keeps pointing at this size's buffers
        if hasattr(net, "sync_target_net_hard"):
            net.sync_target_net_hard()                       # SAC: the init op must not happen inside the warm-up
        saved = [t.clone() for t in self._state_tensors()]
        if want:
            import torch.distributed as dist
            failed, why = 0, None
            # The replicas agree on the fall-back through an all-reduce(MAX) of `failed` -- which pairs up only if every
            # replica has issued the SAME collectives before it.  The warm-up runs real bucket all-reduces: it happens
            # OUTSIDE the try (a replica that fails there aborts with its exception; it must not walk into the flag's
            # all-reduce while the others are still summing buckets).  Only the capture region itself -- in which
            # collectives are recorded, not executed -- may fail on one replica alone and fall back.
            dist.all_reduce(torch.zeros(1, device=net.device), group=optimizer.pg)
            torch.cuda.synchronize()
            self._warm_up(True, exchanging, warmup, saved)
            try:
                self._record(True)
            except Exception as e:                           # (a capture that RCCL / the runtime refuses)
                failed, why = 1, repr(e)
                if capture_invalidated(e):
                    # the others are about to wait for this replica's vote: tell them first (host side, the store), then
                    # fail -- they raise ReplicaLost within a store round trip instead of sitting in a collective's timeout
                    any_replica(True, optimizer.pg, net.device, fatal=True)
                    raise CaptureInvalidated(e) from e
            if any_replica(failed, optimizer.pg, net.device):
                if explicit_arg:
                    raise RuntimeError("capturing the exchange failed on a replica: %s" % (why or "on another replica"))
                sys.stderr.write("pfpn_amd.graphs: the all-reduce could not be captured (%s): two graphs around an eager "
                                 "collective instead\n" % (why or "failed on another replica"))
                self.exchange_fallback = why or "failed on another replica"
                self.close()
                for t, sv in zip(self._state_tensors(), saved):
                    t.copy_(sv)
                want = False
        if not want:
            assert not (exchanging and self.K > 1), "multi-step replays of a data-parallel step need the captured exchange"
            self._capture(False, exchanging, warmup, saved)

    def _capture(self, in_graph, exchanging, warmup, saved):
        self._warm_up(in_graph, exchanging, warmup, saved)
        self._record(in_graph)

    def _warm_up(self, in_graph, exchanging, warmup, saved):
        net, optimizer = self.net, self.opt
        self.in_graph_exchange = bool(in_graph)
        self.split = exchanging and not in_graph
        # warm up on a side stream (lazy hipBLASLt/workspace init), then restore every piece of state
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                net.compute_gradients(*self._args(0))
                if self.in_graph_exchange:
                    # the warm-up DOES run the collective here, on the real bucket: RCCL sets up what a message of this
                    # size needs (channels, buffers) on its first use, which must not fall inside the capture.  Every
                    # replica builds its stepper at the same point, so the calls pair up; the state is restored below.
                    optimizer.pre_reduce(net)
                    optimizer.reduce(net)
                    optimizer.post_reduce(net)
                    optimizer.finish_reduce(net)
                elif self.split:       # the warm-up must not talk to the other replicas: no collective
                    optimizer.pre_reduce(net)
                    optimizer.post_reduce(net)
                else:
                    optimizer.apply_device(net)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.capture_stream = s
        for t, sv in zip(self._state_tensors(), saved):
            t.copy_(sv)

    def _record(self, in_graph):
        net, optimizer, gather_from, s = self.net, self.opt, self.gather_from, self.capture_stream
        want_fail = inject_capture_failure
        rank_here = getattr(optimizer, "rank_for_tests", None)
        # (types compared explicitly: False == 0 and True == 1 in Python -- "off" must not read as "replica 0")
        if in_graph and (want_fail is True or
                         (type(want_fail) is int and type(rank_here) is int and want_fail == rank_here)):
            raise RuntimeError("injected capture failure")
        if in_graph and want_fail == "invalidated":
            raise RuntimeError("injected: operation not permitted when stream is capturing")
        self.graph = torch.cuda.CUDAGraph()
        if self.debug_nodes:
            self.graph.enable_debug_mode()
        self.graph_post = None
        if self.in_graph_exchange:
            _with_collectives.add(self)
        # (thread-local capture mode whenever a process group is alive: its watchdog thread queries events at any time, and in
        # the default -- global -- mode a query from ANOTHER thread invalidates this thread's capture: the "operation failed
        # due to a previous error during capture" that one full-suite run in seven died of in round 4)
        mode = _capture_mode(self.in_graph_exchange)
        with torch.cuda.graph(self.graph, stream=s, **mode):      # same stream as the warm-up: per-stream workspaces exist
            if gather_from is not None:
                kernels.gather_columns(list(gather_from[0]), list(self.input_sets[0]), gather_from[1][0])
            for k in range(self.K):
                self.loss3 = net.compute_gradients(*self._args(k))
                if k + 1 < self.K:
                    # the NEXT step's batch exists before this step's optimiser launch, which can then carry the next
                    # step's prologue as extra blocks (one launch less per step: SyncReplicasApplier.apply_device)
                    if gather_from is not None:
                        kernels.gather_columns(list(gather_from[0]), list(self.input_sets[k + 1]), gather_from[1][k + 1])
                    if hasattr(net, "stage_next_prologue"):
                        net.stage_next_prologue(*self.input_sets[k + 1])
                if self.noise is not None and hasattr(net, "stage_next_noise"):
                    # the NEXT step's noise (the next replay's first step after the last one) is drawn by extra blocks of
                    # this step's optimiser launch: no draw launch between replays
                    net.stage_next_noise(self.noise)
                if self.split:
                    optimizer.pre_reduce(net)
                elif self.in_graph_exchange:
                    optimizer.pre_reduce(net)
                    optimizer.reduce(net)
                    optimizer.post_reduce(net)
                    optimizer.finish_reduce(net)
                else:
                    optimizer.apply_device(net)
                net._next_prologue = None              # (a rider the optimiser path did not take must not outlive its step)
                if gather_from is not None:
                    gather_from[2][k].copy_(self.loss3)
        if self.split:
            self.graph_post = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_post, stream=s, **mode):
                optimizer.post_reduce(net)
        # capture does not execute; state is untouched
        # True: EVERY captured step's optimiser launch took the noise rider, so only the FIRST replay needs a draw from outside.
        # A K-step replay in which some step did not take it (the exchange path, no norm hook, > 4 Adam groups, a noise
        # dict without a contiguous `_buf`: SyncReplicasApplier.apply_device drops the rider there) would run those steps
        # on the PREVIOUS step's Gumbel / normal draws, silently: refused.
        takers = int(getattr(net, "_noise_rode", 0)) if self.noise is not None else 0
        net._noise_rode = 0
        self._noise_fresh = False
        self.noise_in_graph = bool(self.noise is not None and takers == self.K)
        if self.noise is not None and self.K > 1 and takers != self.K:
            self.close()
            raise RuntimeError("a %d-step capture drew its noise inside the graph in %d of its steps: this optimiser path does "
                               "not carry the draw (steps_per_replay=1 refills the noise between replays)" % (self.K, takers))

    def ensure_noise(self):
        """the noise the NEXT replay will use is in `self.noise` after this call (drawn here unless the previous replay's
        last optimiser launch drew it)"""
        if self.noise is not None and not (self.noise_in_graph and self._noise_fresh):
            refill_noise(self.noise, self.net.rng)
            self._noise_fresh = True

    def kernel_nodes(self):
        """kernel launches per replay of the main graph, counted in the runtime's own dump of the captured graph
        (`debug_nodes=True` at construction); None where the runtime cannot print it"""
        if not self.debug_nodes:
            raise ValueError("construct the stepper with debug_nodes=True")
        import re
        import tempfile
        try:
            with tempfile.TemporaryDirectory() as d:
                path = os.path.join(d, "graph.dot")
                self.graph.debug_dump(path)
                txt = open(path).read()
            n = len(re.findall(r"label=\"[^\"]*(?:KERNEL|kernel)", txt))
            return n or None
        except Exception:
            return None

    def close(self):
        """destroy the captured graphs (before the process group they hold collectives of goes away)"""
        self.graph = self.graph_post = None
        _with_collectives.discard(self)

    def _args(self, k):
        return self.input_sets[k] if self.noise is None else self.input_sets[k] + (self.noise,)

    def can_replay(self):
        """True while the next K steps contain no resample tick before their last one."""
        f = getattr(self.net, "steps_until_tick", None)
        return self.K == 1 or f is None or f() >= self.K

    def _state_tensors(self):
        net, opt, st = self.net, self.opt, self.net.store
        ts = [st.flat, st.m, st.v, st.bucket, net.step_dev, opt.beta_pow]
        if getattr(net, "noise_epoch", None) is not None:
            ts.append(net.noise_epoch)
        if getattr(net, "target_store", None) is not None:
            ts.append(net.target_store.flat)
        if net.state_normalizer is not None:
            ts += [net.state_normalizer.mean, net.state_normalizer.std]
        if getattr(net, "max_active", None) is not None:
            ts += [net.max_active, net.sum_active]
        return ts

    def load_rows(self, columns, rows):
        """K = 1: gather `rows` of the staged rollout `columns` (device tensors, one per input) straight into the
        step's static inputs -- the device side of a training round's minibatch slicing."""
        kernels.gather_columns(list(columns), list(self.inputs), rows)

    def __call__(self, batch=None):
        """one replay = K learner steps.  `batch`: K = 1: the step's tensors; K > 1: a list of K such tuples."""
        if batch is not None:
            sets = [batch] if self.K == 1 else batch
            assert len(sets) == self.K
            for dsts, srcs in zip(self.input_sets, sets):
                for dst, src in zip(dsts, srcs):
                    dst.copy_(src, non_blocking=True)
        if self.graph is None:
            raise RuntimeError("this stepper was closed (release_captured_collectives)")
        if not self.can_replay():
            raise ValueError("a resample tick falls inside this %d-step replay: run one-step replays up to it" % self.K)
        self.ensure_noise()
        self.graph.replay()
        self._noise_fresh = self.noise_in_graph            # (the replay's last optimiser launch left the next draw)
        self.replays = getattr(self, "replays", 0) + 1
        if self.split:
            self.opt.reduce(self.net)
            self.graph_post.replay()
            self.opt.finish_reduce(self.net)         # (sharded exchange: the all-gather of the owners' slices)
        for _ in range(self.K):
            self.opt.apply_host(self.net)
        return self.loss3


class GraphedRollout(object):
    """`net.run_batch` for a fixed number E of environments as ONE hipGraph replay per environment step.

    The rollout inference is ~15 short launches (normalise, 3 GEMMs + epilogues, sample, row sum, statistics,
    critic head): ~200 us of host dispatch per call whatever E is.  Captured once, a step costs one H2D copy of
    the states, two RNG fills (the draws stay OUTSIDE the graph, from the network's generator) and one replay.
    Returns the same (action, log_prob, value) device tensors on every call (overwritten by the next one)."""

    def __init__(self, net, E, warmup=2, noise=None, capture=True):
        """`noise`: static noise buffers to read (row views of another instance's: the overlapped rollout draws a time
        step's noise once for all environments and runs the policy on half of them at a time); `capture=False`: only
        own the noise buffers of an E-row call (`refill()`), no graph."""
        self.net, self.E = net, int(E)
        dev = net.device
        self.states = torch.zeros(self.E, net.S, dtype=torch.float32, device=dev)
        net._ensure_buffers(self.E, pin=capture)
        policy, _, _ = net._forward(self.states)
        self.noise = policy.make_noise(net.rng) if noise is None else noise      # static buffers of the right shapes / dtypes
        if not capture:
            return
        stats = [t for t in (getattr(net, "max_active", None), getattr(net, "sum_active", None)) if t is not None]
        saved = [t.clone() for t in stats]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                net.run_batch(self.states, noise=self.noise)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for t, sv in zip(stats, saved):
            t.copy_(sv)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s, **_capture_mode()):
            self.out = net.run_batch(self.states, noise=self.noise)
        self.policy = net.policy

    def refill(self):
        refill_noise(self.noise, self.net.rng)               # fresh draws, same distributions as make_noise

    def __call__(self, states, refill=True):
        net = self.net
        self.states.copy_(torch.as_tensor(states, dtype=torch.float32).view(self.E, net.S), non_blocking=True)
        if refill:
            self.refill()
        self.graph.replay()
        net.policy = self.policy
        return self.out
