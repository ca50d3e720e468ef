"""Host environment pool: E environments stepped by P worker processes, one batched policy call per step.

SURVEY 8(f)-1, the caller side of the hot path.  The reference runs ONE environment per worker process and
one TF session call per environment step (models/distributed_model.py:154-193, :264-290); on a GPU that is a
launch-latency-bound batch of 1.  Here the policy sees all E states at once (`net.run_batch`: one launch
sequence for sample + log-prob + value + activation statistics) while the physics -- the host-bound part --
is spread over processes.  States, actions, rewards and flags travel through shared memory; a pipe per
process carries only the one-byte step / reset commands.  Terminated environments are reset inside the
worker, so `step` always returns a full batch (the terminal observation is kept for the value bootstrap).
"""
import multiprocessing as mp

import numpy as np


def _shared(shape, dtype=np.float32):
    n = int(np.prod(shape))
    raw = mp.RawArray("b", n * np.dtype(dtype).itemsize)
    return raw, np.frombuffer(raw, dtype=dtype).reshape(shape)


def _view(raw, shape, dtype=np.float32):
    return np.frombuffer(raw, dtype=dtype).reshape(shape)


def _worker(conn, env_fns, lo, raws, shapes):
    envs = [fn() for fn in env_fns]
    state = _view(raws["state"], shapes["state"])
    last = _view(raws["last"], shapes["state"])
    action = _view(raws["action"], shapes["action"])
    reward = _view(raws["reward"], shapes["flag"])
    done = _view(raws["done"], shapes["flag"], np.uint8)
    trunc = _view(raws["trunc"], shapes["flag"], np.uint8)
    bonus = _view(raws["bonus"], shapes["flag"])
    try:
        while True:
            cmd = conn.recv_bytes()
            if cmd == b"r":
                for i, e in enumerate(envs):
                    state[lo + i] = e.reset()
            elif cmd == b"s":
                for i, e in enumerate(envs):
                    k = lo + i
                    a = action[k]
                    if hasattr(e.action_space, "low"):                       # the env sees the clipped action
                        a = np.clip(a, e.action_space.low, e.action_space.high)
                    s, r, d, info = e.step(a)
                    reward[k], done[k] = r, d
                    trunc[k] = 1 if info.get("TimeLimit.truncated") else 0
                    bonus[k] = info.get("episode_reward", 0.0)
                    last[k] = s
                    state[k] = e.reset() if d else s
            else:
                break
            conn.send_bytes(b"k")
    finally:
        conn.close()


class EnvPool(object):
    """`env_fns`: E zero-argument constructors (picklable or fork-inherited); `processes`: worker count.

    reset() -> states [E,S]
    step(actions [E,A]) -> (next_states [E,S], rewards [E], done [E] bool, truncated [E] bool, last_states [E,S])
    next_states already holds the first observation of the new episode where done; last_states the
    observation the finished step ended in (for V(s_T) at a time limit)."""

    def __init__(self, env_fns, processes=None, start_method="fork"):
        self.E = len(env_fns)
        probe = env_fns[0]()
        self.observation_space, self.action_space = probe.observation_space, probe.action_space
        S, A = int(np.prod(probe.observation_space.shape)), int(np.prod(probe.action_space.shape))
        del probe
        self.shapes = {"state": (self.E, S), "action": (self.E, A), "flag": (self.E,)}
        self.raws, self.np = {}, {}
        for k, shp, dt in (("state", "state", np.float32), ("last", "state", np.float32),
                           ("action", "action", np.float32), ("reward", "flag", np.float32),
                           ("done", "flag", np.uint8), ("trunc", "flag", np.uint8), ("bonus", "flag", np.float32)):
            self.raws[k], self.np[k] = _shared(self.shapes[shp], dt)
        P = min(self.E, processes or min(self.E, max(1, (mp.cpu_count() or 2) - 1)))
        ctx = mp.get_context(start_method)
        bounds = np.linspace(0, self.E, P + 1).astype(int)
        self.bounds = [int(b) for b in bounds]
        self.conns, self.procs = [], []
        # Forked workers inherit the parent's Python heap -- in a learner process that holds HIP objects (streams, graphs,
        # tensors).  A garbage collection inside a child would finalise its COPIES of them (observed: a segmentation fault
        # in the child's first collection, right behind fork).  gc.freeze() moves everything alive now into the permanent
        # generation, which the children inherit: their collector never looks at the parent's objects.
        import gc
        frozen = start_method == "fork"
        if frozen:
            gc.collect()
            gc.freeze()
        try:
            for p in range(P):
                lo, hi = int(bounds[p]), int(bounds[p + 1])
                parent, child = ctx.Pipe()
                pr = ctx.Process(target=_worker, args=(child, env_fns[lo:hi], lo, self.raws, self.shapes), daemon=True)
                pr.start()
                child.close()
                self.conns.append(parent)
                self.procs.append(pr)
        finally:
            if frozen:
                gc.unfreeze()

    def _round(self, cmd):
        for c in self.conns:
            c.send_bytes(cmd)
        for c in self.conns:
            if c.recv_bytes() != b"k":
                raise RuntimeError("environment worker failed")

    # -- two halves that step independently (the overlapped rollout: one half's environments step on the host while the
    # GPU runs the other half's policy call) ----------------------------------------------------------------------------
    def halves(self):
        """[(env slice, worker connections)] x 2: the first / second half of the worker processes with their
        environments (needs >= 2 processes)."""
        P = len(self.conns)
        if P < 2:
            raise ValueError("the overlapped rollout needs at least two worker processes")
        h = P // 2
        mid = self.bounds[h]
        return [(slice(0, mid), self.conns[:h]), (slice(mid, self.E), self.conns[h:])]

    def step_send(self, half, actions):
        sl, conns = half
        self.np["action"][sl] = np.asarray(actions, np.float32).reshape(self.np["action"][sl].shape)
        for c in conns:
            c.send_bytes(b"s")

    def step_wait(self, half):
        sl, conns = half
        for c in conns:
            if c.recv_bytes() != b"k":
                raise RuntimeError("environment worker failed")
        n = self.np
        return (n["state"][sl].copy(), n["reward"][sl].copy(), n["done"][sl].astype(bool), n["trunc"][sl].astype(bool),
                n["last"][sl].copy())

    def reset(self):
        self._round(b"r")
        return self.np["state"].copy()

    def step(self, actions):
        self.np["action"][...] = np.asarray(actions, np.float32).reshape(self.shapes["action"])
        self._round(b"s")
        n = self.np
        return (n["state"].copy(), n["reward"].copy(), n["done"].astype(bool), n["trunc"].astype(bool),
                n["last"].copy())

    def close(self):
        for c in self.conns:
            try:
                c.send_bytes(b"q")
                c.close()
            except (OSError, BrokenPipeError):
                pass
        for p in self.procs:
            p.join(timeout=2)
            if p.is_alive():
                p.terminate()
        self.conns, self.procs = [], []

    def __del__(self):
        self.close()


class BatchedPPORollout(object):
    """Collects `unroll_length` steps from each of the pool's E environments with one batched policy call
    per step, and turns them into the PPO learner's buffers [state, action, value, log_prob, advantage] with
    the reference worker's conventions (models/workers/ppo.py:43-73, distributed_model.py:264-290): the
    UNCLIPPED action is stored, a true terminal bootstraps with 0 and a time limit / unroll boundary with
    V(last state), GAE runs per environment segment (device scan over ragged sequences)."""

    def __init__(self, net, pool, unroll_length, graph=None, overlap=False):
        """overlap: the pool's two halves alternate -- while half A's environments step on the host, the GPU runs half
        B's policy call (and vice versa), so neither side waits for the other's whole turn.  Same draws as the
        sequential rollout: the noise of a time step is drawn ONCE for all E environments (one generator call per
        tensor, as `run_batch` over E rows makes) and each half's call reads its rows of it."""
        self.net, self.pool, self.T = net, pool, int(unroll_length)
        self.state = pool.reset()
        self.total_samples = 0
        self.overlap = bool(overlap)
        # on a GPU the per-step policy call is one hipGraph replay (pfpn_amd/graphs.py: GraphedRollout)
        if graph is None:
            graph = getattr(net, "device", None) is not None and getattr(net.device, "type", "cpu") == "cuda"
        self._infer = net.run_batch
        if self.overlap:
            self.halves = pool.halves()
            sizes = [h[0].stop - h[0].start for h in self.halves]
            if graph:
                from .graphs import GraphedRollout
                full = GraphedRollout(net, pool.E, capture=False)          # owns the step's noise buffers [E, ...]
                self._noise_owner = full
                self._infer_half = [GraphedRollout(net, n, noise={k: v[h[0]] for k, v in full.noise.items()})
                                    for n, h in zip(sizes, self.halves)]
            else:
                self._noise_owner = None
                self._infer_half = [net.run_batch, net.run_batch]
        elif graph:
            from .graphs import GraphedRollout
            self._infer = GraphedRollout(net, pool.E)

    def collect(self):
        import torch
        net, E, T = self.net, self.pool.E, self.T
        S, A = self.state.shape[1], self.pool.shapes["action"][1]
        dev = net.device
        states = np.empty((T, E, S), np.float32)
        actions = torch.empty(T, E, A, dtype=torch.float32, device=dev)
        values = torch.empty(T, E, dtype=torch.float32, device=dev)
        logps = torch.empty(T, E, dtype=torch.float32, device=dev)
        rewards = np.empty((T, E), np.float32)
        cut = np.zeros((T, E), bool)                      # a segment ends after this step
        boot_state = np.zeros((T, E, S), np.float32)      # ... and bootstraps with V(this state) unless terminal
        boot_zero = np.zeros((T, E), bool)
        def book(t, sl, nxt, r, done, trunc, last):
            rewards[t, sl] = r
            end = done | (t == T - 1)
            cut[t, sl] = end
            boot_zero[t, sl] = done & ~trunc               # a real terminal: bootstrap value 0
            bs = boot_state[t, sl]
            bs[end] = np.where(done[:, None], last, nxt)[end]
            boot_state[t, sl] = bs
            self.state[sl] = nxt

        if self.overlap:
            # half h's environments step while the other half's policy call runs: send(A) | infer(B) | send(B) | wait(A) ...
            pool, halves = self.pool, self.halves
            pending = [False, False]
            for t in range(T):
                if self._noise_owner is not None:
                    self._noise_owner.refill()             # ONE draw per time step for all E environments
                for h, (sl, _) in enumerate(halves):
                    if pending[h]:
                        book(t - 1, sl, *pool.step_wait(halves[h]))
                        pending[h] = False
                    st_h = self.state[sl]
                    a, lp, v = self._infer_half[h](st_h) if self._noise_owner is None else \
                        self._infer_half[h](st_h, refill=False)
                    states[t, sl] = st_h
                    actions[t, sl], logps[t, sl], values[t, sl] = a, lp, v
                    pool.step_send(halves[h], a.cpu().numpy())
                    pending[h] = True
            for h, (sl, _) in enumerate(halves):
                book(T - 1, sl, *pool.step_wait(halves[h]))
        else:
            full = slice(0, E)
            for t in range(T):
                a, lp, v = self._infer(self.state)
                states[t] = self.state
                actions[t], logps[t], values[t] = a, lp, v
                book(t, full, *self.pool.step(a.cpu().numpy()))
        self.total_samples += T * E
        # bootstrap values of all segment ends in one batched critic call
        idx = np.argwhere(cut)                             # (t, e) pairs, time-major
        vb = net.evaluate_batch(boot_state[cut]).cpu().numpy()
        vb[boot_zero[cut]] = 0.0
        # ragged segments, environment-major: reward [sum T_seg], value [sum (T_seg + 1)]
        vals = values.cpu().numpy()
        seg_r, seg_v, order, offsets = [], [], [], [0]
        boot = {(int(t), int(e)): vb[i] for i, (t, e) in enumerate(idx)}
        for e in range(E):
            start = 0
            for t in range(T):
                if cut[t, e]:
                    seg_r.append(rewards[start:t + 1, e])
                    seg_v.append(np.append(vals[start:t + 1, e], boot[(t, e)]))
                    order.extend((tt, e) for tt in range(start, t + 1))
                    offsets.append(offsets[-1] + (t + 1 - start))
                    start = t + 1
        adv = net.gae_batch(torch.from_numpy(np.concatenate(seg_r)).to(dev),
                            torch.from_numpy(np.concatenate(seg_v).astype(np.float32)).to(dev),
                            torch.tensor(offsets, dtype=torch.int64, device=dev))
        ot = torch.tensor([o[0] for o in order], device=dev)
        oe = torch.tensor([o[1] for o in order], device=dev)
        return {"state": torch.from_numpy(states).to(dev)[ot, oe], "action": actions[ot, oe],
                "value": values[ot, oe], "log_prob": logps[ot, oe], "advantage": adv}
