"""CPU: the host environment pool (processes + shared memory) and the batched PPO rollout bookkeeping."""
import numpy as np
import torch

from pfpn_amd.env_pool import BatchedPPORollout, EnvPool
from pfpn_amd.envs import SyntheticDeepMimicEnv


def make_fns(E, length=23):
    return [(lambda i=i: SyntheticDeepMimicEnv(episode_length=length, seed=100 + i)) for i in range(E)]


def test_pool_matches_sequential_envs():
    E, steps = 5, 60
    pool = EnvPool(make_fns(E), processes=2)
    try:
        ref = [fn() for fn in make_fns(E)]
        s_ref = np.stack([e.reset() for e in ref])
        s = pool.reset()
        assert np.array_equal(s, s_ref)
        rng = np.random.default_rng(0)
        for _ in range(steps):
            a = rng.uniform(-1.5, 1.5, (E, 36)).astype(np.float32)
            s, r, d, tr, last = pool.step(a)
            for i, e in enumerate(ref):
                si, ri, di, info = e.step(np.clip(a[i], -1, 1))          # the env sees the clipped action
                assert np.array_equal(last[i], si) and r[i] == np.float32(ri) and d[i] == di
                assert tr[i] == bool(info.get("TimeLimit.truncated", False))
                if di:
                    si = e.reset()                                       # auto-reset: next state of the new episode
                assert np.array_equal(s[i], si)
        assert d.sum() >= 0
    finally:
        pool.close()


class StubNet(object):
    """value = sum of the first three state entries; gae through the oracle (checker)."""
    device = torch.device("cpu")

    def __init__(self, oracle):
        self.oracle = oracle
        self.rng = np.random.default_rng(1)
        self.calls = 0

    def run_batch(self, states):
        st = torch.as_tensor(np.asarray(states, np.float32))
        self.calls += 1
        a = torch.from_numpy(self.rng.uniform(-1.5, 1.5, (st.shape[0], 36)).astype(np.float32))
        return a, torch.full((st.shape[0],), -40.0), st[:, :3].sum(1)

    def evaluate_batch(self, states):
        return torch.as_tensor(np.asarray(states, np.float32)).view(-1, 197)[:, :3].sum(1)

    def gae_batch(self, reward, value, offsets):
        r, v, off = reward.numpy(), value.numpy(), offsets.numpy()
        out = [self.oracle.gae(r[off[e]:off[e + 1]], v[off[e] + e:off[e + 1] + e + 1], 0.95, 0.95 * 0.95)
               for e in range(len(off) - 1)]
        return torch.from_numpy(np.concatenate(out).astype(np.float32))


def test_batched_rollout_buffers(oracle):
    E, T = 4, 40                                  # episodes of 23 steps: every env crosses one time limit
    pool = EnvPool(make_fns(E), processes=2)
    try:
        net = StubNet(oracle)
        ro = BatchedPPORollout(net, pool, T)
        buf = ro.collect()
        assert net.calls == T                     # ONE policy call per step for all environments
        n = E * T
        assert buf["state"].shape == (n, 197) and buf["action"].shape == (n, 36) and buf["advantage"].shape == (n,)
        assert ro.total_samples == n
        assert buf["action"].abs().max() > 1.0    # the unclipped action is stored
        # value buffer is V(state) of the stored state, environment-major segments
        np.testing.assert_allclose(buf["value"].numpy(), buf["state"].numpy()[:, :3].sum(1), rtol=1e-6)
        # replay env 0 by hand: segments [0,23) (time limit -> bootstrap V(last)), [23,40) (unroll boundary)
        st0 = buf["state"].numpy()[:T]
        v0 = st0[:, :3].sum(1)
        e0 = make_fns(1)[0]()
        s = e0.reset()
        rewards, lasts = [], []
        for t in range(T):
            assert np.array_equal(st0[t], s)
            s2, r, d, info = e0.step(np.clip(buf["action"].numpy()[t], -1, 1))
            rewards.append(r); lasts.append(s2)
            s = e0.reset() if d else s2
        adv_a = oracle.gae(np.array(rewards[:23], np.float32), np.append(v0[:23], lasts[22][:3].sum()), 0.95, 0.95 ** 2)
        adv_b = oracle.gae(np.array(rewards[23:], np.float32), np.append(v0[23:], s[:3].sum()), 0.95, 0.95 ** 2)
        np.testing.assert_allclose(buf["advantage"].numpy()[:T], np.concatenate([adv_a, adv_b]), rtol=1e-5, atol=1e-6)
    finally:
        pool.close()


class FnNet(StubNet):
    """a deterministic policy (the action is a function of the state) so that two rollouts can be compared row by row."""
    def run_batch(self, states):
        st = torch.as_tensor(np.asarray(states, np.float32))
        self.calls += 1
        return 1.5 * torch.tanh(st[:, :36] * 2.0), torch.full((st.shape[0],), -40.0), st[:, :3].sum(1)


def test_pool_halves_step_independently():
    """step_send / step_wait on the two halves, in the overlapped order, give what the sequential environments give."""
    E, steps = 6, 50
    pool = EnvPool(make_fns(E), processes=4)
    try:
        ref = [fn() for fn in make_fns(E)]
        s = pool.reset()
        assert np.array_equal(s, np.stack([e.reset() for e in ref]))
        halves = pool.halves()
        assert halves[0][0].start == 0 and halves[0][0].stop == halves[1][0].start and halves[1][0].stop == E
        rng = np.random.default_rng(0)
        for _ in range(steps):
            a = rng.uniform(-1.5, 1.5, (E, 36)).astype(np.float32)
            pool.step_send(halves[0], a[halves[0][0]])
            pool.step_send(halves[1], a[halves[1][0]])          # both halves in flight
            outs = [pool.step_wait(halves[1]), pool.step_wait(halves[0])][::-1]
            for (sl, _), (sh, r, d, tr, last) in zip(halves, outs):
                for j, i in enumerate(range(sl.start, sl.stop)):
                    si, ri, di, info = ref[i].step(np.clip(a[i], -1, 1))
                    assert np.array_equal(last[j], si) and r[j] == np.float32(ri) and d[j] == di
                    assert np.array_equal(sh[j], ref[i].reset() if di else si)
    finally:
        pool.close()


def test_overlapped_rollout_matches_sequential_rollout(oracle):
    """the two-half rollout (one half's environments step while the other half's policy call runs) fills the same buffers
    as the sequential one: same states, actions, values, advantages, row for row."""
    E, T = 6, 40
    bufs, calls = [], []
    for overlap in (False, True):
        pool = EnvPool(make_fns(E), processes=4)
        try:
            net = FnNet(oracle)
            ro = BatchedPPORollout(net, pool, T, overlap=overlap)
            bufs.append(ro.collect())
            bufs.append(ro.collect())                           # a second unroll continues from the first one's states
            calls.append(net.calls)
        finally:
            pool.close()
    assert calls == [2 * T, 4 * T]                              # one call per step | one per step and half
    for a, b in ((bufs[0], bufs[2]), (bufs[1], bufs[3])):
        for k in ("state", "action", "value", "log_prob", "advantage"):
            assert torch.equal(a[k], b[k]), k
