"""Host-side environment shims.

The reference steps PyBullet DeepMimic humanoids on the host CPU (envs/, out of scope: SURVEY
section 2 #14); only their DIMENSIONS matter to the hot path (S=197 observations, A=36 actions,
reference envs/deepmimic.py:420-464, :555-592).  gym/pybullet are not installed in the build image,
so `make_env` returns the real environment when `gym` + the reference's `envs` package import, and
otherwise a synthetic stand-in with the same spaces (random linear dynamics, dense reward) that lets
the rollout loop, buffers and learner run end to end.
"""
import numpy as np


class Box(object):
    def __init__(self, low, high):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(np.float32)


class SyntheticDeepMimicEnv(object):
    """Same observation/action spaces as DeepMimic{Walk,Kick,Punch}-v0; 600-step episodes with the
    gym TimeLimit convention (info['TimeLimit.truncated'])."""

    def __init__(self, state_dim=197, action_dim=36, episode_length=600, seed=0, step_cost=0.0):
        # step_cost: seconds of host CPU a step burns (busy, like physics -- a PyBullet DeepMimic step is ~1 ms), so the
        # overlap of environment stepping with the policy call can be measured without pybullet
        self.step_cost = float(step_cost)
        self.observation_space = Box(-np.inf * np.ones(state_dim), np.inf * np.ones(state_dim))
        self.action_space = Box(-np.ones(action_dim), np.ones(action_dim))
        self.episode_length = episode_length
        self.rng = np.random.RandomState(seed)
        self.W = (self.rng.randn(action_dim, state_dim) * 0.1).astype(np.float32)
        self.t = 0

    def seed(self, s):
        self.rng = np.random.RandomState(s)

    def reset(self):
        self.t = 0
        self.s = self.rng.randn(self.observation_space.shape[0]).astype(np.float32)
        return self.s.copy()

    def step(self, a):
        self.t += 1
        if self.step_cost > 0:
            import time
            t_end = time.perf_counter() + self.step_cost
            while time.perf_counter() < t_end:
                pass
        self.s = (0.95 * self.s + np.asarray(a, np.float32) @ self.W
                  + 0.05 * self.rng.randn(self.s.shape[0]).astype(np.float32))
        reward = float(np.exp(-0.05 * np.square(self.s[:8]).sum()))
        fell = bool(abs(self.s[1]) > 6.0)
        timeout = self.t >= self.episode_length
        info = {"TimeLimit.truncated": True} if (timeout and not fell) else {}
        return self.s.copy(), reward, fell or timeout, info

    def render(self):
        pass


class SyntheticReachEnv(object):
    """A control problem with a KNOWN optimum and an immediate reward, for the learning tests (tests/test_gpu_learning.py):
    DeepMimic's spaces (S = 197, A = 36), reward_t = exp(-(4 / A) |a_t - g(s_t)|^2) with the target g(s) = 0.5 tanh(s[:A]),
    action-independent AR(1) state (s' = 0.9 s + sqrt(1 - 0.81) noise: unit stationary variance), 200-step episodes that end
    by the time limit only.  Returns: optimum 200 (a = g(s)); the best state-blind policy (a = 0) ~ 134; uniform random
    actions ~ 36.  Q(s, a) = reward(s, a) + a function of s alone, so an off-policy critic sees the action's effect in ONE
    step -- unlike SyntheticDeepMimicEnv, whose reward answers an action only through the next states."""

    def __init__(self, state_dim=197, action_dim=36, episode_length=200, seed=0):
        self.observation_space = Box(-np.inf * np.ones(state_dim), np.inf * np.ones(state_dim))
        self.action_space = Box(-np.ones(action_dim), np.ones(action_dim))
        self.episode_length = episode_length
        self.rng = np.random.RandomState(seed)
        self.t = 0

    def seed(self, s):
        self.rng = np.random.RandomState(s)

    def target(self, s):
        return 0.5 * np.tanh(s[:self.action_space.shape[0]])

    def reset(self):
        self.t = 0
        self.s = self.rng.randn(self.observation_space.shape[0]).astype(np.float32)
        return self.s.copy()

    def step(self, a):
        self.t += 1
        d = np.asarray(a, np.float32) - self.target(self.s)
        reward = float(np.exp(-4.0 / d.shape[0] * np.square(d).sum()))
        self.s = (0.9 * self.s + np.sqrt(1 - 0.81) * self.rng.randn(self.s.shape[0])).astype(np.float32)
        timeout = self.t >= self.episode_length
        return self.s.copy(), reward, timeout, ({"TimeLimit.truncated": True} if timeout else {})

    def render(self):
        pass


def make_env(name, render=False, seed=0):
    if str(name).startswith("SyntheticReach"):
        return SyntheticReachEnv(seed=seed)
    try:
        import gym  # noqa: F401
        import envs  # noqa: F401  (the reference's PyBullet package, if on sys.path)
        e = gym.make(name)
        if render:
            e.render()
        return e
    except Exception:
        return SyntheticDeepMimicEnv(seed=seed)
