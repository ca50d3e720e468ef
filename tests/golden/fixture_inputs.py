"""Deterministic inputs of the graph-evaluated golden fixtures (shared by the generator, which runs in the build
container, and by the tests, which re-create the large weight matrices instead of storing them).

`hash_uniform` is an explicit counter-based generator (splitmix64 finaliser on uint64) so the values do not depend
on a NumPy version's Generator internals."""
import numpy as np

F = np.float32
S, A, N, H1, H2 = 197, 36, 35, 1024, 512

TRAINABLES = ["actor/fc1/weight", "actor/fc1/bias", "actor/fc2/weight", "actor/fc2/bias",
              "actor/samples", "actor/samples_std", "actor/fc_policy/weight", "actor/fc_policy/bias",
              "critic/fc1/weight", "critic/fc1/bias", "critic/fc2/weight", "critic/fc2/bias",
              "critic/fc3/weight", "critic/fc3/bias"]                     # order of the saved graph's ApplyAdam list
SHAPES = {"actor/fc1/weight": (S, H1), "actor/fc1/bias": (H1,), "actor/fc2/weight": (H1, H2), "actor/fc2/bias": (H2,),
          "actor/samples": (A, N), "actor/samples_std": (A, N), "actor/fc_policy/weight": (H2, A * N),
          "actor/fc_policy/bias": (A * N,), "critic/fc1/weight": (S, H1), "critic/fc1/bias": (H1,),
          "critic/fc2/weight": (H1, H2), "critic/fc2/bias": (H2,), "critic/fc3/weight": (H2, 1), "critic/fc3/bias": (1,)}
BIG = [k for k, s in SHAPES.items() if int(np.prod(s)) > 4096]
ROW_STRIDE = 64          # rows of a big matrix kept in the fixture: [::ROW_STRIDE]
P_ROW_STRIDE = 16        # compact fixtures: rows of the [B, A] mixture probabilities kept


def hash_uniform(shape, seed):
    """float64 uniforms in [0,1), value k of stream `seed` = splitmix64(seed * 2^32 + k) >> 11."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape)


def hash_normal(shape, seed):
    """Box-Muller on two hash streams (float64)."""
    u1 = hash_uniform(shape, 2 * seed + 1000003)
    u2 = hash_uniform(shape, 2 * seed + 1000004)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def make_variables(seed):
    """A trained-looking parameter set: weights large enough that ReLU6 clamps on both sides and the mixture
    weights are far from uniform; non-zero biases; particles off the grid; Adam moments and powers mid-training."""
    P, M, V = {}, {}, {}
    gains = {"actor/fc1/weight": 1.6, "actor/fc2/weight": 1.8, "actor/fc_policy/weight": 2.5,
             "critic/fc1/weight": 1.6, "critic/fc2/weight": 1.8, "critic/fc3/weight": 1.0}
    for i, k in enumerate(TRAINABLES):
        sh = SHAPES[k]
        if k.endswith("weight"):
            P[k] = (hash_normal(sh, seed * 100 + i) * gains[k] / np.sqrt(sh[0])).astype(F)
        elif k.endswith("bias"):
            P[k] = (hash_normal(sh, seed * 100 + i) * 0.3 + (0.5 if "fc_policy" not in k else 0.0)).astype(F)
        elif k.endswith("samples"):
            grid = (-1 + 2 * np.arange(N) / (N - 1))[None, :]
            P[k] = (grid + hash_normal(sh, seed * 100 + i) * 0.05).astype(F)
        else:   # samples_std (log std)
            P[k] = (np.log(2.0 / (N - 1)) + hash_normal(sh, seed * 100 + i) * 0.3).astype(F)
        M[k] = (hash_normal(sh, seed * 100 + 40 + i) * 1e-3).astype(F)
        V[k] = ((hash_uniform(sh, seed * 100 + 60 + i) * 1e-5) + 1e-12).astype(F)
    return P, M, V


def initial_state(seed, step):
    """Everything a learner replica holds before the step of fixture `seed`: trainables, Adam slots and powers after
    `step` applies, state-normaliser statistics."""
    P, M, V = make_variables(seed)
    mean = (hash_normal((S,), seed * 100 + 90) * 0.2 + 0.3).astype(F)
    std = (1.5 + hash_uniform((S,), seed * 100 + 91)).astype(F)
    b1p, b2p = beta_powers(step)
    return {"P": P, "M": M, "V": V, "mean": mean, "std": std, "b1p": b1p, "b2p": b2p, "step": step}


def make_batch(B, seed):
    return {"state": (hash_normal((B, S), seed * 10 + 1) * 2 + 0.3).astype(F),
            "action": (hash_uniform((B, A), seed * 10 + 2) * 2.2 - 1.1).astype(F),
            "value": (hash_normal((B,), seed * 10 + 3) * 0.5).astype(F),
            "advantage": hash_normal((B,), seed * 10 + 4).astype(F)}


def beta_powers(t):
    """The fp32 beta1_power / beta2_power variables after t applies (each apply multiplies in fp32)."""
    b1p, b2p = F(0.9), F(0.999)
    for _ in range(t):
        b1p, b2p = F(b1p * F(0.9)), F(b2p * F(0.999))
    return b1p, b2p


def summarize(x):
    """float64 checksums of a tensor: sum, sum of squares, sum of |x|."""
    x = np.asarray(x, np.float64)
    return np.asarray([x.sum(), (x * x).sum(), np.abs(x).sum()])


def subsample(name, x):
    return np.ascontiguousarray(x[::ROW_STRIDE]) if name in BIG else x


def make_batch_off_kinks(B, bseed, vseed, step, margin=1e-4):
    """make_batch(B, bseed) with every row redrawn whose trunk pre-activations (either trunk, either layer, evaluated in
    float64 with the fixture's parameters make_variables(vseed)) come within `margin` of a ReLU6 kink (0 or 6).  At 4096 rows
    ~12.6 M pre-activations are in play and fp32 evaluations that differ only in summation order (the reference's BLAS,
    the MFMA GEMMs) WILL put one of them on different sides of a kink: one unit's gradient then moves by a whole row's
    contribution, and everything below it by ~1e-3 of its scale -- a property of the batch, not of either implementation.
    The big fixtures therefore use batches with a clear margin (the small ones carry the MASK_FLIPS allowance instead).
    Deterministic: generator and tests call this same function."""
    init = initial_state(vseed, step)
    P, mean, std = init["P"], init["mean"].astype(np.float64), init["std"].astype(np.float64)
    b = make_batch(B, bseed)

    def bad_rows(state):
        x = np.clip((state.astype(np.float64) - mean) / std, -5.0, 5.0)
        bad = np.zeros(len(state), bool)
        for t in ("actor", "critic"):
            h1 = x @ P[t + "/fc1/weight"].astype(np.float64) + P[t + "/fc1/bias"]
            bad |= ((np.abs(h1) < margin) | (np.abs(h1 - 6.0) < margin)).any(1)
            h2 = np.clip(h1, 0.0, 6.0) @ P[t + "/fc2/weight"].astype(np.float64) + P[t + "/fc2/bias"]
            bad |= ((np.abs(h2) < margin) | (np.abs(h2 - 6.0) < margin)).any(1)
        return bad

    for attempt in range(1, 64):
        bad = bad_rows(b["state"])
        if not bad.any():
            return b
        fresh = (hash_normal((B, S), bseed * 10 + 1 + 7919 * attempt) * 2 + 0.3).astype(F)
        b["state"][bad] = fresh[bad]
    raise RuntimeError("could not move every row off the ReLU6 kinks")


def batch_of(fx, w):
    """worker w's learner inputs of a fixture: stored ones, or -- compact fixtures -- recreated from the hash streams
    (the generator drew them as make_batch_off_kinks(B, seed * 10 + w, seed, step)); the old log-prob is always stored (a
    graph output)."""
    if "in%d:state" % w in fx:
        return {k: fx["in%d:%s" % (w, k)] for k in ("state", "action", "value", "advantage", "log_prob")}
    B, W, step, train_flag, seed = [int(v) for v in fx["meta"]]
    b = make_batch_off_kinks(B, seed * 10 + w, seed, step)
    b["log_prob"] = fx["in%d:log_prob" % w]
    return b
