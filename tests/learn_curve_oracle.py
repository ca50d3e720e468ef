"""Test infrastructure (lives under tests/: only tests, smoke() and bench.py's cpu_baseline leg may touch oracle/).
The same DPPO-PFPN training loop as tools/learn_curve.py, on the CPU with the NumPy oracle: one
environment, 512-step unrolls, GAE, 16 shuffled 32-row minibatches per round, resample tick -- the reference worker's schedule
(models/workers/ppo.py:43-73, models/distributed_model.py:320-352) with the oracle's restated kernels.  Not a parity check (the two
sides draw different random numbers): a STATISTICAL twin -- does the learning curve of the HIP product have the shape the algorithm
itself produces on this environment?

  python tests/learn_curve_oracle.py --seed 34114 --steps 4000 [--lr 1e-4] [--resample-interval 150]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from oracle import pfpn_oracle as orc                      # noqa: E402
from pfpn_amd.envs import SyntheticDeepMimicEnv            # noqa: E402  (host-only module: no GPU, no library)

F = np.float32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=34114)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--resample-interval", type=int, default=150)
    ap.add_argument("--unroll", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(R, "gpurun_out", "learn_oracle.jsonl"))
    args = ap.parse_args()
    S, A, N = 197, 36, 35
    rng = np.random.default_rng(args.seed)
    st = orc.DPPOState(rng, S, A, N, resample_interval=args.resample_interval)
    env = SyntheticDeepMimicEnv(seed=args.seed)
    env.seed(args.seed + 1)
    shuffle_rng = np.random.RandomState(args.seed + 1)
    gamma, lam = 0.95, 0.95
    returns, ticks = [], []

    def forward(s):
        x = orc.state_normalize(s[None], st.norm_mean, st.norm_std, 5.0)
        logits, _ = orc.actor_forward(st.P, x, A, N)
        v, _ = orc.critic_forward(st.P, x)
        st.max_active, st.sum_active = orc.active_stats(logits[0], st.max_active, st.sum_active)     # every fetch
        return logits, float(v[0])

    exp = {k: [] for k in ("state", "action", "reward", "value", "log_prob", "advantage")}
    s, ep_ret, ep_len, stamp, t0 = env.reset(), 0.0, 0, 0, time.time()
    while st.global_step < args.steps:
        logits, v = forward(s)
        _, a = orc.sample_plain(logits, st.P["actor/samples"], st.P["actor/samples_std"], rng.random((1, A)),
                                rng.normal(0, 1, (1, A)).astype(F))
        lp, _ = orc.mixture_log_prob(logits, a, st.P["actor/samples"], st.P["actor/samples_std"])
        s2, r, over, info = env.step(np.clip(a[0], -1, 1))
        for k, val in (("state", s), ("action", a[0]), ("reward", r), ("value", v), ("log_prob", float(lp[0]))):
            exp[k].append(val)
        ep_ret += r
        ep_len += 1
        s = s2
        due = len(exp["state"]) >= args.unroll
        if over or due:
            true_end = over and not info.get("TimeLimit.truncated", False)
            boot = 0.0 if true_end else forward(s)[1]
            vals = np.append(np.asarray(exp["value"][stamp:], F), F(boot))
            exp["advantage"].extend(orc.gae(exp["reward"][stamp:], vals, gamma, gamma * lam))
            stamp = len(exp["value"])
        if over:
            returns.append(ep_ret)
            s, ep_ret, ep_len = env.reset(), 0.0, 0
        if due:
            n = len(exp["state"])
            cols = {k: np.asarray(exp[k], F) for k in exp}
            ids = np.arange(n)
            shuffle_rng.shuffle(ids)
            for lo in range(0, n, args.batch):
                rows = ids[lo:lo + args.batch]
                b = {k: cols[k][rows] for k in ("state", "action", "value", "log_prob", "advantage")}
                tick = st.train_flag + 1 >= st.resample_interval
                if tick:
                    dead = int((st.max_active < 0.05 / N).sum())
                    ticks.append({"step": st.global_step + 1, "dead": dead})
                orc.dppo_learner_step(st, [b], lr=args.lr, u_resample=rng.random((A, N)),
                                      u_noise=(rng.random((A, N)) * 2 - 1).astype(F))
            exp = {k: [] for k in exp}
            stamp = 0
            if len(returns) and (st.global_step // 16) % 20 == 0:
                print("step %d episodes %d last10 %.1f (%.0f s)" % (st.global_step, len(returns), np.mean(returns[-10:]),
                                                                  time.time() - t0), flush=True)
    g = 20
    line = {"run": "oracle-dppo:%d:%d" % (args.seed, args.steps), "lr": args.lr, "resample_interval": args.resample_interval,
            "episodes": len(returns), "seconds": round(time.time() - t0, 1),
            "curve": [round(float(np.mean(returns[i:i + g])), 2) for i in range(0, len(returns) - g + 1, g)],
            "first_k": float(np.mean(returns[:10])), "last_k": float(np.mean(returns[-10:])),
            "dead_per_tick": [t["dead"] for t in ticks]}
    print(json.dumps(line), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "a") as f:
        f.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()
