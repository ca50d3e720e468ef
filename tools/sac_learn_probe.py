"""What a SAC-PFPN run does in its first learner steps, step by step (eager steps: PFPN_GRAPH_TRAIN=0):
alpha, mean log-prob of the policy's own sample, mean |a|, share of |a| > 0.95, mean Q(s, a_pi), the losses.

  PFPN_GRAPH_TRAIN=0 python tools/sac_learn_probe.py [--steps 1500] [--every 25] [--seed 33406] [--worker k=v ...] [--net k=v ...]
"""
import argparse
import json
import os
import sys

os.environ.setdefault("PFPN_GRAPH_TRAIN", "0")
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))


def kv(items):
    out = {}
    for it in items or []:
        k, v = it.split("=")
        out[k] = int(v) if v.lstrip("-").isdigit() else float(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--every", type=int, default=25)
    ap.add_argument("--seed", type=int, default=33406)
    ap.add_argument("--net", action="append")
    ap.add_argument("--worker", action="append")
    ap.add_argument("--env", action="append", help="attributes set on the synthetic environment (reward_scale=..., no_fall=1)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import _learning
    from pfpn_amd import envs
    env_over = kv(args.env)
    if env_over:
        plain = envs.SyntheticDeepMimicEnv.step

        def step(self, a):
            s, r, done, info = plain(self, a)
            r *= env_over.get("reward_scale", 1.0)
            if env_over.get("no_fall") and done and not info.get("TimeLimit.truncated") and self.t < self.episode_length:
                done = False
            return s, r, done, info
        envs.SyntheticDeepMimicEnv.step = step
    from pfpn_amd.networks.actor_critic import sac as sac_mod
    cls = sac_mod.ParticleFilteringSACNetwork
    plain_cg = cls.compute_gradients
    rows = []

    def compute_gradients(self, *a, **kw):
        out = plain_cg(self, *a, **kw)
        n = self.global_step
        if n % args.every == 0:
            act, lp = self.last["a"], self.last["logp"]
            q = self.last.get("q_pi")
            l3 = self.sac_out["loss3"].cpu().numpy()
            rows.append({"step": n, "alpha": float(torch.exp(self.store["alpha/log_alpha"]).item()),
                         "logp": float(lp.mean().item()), "abs_a": float(act.abs().mean().item()),
                         "edge": float((act.abs() > 0.95).float().mean().item()),
                         "q_pi": float(q.mean().item()) if q is not None else None,
                         "reward": float(a[2].mean().item()), "alive": float(a[3].mean().item()),
                         "policy_loss": float(l3[1]), "value_loss": float(l3[2])})
            print(json.dumps(rows[-1]), flush=True)
        return out
    cls.compute_gradients = compute_gradients
    worker, rec = _learning.launch("sac", args.seed, args.steps, worker_over=dict({"buffer_capacity": 20000}, **kv(args.worker)),
                                   net_over=kv(args.net), quiet=True)
    print(json.dumps({"episodes": [(e[0], e[2], round(e[3], 1)) for e in rec.episodes[:60]]}))


if __name__ == "__main__":
    main()
