"""An INDEPENDENT SAC-PFPN learner on the CPU: plain PyTorch autograd, written from the reference's files -- not from
oracle/pfpn_oracle.py and not from the HIP kernels (no hand-derived backward anywhere: the only gradients written out are the
reference's own `tf.custom_gradient`s).  Purpose: a statistical twin of `tools/sac_learn_probe.py` -- when the HIP product's
SAC run on the synthetic environment does something surprising (the policy runs into the corners of the action box within
~350 learner steps), does an implementation that shares NO code with it do the same?

What is restated, with the reference lines:
  networks/utils.py:85-134     MixtureGaussianDistribution.log_prob (probability space, `foo` NaN/Inf guard, tanh correction)
  networks/utils.py:144-186    .sample, rsample branch: RelaxedOneHotCategorical(T = 1) sample w, p ~ Normal(loc, scale),
                               argmax one-hot m, custom gradients `mask2` (pre-tanh) and `mask` (tanh)
  networks/actor_critic/a2c.py:476-523   particle grid in tanh mode
  networks/actor_critic/sac.py:33-40, 97-175   log_alpha = 0, twin Q on [normalised state, action], vf_target, q_target,
                               value loss, policy loss with the alpha term, target entropy = -A
  networks/actor_critic/sac.py:58-73     soft target update (tau = 0.005), hard copy at start
  networks/utils.py:46-68      state normaliser, moving-average form
  models/workers/base_worker.py:25-120   critic Adam on value_loss w.r.t. critic variables, actor Adam on policy_loss w.r.t.
                               actor variables + log_alpha, ONE joint clip_by_global_norm(1.0)
  models/workers/ddpg.py:29-94 replay ring, a learner step per environment step once the ring holds a minibatch
  settings/deepmimic/deepmimic_sac_base.py  batch 256, lr 1e-4 / 1e-4, gamma 0.95

  python tools/sac_twin_torch.py --steps 1500 --every 25 [--seed 33406]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)

S, A, N = 197, 36, 35


def trunc_normal(shape, std, g):
    t = torch.empty(shape)
    torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=g)
    return t.requires_grad_()


def mlp_params(sizes, g):
    ps = []
    for i, o in zip(sizes[:-1], sizes[1:]):
        ps += [trunc_normal((i, o), 0.01, g), torch.zeros(o, requires_grad=True)]
    return ps


def mlp(x, ps, last_linear=True):
    n = len(ps) // 2
    for l in range(n):
        x = x @ ps[2 * l] + ps[2 * l + 1]
        if not (last_linear and l == n - 1):
            x = torch.clamp(x, 0.0, 6.0)                          # tf.nn.relu6
    return x


class Mask2(torch.autograd.Function):                             # utils.py:165-176
    @staticmethod
    def forward(ctx, w, p, m, tanh_p):
        ctx.save_for_backward(m, tanh_p)
        return m * p

    @staticmethod
    def backward(ctx, dy):
        m, tanh_p = ctx.saved_tensors
        tanh_t = (m * tanh_p).sum(-1, keepdim=True)
        gap = (tanh_p - tanh_t) / torch.clamp(1 - tanh_t ** 2, min=1e-6)
        return gap * dy, m * dy, None, None


class Mask(torch.autograd.Function):                              # utils.py:177-185
    @staticmethod
    def forward(ctx, w, p, m):
        ctx.save_for_backward(m, p)
        return m * p

    @staticmethod
    def backward(ctx, dy):
        m, p = ctx.saved_tensors
        t = (m * p).sum(-1, keepdim=True)
        return (p - t) * dy, m * dy, None


class Foo(torch.autograd.Function):                               # utils.py:109-117
    @staticmethod
    def forward(ctx, p):
        return p.clone()

    @staticmethod
    def backward(ctx, dy):
        return torch.where(torch.isnan(dy) | torch.isinf(dy), torch.zeros_like(dy), dy)


def rsample(logits, loc, logstd, g):
    B = logits.shape[0]
    u = torch.rand(B, A, N, generator=g).clamp_(1e-10, 1.0)
    gumbel = -torch.log(-torch.log(u))
    w = torch.softmax(logits + gumbel, -1)                        # RelaxedOneHotCategorical(1.0, logits).sample()
    p = loc + torch.exp(logstd) * torch.randn(B, A, N, generator=g)
    m = torch.nn.functional.one_hot(w.argmax(-1), N).to(w.dtype)
    tanh_p = torch.tanh(p)
    s_ = Mask2.apply(w, p, m, tanh_p).sum(-1)
    a = Mask.apply(w, tanh_p, m).sum(-1)
    return a, s_


def log_prob(logits, loc, logstd, s_):                            # utils.py:108-134, value = (a, s_)
    scale = torch.exp(logstd)
    z = (s_.unsqueeze(-1) - loc) / scale
    phi = torch.exp(-0.5 * z * z) / (scale * math.sqrt(2 * math.pi))
    p = Foo.apply((torch.softmax(logits, -1) * phi).sum(-1))
    lp = torch.log(p) - 2 * (math.log(2.0) - s_ - torch.nn.functional.softplus(-2 * s_))
    return lp.sum(-1)


def run(steps=1500, every=25, seed=33406, env_name="DeepMimicKick-v0", log_alpha0=0.0, threads=6, verbose=True,
        lr_actor=1e-4, lr_critic=1e-4):
    """-> (rows, episodes): one diagnostics row every `every` learner steps, (step, length, return) per finished episode"""
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    from pfpn_amd.networks.actor_critic.a2c import particle_grid   # host NumPy restatement of a2c.py:476-523
    loc0, logstd0 = particle_grid(A, N, tanh_mode=True)
    loc = torch.from_numpy(loc0).clone().requires_grad_()
    logstd = torch.from_numpy(logstd0).clone().requires_grad_()
    actor = mlp_params([S, 1024, 512, A * N], g)
    q1, q2 = mlp_params([S + A, 1024, 512, 1], g), mlp_params([S + A, 1024, 512, 1], g)
    q1t, q2t = [p.detach().clone() for p in q1], [p.detach().clone() for p in q2]
    log_alpha = torch.tensor(float(log_alpha0), requires_grad=True)
    actor_vars = actor + [loc, logstd, log_alpha]
    critic_vars = q1 + q2
    opt_c = torch.optim.Adam(critic_vars, lr=lr_critic, eps=1e-8)
    opt_a = torch.optim.Adam(actor_vars, lr=lr_actor, eps=1e-8)
    mean, std = torch.zeros(S), torch.ones(S)
    gamma, tau, Bsz = 0.95, 0.005, 256

    def norm(s):
        return torch.clamp((s - mean) / std, -5.0, 5.0)

    def policy(x):
        return mlp(x, actor).view(-1, A, N)

    from pfpn_amd.envs import make_env
    env = make_env(env_name, seed=seed)
    env.seed(seed + 1)
    nprng = np.random.RandomState(seed + 1)
    cap = 20000
    ring = {k: np.zeros((cap,) + sh, np.float32) for k, sh in
            (("state", (S,)), ("action", (A,)), ("reward", ()), ("not_terminal", ()), ("state_", (S,)))}
    size = ptr = step = 0
    s = env.reset()
    ep_ret = ep_len = 0
    episodes, rows, t0 = [], [], time.time()
    while step < steps:
        with torch.no_grad():
            a, _ = rsample(policy(norm(torch.from_numpy(s)[None])), loc, logstd, g)
        act = a[0].numpy()
        s2, r, over, info = env.step(np.clip(act, -1, 1))
        ended = over and not info.get("TimeLimit.truncated", False)
        for k, v in (("state", s), ("action", act), ("reward", r), ("not_terminal", 0.0 if ended else 1.0), ("state_", s2)):
            ring[k][ptr] = v
        ptr, size = (ptr + 1) % cap, min(size + 1, cap)
        ep_ret += r
        ep_len += 1
        s = s2
        if over:
            episodes.append((step, ep_len, round(ep_ret, 1)))
            s, ep_ret, ep_len = env.reset(), 0.0, 0
        if size < Bsz:
            continue
        idx = nprng.choice(size, Bsz)
        b = {k: torch.from_numpy(v[idx]) for k, v in ring.items()}
        # state normaliser: the step normalises with the statistics of the previous apply, the update uses this batch
        x, x2 = norm(b["state"]), norm(b["state_"])
        alpha = torch.exp(log_alpha).detach()
        logits = policy(x)
        a_pi, s_pi = rsample(logits, loc, logstd, g)
        logp = log_prob(logits, loc, logstd, s_pi)
        q1p = mlp(torch.cat([x, a_pi], -1), q1).squeeze(-1)
        q2p = mlp(torch.cat([x, a_pi], -1), q2).squeeze(-1)
        q1r = mlp(torch.cat([x, b["action"]], -1), q1).squeeze(-1)
        q2r = mlp(torch.cat([x, b["action"]], -1), q2).squeeze(-1)
        with torch.no_grad():                                     # target net: shared actor, own Q copies (sac.py:141-153)
            lg2 = policy(x2)
            a2, s2_ = rsample(lg2, loc, logstd, g)
            lp2 = log_prob(lg2, loc, logstd, s2_)
            xa2 = torch.cat([x2, a2], -1)
            vf = torch.minimum(mlp(xa2, q1t).squeeze(-1), mlp(xa2, q2t).squeeze(-1)) - alpha * lp2
            q_target = b["reward"] + gamma * b["not_terminal"] * vf
        value_loss = ((q_target - q1r) ** 2 + (q_target - q2r) ** 2).mean() * 0.5          # value_loss_coef
        l = alpha * logp - torch.minimum(q1p, q2p) - log_alpha * (logp + (-A)).detach()
        policy_loss = l.mean()
        gc = torch.autograd.grad(value_loss, critic_vars, retain_graph=True)
        ga = torch.autograd.grad(policy_loss, actor_vars)
        total = torch.sqrt(sum((t ** 2).sum() for t in gc + ga))
        scale = 1.0 / max(float(total), 1.0)                      # tf.clip_by_global_norm(., 1.0) over all of them jointly
        for p_, g_ in zip(critic_vars, gc):
            p_.grad = g_ * scale
        for p_, g_ in zip(actor_vars, ga):
            p_.grad = g_ * scale
        opt_c.step()
        opt_a.step()
        with torch.no_grad():
            for t_, p_ in zip(q1t + q2t, q1 + q2):
                t_.mul_(1 - tau).add_(tau * p_)
            d = min(0.9999, (1 + step) / (10 + step))             # utils.py:60-68
            m_, v_ = b["state"].mean(0), b["state"].var(0, unbiased=False)
            mean = d * mean + (1 - d) * m_
            std = torch.clamp(d * std + (1 - d) * torch.sqrt(v_), min=1e-6)
        if step % every == 0:
            row = {"step": step, "alpha": float(alpha), "logp": float(logp.detach().mean()),
                   "abs_a": float(a_pi.detach().abs().mean()), "edge": float((a_pi.detach().abs() > 0.95).float().mean()),
                   "q_pi": float(torch.minimum(q1p, q2p).detach().mean()), "reward": float(b["reward"].mean()),
                   "policy_loss": float(policy_loss.detach()), "value_loss": float(value_loss.detach()),
                   "sec": round(time.time() - t0, 1)}
            rows.append(row)
            if verbose:
                print(json.dumps(row), flush=True)
        step += 1
    return rows, episodes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--every", type=int, default=25)
    ap.add_argument("--seed", type=int, default=33406)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--log-alpha", type=float, default=0.0, help="initial log_alpha (reference: 0)")
    ap.add_argument("--env", default="DeepMimicKick-v0", help="DeepMimic*: the synthetic stand-in; SyntheticReach-v0")
    ap.add_argument("--lr-actor", type=float, default=1e-4)
    ap.add_argument("--lr-critic", type=float, default=1e-4)
    ap.add_argument("--out", default=os.path.join(R, "gpurun_out", "sac_twin_torch.jsonl"))
    args = ap.parse_args()
    rows, episodes = run(args.steps, args.every, args.seed, args.env, args.log_alpha, args.threads, lr_actor=args.lr_actor,
                         lr_critic=args.lr_critic)
    out = {"run": "sac-twin-torch:%d:%d" % (args.seed, args.steps), "env": args.env, "log_alpha0": args.log_alpha, "rows": rows,
           "episodes": episodes[:80]}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "a") as f:
        f.write(json.dumps(out) + "\n")
    print(json.dumps({"episodes": episodes[:40]}))


if __name__ == "__main__":
    main()
