"""All-core CPU restatement of the DPPO-PFPN learner step on torch-CPU tensors (fp32).

*** TEST / BENCH INFRASTRUCTURE -- NOT PRODUCT CODE (same rule as pfpn_oracle.py). ***
Purpose: `bench.py`'s `cpu_baseline` needs a CPU path that actually USES the host's cores.  The NumPy oracle is an
op-for-op restatement whose elementwise half is single-threaded (on a 256-thread host it is slower with 64 BLAS threads
than with one).  This module states the same step with torch-CPU ops: the forward is the reference's op chain
(networks/utils.py:108-134, networks/actor_critic/ppo.py:31-54, actor_critic.py:132-184) in probability space with the
`foo` NaN/Inf guard as a custom autograd function (utils.py:109-117), the backward is torch autograd (TF builds it with
`tf.gradients`), `clip_by_global_norm` and TF-Adam are written out (base_worker.py:97-102, training_ops ApplyAdam).
GEMMs and elementwise kernels both run on torch's intra-op thread pool.  tests/test_oracle_math.py checks it against
pfpn_oracle.dppo_grads / dppo_learner_step, so the number it produces is the number of the right computation.
"""
import numpy as np
import torch

HALF_LOG_2PI = 0.9189385175704956
ORDER = ["actor/fc1/weight", "actor/fc1/bias", "actor/fc2/weight", "actor/fc2/bias",
         "actor/samples", "actor/samples_std", "actor/fc_policy/weight", "actor/fc_policy/bias",
         "critic/fc1/weight", "critic/fc1/bias", "critic/fc2/weight", "critic/fc2/bias",
         "critic/fc3/weight", "critic/fc3/bias"]


class _Guard(torch.autograd.Function):
    """`foo` of networks/utils.py:109-117: identity forward, NaN / Inf incoming gradients become 0."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return torch.where(torch.isnan(g) | torch.isinf(g), torch.zeros_like(g), g)


class TorchCPUState(object):
    def __init__(self, oracle_state):
        st = oracle_state
        self.A, self.N = st.A, st.N
        self.P = {k: torch.from_numpy(np.array(st.P[k], np.float32)).requires_grad_(True) for k in ORDER}
        self.m = {k: torch.from_numpy(np.array(st.m[k], np.float32)) for k in ORDER}
        self.v = {k: torch.from_numpy(np.array(st.v[k], np.float32)) for k in ORDER}
        self.b1p, self.b2p = float(st.b1p), float(st.b2p)
        self.mean = torch.from_numpy(np.array(st.norm_mean, np.float32))
        self.std = torch.from_numpy(np.array(st.norm_std, np.float32))
        self.step = int(st.global_step)


def losses(st, batch):
    """-> (loss, policy_loss, value_loss*0.5, logp, value); differentiable w.r.t. st.P."""
    P, A, N = st.P, st.A, st.N
    s = batch["state"]
    x = torch.clamp((s - st.mean) / st.std, -5.0, 5.0)

    def fc(h, name, act):
        y = h @ P[name + "/weight"] + P[name + "/bias"]
        return torch.clamp(y, 0.0, 6.0) if act else y      # relu6; its subgradient at the kinks matches Relu6Grad a.e.
    ha = fc(fc(x, "actor/fc1", True), "actor/fc2", True)
    logits = fc(ha, "actor/fc_policy", False).view(-1, A, N)
    hc = fc(fc(x, "critic/fc1", True), "critic/fc2", True)
    value = fc(hc, "critic/fc3", False)[:, 0]
    w = torch.softmax(logits, dim=-1)
    loc, logstd = P["actor/samples"], P["actor/samples_std"]
    scale = torch.exp(logstd)
    z = (batch["action"][..., None] - loc) / scale
    phi = torch.exp(-0.5 * z * z - (HALF_LOG_2PI + torch.log(scale)))
    p = _Guard.apply((w * phi).sum(-1))
    logp = torch.log(p).sum(-1)
    adv = batch["advantage"]
    nadv = (adv - adv.mean()) / (torch.sqrt(((adv - adv.mean()) ** 2).mean()) + 1e-8)
    ratio = torch.exp(logp - batch["log_prob"])
    policy_loss = -torch.minimum(ratio * nadv, torch.clamp(ratio, 0.8, 1.2) * nadv).mean()
    target = adv + batch["value"]
    value_loss = 0.5 * ((value - target) ** 2).mean()
    return policy_loss + value_loss, policy_loss, value_loss, logp, value


def learner_step(st, batch, lr=1e-4, clip_norm=1.0, b1=0.9, b2=0.999, eps=1e-8):
    """one single-replica learner step, in place on `st`; returns (loss, policy_loss, value_loss)."""
    loss, pl, vl, _, _ = losses(st, batch)
    grads = torch.autograd.grad(loss, [st.P[k] for k in ORDER])
    with torch.no_grad():
        s = batch["state"]
        d = min(0.9999, (1.0 + st.step) / (10.0 + st.step))
        mean_b = s.mean(0)
        var_b = ((s - mean_b) ** 2).mean(0)
        new_mean = d * st.mean + (1 - d) * mean_b
        new_std = torch.clamp(d * st.std + (1 - d) * torch.sqrt(var_b), min=1e-6)
        norm = torch.sqrt(2.0 * sum(0.5 * (g * g).sum() for g in grads))
        scale = clip_norm * torch.minimum(1.0 / norm, torch.tensor(1.0 / clip_norm))
        alpha = lr * np.sqrt(1.0 - st.b2p) / (1.0 - st.b1p)
        for k, g in zip(ORDER, grads):
            g = g * scale
            st.m[k] += (g - st.m[k]) * (1 - b1)
            st.v[k] += (g * g - st.v[k]) * (1 - b2)
            st.P[k] -= (st.m[k] * alpha) / (torch.sqrt(st.v[k]) + eps)
        st.b1p, st.b2p = float(np.float32(st.b1p) * np.float32(b1)), float(np.float32(st.b2p) * np.float32(b2))
        st.mean, st.std = new_mean, new_std
        st.step += 1
    return loss.item(), pl.item(), vl.item()
