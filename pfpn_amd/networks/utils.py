"""Mixture-of-particles distribution, GAE scan and the online state normaliser on MI355X.

Host-side mirror of reference networks/utils.py: `MixtureGaussianDistribution` (:85-236), `discount`
(:5-15) and `online_normalizer` (:46-82).  All arithmetic runs in the HIP kernels of libpfpn_hip.so;
there is no CPU path here.
"""
import numpy as np
import torch

from .. import kernels


class MixtureGaussianDistribution(object):
    """Per action dimension, a mixture of N Gaussians ("particles") with state-independent
    loc/logstd [A,N] and state-dependent categorical logits [B,A,N] (reference utils.py:86-101).

    `normalize_output` selects the tanh-squashed (SAC) formulation: `value` is then the tuple
    (tanh-action, pre-tanh action) like the reference's `log_prob((a, s_))`.
    Random draws take explicit noise tensors (or a torch.Generator) so that CPU oracle and GPU agree
    bit-for-bit on the drawn particle index."""

    def __init__(self, logits, loc, logstd, normalize_output=False):
        A, N = loc.shape
        self.A, self.N = A, N
        self.logits = logits.view(-1, A * N)
        self.loc, self.logstd = loc, logstd
        self.normalize_output = bool(normalize_output)
        self.dis_action = None

    @property
    def scale(self):
        return self.logstd.exp()

    @property
    def batch(self):
        return self.logits.shape[0]

    # -- log-prob ------------------------------------------------------------------------------
    def log_prob(self, value):
        """reference utils.py:108-134.  value: [B,A]; in tanh mode a tuple (a, s_) or the tanh action."""
        x = self._pre_tanh(value)
        return kernels.logprob_fwd(self.logits, x, self.loc, self.logstd, self.normalize_output)

    def prob(self, value):
        return self.log_prob(value).exp()          # utils.py:103-106

    def log_prob_backward(self, value, dlogp, want_dx=False):
        """Gradient of sum_b dlogp[b]*log_prob[b] w.r.t. logits / loc / logstd (/ pre-tanh value)."""
        x = self._pre_tanh(value)
        return kernels.logprob_bwd(self.logits, x, self.loc, self.logstd, dlogp, self.normalize_output, want_dx)

    def _pre_tanh(self, value):
        if self.normalize_output:
            if isinstance(value, (tuple, list)):
                return value[1].contiguous()
            return torch.atanh(value).contiguous()  # utils.py:124
        return value.contiguous()

    # -- entropy -------------------------------------------------------------------------------
    def entropy(self):
        """categorical entropy of the weights, [B,A] (utils.py:146-151)."""
        return kernels.entropy_fwd(self.logits, self.A, self.N)

    def entropy_backward(self, dH, dlogits=None):
        return kernels.entropy_bwd(self.logits, dH, self.A, self.N, dlogits)

    # -- sampling ------------------------------------------------------------------------------
    def make_noise(self, generator=None):
        B, dev = self.batch, self.logits.device
        if self.normalize_output:
            # TFP 0.7 ExpRelaxedOneHotCategorical: u ~ U(tiny, 1), g = -log(-log u)
            gumbel = kernels.gumbel_from_uniform_(torch.rand(B, self.A * self.N, device=dev, generator=generator))
            eps = torch.randn(B, self.A * self.N, device=dev, generator=generator)
            return {"gumbel": gumbel, "eps": eps}
        u = torch.rand(B, self.A, device=dev, dtype=torch.float64, generator=generator)
        eps = torch.randn(B, self.A, device=dev, generator=generator)
        return {"u_cat": u, "eps_sel": eps}

    def sample(self, n=1, noise=None, generator=None, max_active=None, sum_active=None, want_logp=False):
        """reference utils.py:153-200.  Plain branch -> action [B,A] (and log-prob of it when asked);
        rsample branch (normalize_output) -> (tanh action, pre-tanh action)."""
        assert n == 1
        noise = noise or self.make_noise(generator)
        self.noise = noise
        if self.normalize_output:
            a, s_, k = kernels.rsample_fwd(self.logits, self.loc, self.logstd, noise["gumbel"], noise["eps"])
            self.dis_action = k
            if max_active is not None:
                kernels.active_stats(self.logits, max_active, sum_active)
            return a, s_
        k, action, logp = kernels.sample(self.logits, self.loc, self.logstd, noise["u_cat"], noise["eps_sel"],
                                         max_active, sum_active, want_logp)
        self.dis_action = k
        return (action, logp) if want_logp else action

    def sample_backward(self, da, ds, dlogits=None, dloc=None, dlogstd=None):
        """backward of the rsample branch's custom gradients (utils.py:165-183)."""
        n = self.noise
        return kernels.rsample_bwd(self.logits, self.loc, self.logstd, n["gumbel"], n["eps"], da, ds,
                                   dlogits, dloc, dlogstd)

    def mean(self):
        """deterministic action (utils.py:202-236): loc[a, argmax]; tanh(loc[...]) in tanh mode."""
        k, a = kernels.mean_action(self.logits, self.loc, self.normalize_output)
        self.dis_action = k
        return a


def discount(val, factor, boostrap_val=0.0, normalize=False, device=None):
    """reference utils.py:5-15: reverse scan v_t = val_t + factor*v_{t+1} with a float64 accumulator.
    Implemented with the GAE kernel (gamma = 0 turns its delta into `val`)."""
    if normalize:
        raise NotImplementedError("normalize=True is unused by the PFPN settings")
    val = np.asarray(val, dtype=np.float32)
    T = len(val)
    device = device or torch.device("cuda", torch.cuda.current_device())
    r = torch.from_numpy(val).to(device)
    v = torch.zeros(T + 1, dtype=torch.float32, device=device)
    offs = torch.tensor([0, T], dtype=torch.int64, device=device)
    out = kernels.gae(r, v, offs, 0.0, float(factor))
    if boostrap_val:
        # the scan is linear: add factor^(T-t) * bootstrap
        k = torch.arange(T, 0, -1, device=device, dtype=torch.float64)
        out = (out.double() + (float(factor) ** k) * float(boostrap_val)).float()
    return out.cpu().numpy()


class OnlineNormalizer:
    """online_normalizer(moving_average=True), reference utils.py:46-68: mean/std EMA with decay
    min(0.9999, (1+step)/(10+step)) on population moments of each training batch."""

    def __init__(self, shape, device):
        self.mean = torch.zeros(shape, dtype=torch.float32, device=device)
        self.std = torch.ones(shape, dtype=torch.float32, device=device)

    def normalize(self, s, clip, out=None):
        return kernels.state_norm(s, self.mean, self.std, clip, out)

    def updated(self, s, step_dev, mean_out, std_out):
        """EMA statistics after seeing batch `s` (does not modify self)."""
        return kernels.state_norm_update(s, step_dev, self.mean, self.std, mean_out, std_out)
