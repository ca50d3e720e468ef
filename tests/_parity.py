"""Shared comparison helpers of the GPU parity tests."""
import numpy as np


def assert_close_mostly(got, ref, rtol, atol, frac=3e-3, err_msg=""):
    """allclose except for a tiny fraction of elements: a pre-activation that rounds to the other side of
    a ReLU6 kink on the GPU flips one unit's mask for one row, which changes one column of dW."""
    bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
    assert bad.sum() <= max(4, frac * bad.size), "%s: %d / %d elements off (max abs %.3g)" % (err_msg, bad.sum(), bad.size, np.abs(got - ref).max())
    # and the tensors agree globally
    assert np.linalg.norm(got - ref) <= 2e-2 * (np.linalg.norm(ref) + 1e-30), err_msg



def close_scaled(got, ref, tol, name="", floor=0.0, flips=0):
    """|got - ref| <= tol * max(max|ref|, floor) for all but `flips` entries (non-finite entries must be identical).
    The scale of a tensor -- not of each element -- is what fp32 sums with cancellation can hold; `flips` admits the
    few entries a ReLU6 mask bit that rounds the other way on the GPU moves (one unit of one row)."""
    got, ref = np.asarray(got, np.float64).reshape(np.shape(ref)), np.asarray(ref, np.float64)
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(got)) and np.array_equal(got[~fin], ref[~fin]), name
    if not fin.any():
        return
    scale = max(np.abs(ref[fin]).max(), floor)
    err = np.abs(got[fin] - ref[fin])
    assert (err > tol * scale).sum() <= flips, "%s: %d / %d entries beyond %.1e x %.3g (max abs err %.3g)" % (
        name, (err > tol * scale).sum(), err.size, tol, scale, err.max())


def away_from_clip(logp, logp_old, eps=0.2, margin=5e-4):
    """PPO's surrogate is discontinuous in the ratio at 1 -+ eps: a row whose ratio sits within fp32 log-prob error
    (~1e-4 at |logp| ~ 40) of a bound takes the other branch on the other implementation and moves every gradient
    by 1/B of a row.  Returns logp_old with such rows reset to ratio 1."""
    ratio = np.exp(np.asarray(logp, np.float64) - np.asarray(logp_old, np.float64))
    near = (np.abs(ratio - (1 - eps)) < margin) | (np.abs(ratio - (1 + eps)) < margin)
    out = np.array(logp_old, copy=True)
    out[near] = np.asarray(logp)[near]
    return out


def near_kink_rows(P, x, margin=2e-6):
    """Rows of the normalised batch x for which some hidden unit's pre-activation lies within `margin` of a ReLU6 kink
    (0 or 6) in either trunk.  Relu6Grad is a step function: on such a row the GPU's GEMM (other summation order,
    pre-activations differ by ~1e-7) may take the other branch, which moves a whole dW column by one row's
    contribution.  The step tests re-draw those rows so that gradients can be compared without a flip allowance."""
    bad = np.zeros(len(x), bool)
    for net in ("actor", "critic"):
        h = x
        for layer in ("fc1", "fc2"):
            pre = h @ P["%s/%s/weight" % (net, layer)] + P["%s/%s/bias" % (net, layer)]
            bad |= (np.minimum(np.abs(pre), np.abs(pre - 6.0)) < margin).any(axis=1)
            h = np.minimum(np.maximum(pre, 0), 6).astype(np.float32)
    return bad
