"""Shared comparison helpers of the GPU parity tests."""
import numpy as np


def assert_close_mostly(got, ref, rtol, atol, frac=3e-3, err_msg=""):
    """allclose except for a tiny fraction of elements: a pre-activation that rounds to the other side of
    a ReLU6 kink on the GPU flips one unit's mask for one row, which changes one column of dW."""
    bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
    assert bad.sum() <= max(4, frac * bad.size), "%s: %d / %d elements off (max abs %.3g)" % (err_msg, bad.sum(), bad.size, np.abs(got - ref).max())
    # and the tensors agree globally
    assert np.linalg.norm(got - ref) <= 2e-2 * (np.linalg.norm(ref) + 1e-30), err_msg

