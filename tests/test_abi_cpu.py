"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/pfpn_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from pfpn_amd import build
    return build.build()


def test_library_builds_and_exports_header(lib_path):
    from pfpn_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 28
    lib = ctypes.CDLL(lib_path)
    for name in protos:
        assert hasattr(lib, name), "libpfpn_hip.so does not export %s" % name
    # and nothing pfpn_* is exported that the header does not declare
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (pfpn_\w+)", syms))
    assert exported == set(protos), exported ^ set(protos)


def test_every_export_has_a_host_wrapper():
    """every entry point of the header is called from pfpn_amd/ (a wrapper deleted by accident shows up here, not as a
    missing key in a bench line)."""
    from pfpn_amd import _lib
    src = ""
    for d, _, files in os.walk(os.path.join(ROOT, "pfpn_amd")):
        for f in files:
            if f.endswith(".py"):
                src += open(os.path.join(d, f)).read()
    diagnostics = {"pfpn_abi_version", "pfpn_debug_stamps"}        # called from tests / tools/stamp_probe.py only
    missing = [name for name in _lib.parse_header() if name not in diagnostics and not re.search(r"\b%s\b" % name, src)]
    assert not missing, missing


def test_loader_attaches_prototypes(lib_path):
    from pfpn_amd import _lib
    lib = _lib.load()
    assert lib.pfpn_abi_version() == 1
    assert lib.pfpn_mixture_workspace_bytes(4096, 36, 35) > 4096 * 36 * 4
    assert lib.pfpn_logprob_fwd.argtypes[4] is ctypes.c_int and lib.pfpn_logprob_fwd.restype is ctypes.c_int


def test_argument_validation_without_gpu(lib_path):
    """null pointers / bad shapes are rejected on the host before any launch."""
    from pfpn_amd import _lib
    lib = _lib.load()
    assert lib.pfpn_logprob_fwd(None, None, None, None, 4, 36, 35, 0, None, None, None, 0, None) == 1
    assert lib.pfpn_gae(None, None, None, 1, 0.95, 0.9, None, None) == 1
    assert lib.pfpn_tf_adam(None, None, None, None, 10, 1e-4, 0.9, 0.999, 1e-8, None, 1, None, 1.0, None, None) == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pfpn_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PfpnLibraryError):
        _lib.load()


def test_no_cpu_fallback_in_product():
    """the shipped package never imports the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pfpn_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\S*oracle", src, re.M), "%s imports the oracle" % f
                assert "pfpn_oracle" not in src, "%s references the oracle module" % f


def test_network_needs_a_gpu():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pfpn_amd.networks import ParticleFilteringClipPPONetwork
    net = ParticleFilteringClipPPONetwork(trainable=True, state_shape=[197], action_shape=[36],
                                          action_lower_bound=-np.ones(36), action_upper_bound=np.ones(36),
                                          common_net_shape=[], critic_net_shape=[64], actor_net_shape=[64],
                                          particles=35, resample=-1)
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        net.init()
