"""ctypes loader of the oracle's C twin (oracle/pfpn_cpu_twin.c).  TEST / BENCH INFRASTRUCTURE: nothing under pfpn_amd/
imports this.  `load()` builds the library with `make -C oracle` when it is missing."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libpfpn_cpu_twin.so")
_lib = None
F = np.float32


def load():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "pfpn_cpu_twin.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = ctypes.CDLL(LIB)
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.c_int
        _lib.pfpn_logprob_fwd_cpu.argtypes = [fp, fp, fp, fp, ip, ip, ip, ip, fp, fp]
        _lib.pfpn_logprob_bwd_cpu.argtypes = [fp, fp, fp, fp, fp, ip, ip, ip, ip, fp, fp, fp, fp]
        _lib.pfpn_gae_cpu.argtypes = [fp, fp, ctypes.POINTER(ctypes.c_longlong), ip, ctypes.c_double, ctypes.c_double, fp]
        _lib.pfpn_cpu_set_threads.argtypes = [ip]
        _lib.pfpn_cpu_set_threads.restype = None
        for f in (_lib.pfpn_logprob_fwd_cpu, _lib.pfpn_logprob_bwd_cpu, _lib.pfpn_gae_cpu, _lib.pfpn_cpu_threads):
            f.restype = ip
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _c(a):
    return np.ascontiguousarray(a, dtype=F)


def threads():
    return int(load().pfpn_cpu_threads())


def set_threads(n):
    """OpenMP threads of the following calls (a box may show 256 CPUs and grant 16: more threads than cores is slower)"""
    load().pfpn_cpu_set_threads(int(n))


def logprob_fwd(logits, x, loc, logstd, tanh_mode=False):
    """-> (logp [B], p [B,A])"""
    logits, x, loc, logstd = _c(logits), _c(x), _c(loc), _c(logstd)
    B, A, N = logits.shape
    logp, p = np.empty(B, F), np.empty((B, A), F)
    rc = load().pfpn_logprob_fwd_cpu(_p(logits), _p(x), _p(loc), _p(logstd), B, A, N, int(tanh_mode), _p(logp), _p(p))
    assert rc == 0, rc
    return logp, p


def logprob_bwd(logits, x, loc, logstd, dlogp, tanh_mode=False):
    """-> (dlogits [B,A,N], dloc [A,N], dlogstd [A,N], dx [B,A])"""
    logits, x, loc, logstd, dlogp = _c(logits), _c(x), _c(loc), _c(logstd), _c(dlogp)
    B, A, N = logits.shape
    dl, dloc, dls, dx = np.empty((B, A, N), F), np.empty((A, N), F), np.empty((A, N), F), np.empty((B, A), F)
    rc = load().pfpn_logprob_bwd_cpu(_p(logits), _p(x), _p(loc), _p(logstd), _p(dlogp), B, A, N, int(tanh_mode),
                                     _p(dl), _p(dloc), _p(dls), _p(dx))
    assert rc == 0, rc
    return dl, dloc, dls, dx


def gae(reward, value, offsets, gamma, gae_gamma):
    """E ragged episodes: reward [n], value [n + E], offsets [E + 1] -> adv [n]"""
    reward, value = _c(reward), _c(value)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    adv = np.empty(reward.shape[0], F)
    rc = load().pfpn_gae_cpu(_p(reward), _p(value), off.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), len(off) - 1,
                             float(gamma), float(gae_gamma), _p(adv))
    assert rc == 0, rc
    return adv
