"""Thin tensor-level wrappers over the C ABI (libpfpn_hip.so).

PyTorch is plumbing here: device memory + the current HIP stream.  Every function validates
shapes/dtypes on the host before launching (a faulting kernel can take the whole node down) and
launches on torch's current stream, so calls can be captured in a torch.cuda.CUDAGraph.
"""
import ctypes
import os

import torch

from . import _lib

_ws = {}


def _stream():
    """raw handle of torch's current HIP stream (the private accessor is ~10x cheaper than building a Stream object,
    and this runs once per kernel launch)."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    except AttributeError:
        return torch.cuda.current_stream().cuda_stream


def _f32(t, *shape, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous float32 CUDA tensor" % name)
    if shape and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t.data_ptr()


def workspace(nbytes, device):
    """Grow-only per-device scratch buffer (never reallocated inside a captured region if it was
    sized by a warm-up call first)."""
    # one buffer per (device, stream): kernels on a side stream must not share slabs with the main stream
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def _mix_ws(B, A, N, device):
    lib = _lib.load()
    nbytes = lib.pfpn_mixture_workspace_bytes(B, A, N)
    buf = workspace(nbytes, device)
    return buf.data_ptr(), buf.numel()


def logprob_fwd(logits, x, loc, logstd, tanh_mode=False, want_pmix=False, out=None):
    """logits [B,A,N] (or [B,A*N]), x [B,A], loc/logstd [A,N] -> logp [B] (, pmix [B,A]).  `out`: a [B] tensor to write into."""
    lib = _lib.load()
    A, N = loc.shape
    B = x.shape[0]
    pl = _f32(logits.view(B, A * N), B, A * N, name="logits")
    px = _f32(x, B, A, name="x")
    ploc, pls = _f32(loc, A, N, name="loc"), _f32(logstd, A, N, name="logstd")
    logp = torch.empty(B, dtype=torch.float32, device=x.device) if out is None else out
    _f32(logp, B, name="logp")
    pmix = torch.empty(B, A, dtype=torch.float32, device=x.device) if want_pmix else None
    ws, wsn = _mix_ws(B, A, N, x.device)
    _lib.check(lib.pfpn_logprob_fwd(pl, px, ploc, pls, B, A, N, int(tanh_mode), logp.data_ptr(),
                                    pmix.data_ptr() if want_pmix else None, ws, wsn, _stream()),
               "pfpn_logprob_fwd")
    return (logp, pmix) if want_pmix else logp


def logprob_bwd(logits, x, loc, logstd, dlogp, tanh_mode=False, want_dx=False, out=None, slabs_ws=None, dbias=None):
    """-> dlogits [B,A,N], dloc [A,N], dlogstd [A,N] (, dx [B,A]).  `out`: dict with preallocated `dlogits`, `dloc`,
    `dlogstd` to write into (no 1-GiB temporaries at the 128 x 256 shape).
    `dbias` [A*N] (not with `slabs_ws`): also the column sums of dlogits -- inside the launch where the wide kernel serves
    the grid (N >= 64), by a pass over dlogits otherwise.
    `slabs_ws` (a private uint8 workspace, mixture_workspace()): dloc / dlogstd are NOT finalised -- their per-block
    partial sums stay in it (logprob_bwd_slabs() gives the geometry) for a deferred column-sum job."""
    lib = _lib.load()
    A, N = loc.shape
    B = x.shape[0]
    pl = _f32(logits.view(B, A * N), B, A * N, name="logits")
    px = _f32(x, B, A, name="x")
    ploc, pls = _f32(loc, A, N, name="loc"), _f32(logstd, A, N, name="logstd")
    pg = _f32(dlogp, B, name="dlogp")
    if out is not None:
        dlogits, dloc, dlogstd = out["dlogits"].view(B, A, N), out["dloc"], out["dlogstd"]
        _f32(dlogits, B, A, N, name="dlogits"); _f32(dloc, A, N, name="dloc"); _f32(dlogstd, A, N, name="dlogstd")
    else:
        dlogits = torch.empty(B, A, N, dtype=torch.float32, device=x.device)
        dloc = torch.empty(A, N, dtype=torch.float32, device=x.device)
        dlogstd = torch.empty(A, N, dtype=torch.float32, device=x.device)
    dx = torch.empty(B, A, dtype=torch.float32, device=x.device) if want_dx else None
    ws, wsn = _mix_ws(B, A, N, x.device) if slabs_ws is None else (slabs_ws.data_ptr(), slabs_ws.numel())
    if dbias is not None:
        if slabs_ws is not None:
            raise ValueError("dbias needs the finished form (no slabs_ws)")
        rc = lib.pfpn_logprob_bwd_bias(pl, px, ploc, pls, pg, B, A, N, int(tanh_mode), dlogits.data_ptr(), dloc.data_ptr(),
                                       dlogstd.data_ptr(), dx.data_ptr() if want_dx else None,
                                       _f32(dbias.view(-1), A * N, name="dbias"), ws, wsn, _stream())
        if rc != 1:
            _lib.check(rc, "pfpn_logprob_bwd_bias")
            return (dlogits, dloc, dlogstd, dx) if want_dx else (dlogits, dloc, dlogstd)
    _lib.check(lib.pfpn_logprob_bwd(pl, px, ploc, pls, pg, B, A, N, int(tanh_mode), dlogits.data_ptr(),
                                    dloc.data_ptr(), dlogstd.data_ptr(), dx.data_ptr() if want_dx else None,
                                    int(slabs_ws is None), ws, wsn, _stream()), "pfpn_logprob_bwd")
    if dbias is not None:           # (a grid outside the wide kernel: the column sums as a pass of their own)
        act_bwd_bias_grad(None, dlogits.view(B, A * N), dbias.view(1, -1), relu6=False)
    return (dlogits, dloc, dlogstd, dx) if want_dx else (dlogits, dloc, dlogstd)


def ppo_wide_fused(logits, action, loc, logstd, logp_old, adv, value, value_old, stats, clip_eps=0.2, value_coef=0.5,
                   out=None, dbias=None):
    """The 256-particle PPO pass as ONE launch that reads the logits once (pfpn_ppo_wide_fused), called directly (probes,
    tests; ppo_mixture_fwd_bwd takes it by itself where it applies).  `out`: dict with logp, dlogits, dloc, dlogstd, dvalue, loss3 (+ `dlogp`); returns (out, chunks,
    err_flag tensor) -- err_flag != 0 after a synchronise means a bounded in-kernel wait expired (results invalid)."""
    lib = _lib.load()
    A, N = loc.shape
    B = action.shape[0]
    dev = action.device
    if out is None:
        out = {"logp": torch.empty(B, dtype=torch.float32, device=dev), "dlogits": torch.empty(B, A * N, dtype=torch.float32, device=dev),
               "dloc": torch.empty(A, N, dtype=torch.float32, device=dev), "dlogstd": torch.empty(A, N, dtype=torch.float32, device=dev),
               "dvalue": torch.empty(B, dtype=torch.float32, device=dev), "loss3": torch.empty(3, dtype=torch.float32, device=dev)}
    if "dlogp" not in out:
        out["dlogp"] = _dlogp_scratch(B, dev)
    nbytes = int(lib.pfpn_mixture_workspace_bytes(B, A, N))
    buf = workspace(nbytes, dev)
    chunks = ctypes.c_int(0)
    _lib.check(lib.pfpn_ppo_wide_fused(_f32(logits.view(B, A * N), B, A * N), _f32(action, B, A), _f32(loc, A, N), _f32(logstd, A, N),
                                       _f32(logp_old, B), _f32(adv, B), _f32(value, B), _f32(value_old, B), _f32(stats, 4),
                                       B, A, N, float(clip_eps), float(value_coef), _f32(out["logp"], B),
                                       _f32(out["dlogits"].view(B, A * N), B, A * N), _f32(out["dloc"], A, N),
                                       _f32(out["dlogstd"], A, N), _f32(out["dvalue"], B), _f32(out["dlogp"], B),
                                       _f32(dbias.view(-1), A * N) if dbias is not None else None, _f32(out["loss3"], 3),
                                       ctypes.addressof(chunks), fault_word(dev).data_ptr(), buf.data_ptr(), buf.numel(),
                                       _stream()), "pfpn_ppo_wide_fused")
    # the ticket area is the last 4224 words of the pfpn_mixture_workspace_bytes() prefix of the workspace; word 65 = the flag
    tick = buf[:nbytes].view(torch.int32)[-(128 + 4096):]
    return out, chunks.value, tick[65:66]


def mixture_workspace(B, A, N, device):
    """a private workspace of the mixture kernels' size (slabs that must survive until a deferred reduction)"""
    return torch.empty(int(_lib.load().pfpn_mixture_workspace_bytes(B, A, N)), dtype=torch.uint8, device=device)


def rollout_workspace(B, A, N, device):
    return torch.empty(int(_lib.load().pfpn_rollout_workspace_bytes(B, A, N)), dtype=torch.uint8, device=device)


def logprob_bwd_slabs(logits, dlogits, B, A, N):
    """(nslab, stride) of the partial sums logprob_bwd(slabs_ws=...) leaves at the start of its workspace"""
    n, st = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.load().pfpn_logprob_bwd_slabs(logits.data_ptr(), dlogits.data_ptr(), B, A, N, ctypes.addressof(n),
                                                  ctypes.addressof(st)), "pfpn_logprob_bwd_slabs")
    return n.value, st.value


def rsample_bwd_slabs(B, A, N):
    """(nslab, stride, float_offset) of the partial sums rsample_bwd(slabs_ws=...) leaves in its workspace"""
    n, st, off = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_longlong(0)
    _lib.check(_lib.load().pfpn_rsample_bwd_slabs(B, A, N, ctypes.addressof(n), ctypes.addressof(st),
                                                  ctypes.addressof(off)), "pfpn_rsample_bwd_slabs")
    return n.value, st.value, off.value


def new_adv_stats(device, clip=None):
    """the 4-float {mean, rstd, clip_lo, clip_hi} array of the loss heads; `clip`: None | c (-> +-c) | (lo, hi)."""
    lo, hi = -float("inf"), float("inf")
    if clip:
        lo, hi = (min(clip), max(clip)) if hasattr(clip, "__len__") else (-float(clip), float(clip))
    return torch.tensor([0.0, 1.0, lo, hi], dtype=torch.float32, device=device)


def adv_stats(adv, normalize=True, out=None):
    """moments of the advantages into out[0:2] (out[2:4] = the clip bounds are the caller's; a fresh `out` has none)."""
    lib = _lib.load()
    B = adv.shape[0]
    pa = _f32(adv, B, name="advantage")
    if out is None:
        out = new_adv_stats(adv.device)
    _f32(out, 4, name="adv_stats")
    _lib.check(lib.pfpn_adv_stats(pa, B, int(normalize), out.data_ptr(), _stream()), "pfpn_adv_stats")
    return out


_dlogp_bufs = {}
# the one-launch (2L) PPO pass of the 256-particle shapes (PFPN_WIDE_FUSED=0: the two-launch composition everywhere --
# the setting for a GPU this process does not have to itself: the pass's waves wait for each other, see DeviceFault)
WIDE_FUSED = os.environ.get("PFPN_WIDE_FUSED", "1") != "0"


class DeviceFault(RuntimeError):
    """A kernel whose waves wait for each other (the one-launch PPO pass) gave up a bounded wait: its gradients are NaN and
    every parameter the optimiser touched since is invalid.  Raised by check_faults() at the next host synchronisation."""


_fault_words = {}


def fault_word(device):
    """The package's sticky fault word of `device` (int32[1], zeroed once): handed to every entry point that records a
    failure it cannot return (pfpn_ppo_wide_fused's `fault_flag`).  Nothing on the device ever clears it."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    w = _fault_words.get(idx)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the first one-launch PPO pass of a device must run eagerly (warm-up) before a capture: "
                               "its fault word would otherwise live in the graph's private pool")
        w = _fault_words[idx] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def check_faults():
    """Read the fault words (one 4-byte copy per device that has one; nothing at all for shapes that never took a waiting
    kernel) and raise DeviceFault if a launch recorded a failure since the process started.  Called wherever the host
    synchronises anyway: loss read-back, end of a training round, checkpoint save."""
    for idx, w in _fault_words.items():
        if int(w.item()):
            raise DeviceFault(
                "cuda:%d: a one-launch PPO pass (pfpn_ppo_wide_fused) gave up waiting for its own waves -- something else "
                "held compute units while it ran (another process or stream, a CU mask).  Its gradients were NaN and the "
                "optimiser has applied them: the parameters of this run are lost from that step on; restart from the last "
                "checkpoint with PFPN_WIDE_FUSED=0 (the two-launch composition needs no co-residency)." % idx)


def clear_faults():
    """tests: forget recorded faults"""
    for w in _fault_words.values():
        w.zero_()


def _dlogp_scratch(B, device):
    # (per stream, like workspace(): two steps in flight on two streams must not share d loss / d logp)
    key = (B, device.index, _stream())
    if key not in _dlogp_bufs:
        _dlogp_bufs[key] = torch.empty(B, dtype=torch.float32, device=device)
    return _dlogp_bufs[key]


def ppo_mixture_fwd_bwd(logits, action, loc, logstd, logp_old, adv, value, value_old, stats,
                        clip_eps=0.2, value_coef=0.5, out=None, dbias=None, finalize=True, ws=None):
    """Fused learner kernel.  Returns dict(logp, dlogits, dloc, dlogstd, dvalue, loss3).
    `dbias` [A*N] (optional): receives the column sums of dlogits (fc_policy bias gradient).
    `finalize=False` leaves dloc / dlogstd / dbias / loss3 as per-block slabs in `ws` (a private uint8 workspace)
    for `reduce_jobs` (geometry: `ppo_mixture_slabs`)."""
    lib = _lib.load()
    A, N = loc.shape
    B = action.shape[0]
    dev = action.device
    pl = _f32(logits.view(B, A * N), B, A * N, name="logits")
    pa = _f32(action, B, A, name="action")
    ploc, pls = _f32(loc, A, N, name="loc"), _f32(logstd, A, N, name="logstd")
    ptrs = [_f32(logp_old, B, name="log_prob"), _f32(adv, B, name="advantage"), _f32(value, B, name="value"),
            _f32(value_old, B, name="value_old"), _f32(stats, 4, name="adv_stats")]
    if out is None:
        out = {
            "logp": torch.empty(B, dtype=torch.float32, device=dev),
            "dlogits": torch.empty(B, A * N, dtype=torch.float32, device=dev),
            "dloc": torch.empty(A, N, dtype=torch.float32, device=dev),
            "dlogstd": torch.empty(A, N, dtype=torch.float32, device=dev),
            "dvalue": torch.empty(B, dtype=torch.float32, device=dev),
            "loss3": torch.empty(3, dtype=torch.float32, device=dev),
        }
    _f32(out["dlogits"].view(B, A * N), B, A * N, name="dlogits")
    _f32(out["dloc"], A, N, name="dloc"); _f32(out["dlogstd"], A, N, name="dlogstd")
    _f32(out["logp"], B, name="logp"); _f32(out["dvalue"], B, name="dvalue"); _f32(out["loss3"], 3, name="loss3")
    ws, wsn = _mix_ws(B, A, N, dev) if ws is None else (ws.data_ptr(), ws.numel())
    rc = lib.pfpn_ppo_mixture_fwd_bwd(pl, pa, ploc, pls, *ptrs, B, A, N, float(clip_eps),
                                      float(value_coef), out["logp"].data_ptr(),
                                      out["dlogits"].data_ptr(), out["dloc"].data_ptr(),
                                      out["dlogstd"].data_ptr(), out["dvalue"].data_ptr(),
                                      _f32(dbias.view(-1), A * N, name="dbias") if dbias is not None else None,
                                      out["loss3"].data_ptr() if finalize else None, ws, wsn, _stream())
    if rc == 1 and B >= 1 and finalize:
        # shape outside the fused kernel's coverage (N > 64, no template instance, A*N % 4 != 0): the
        # documented composition of the three HIP entry points (include/pfpn_hip.h)
        # (the row log-prob couples all A segments of a row; at N >= 64 a row and the [A,N] bank no longer fit on one
        # CU, so the second pass over the logits is structural: 3L of traffic)
        # TWO launches (round 4): the forward's row sums AND the PPO head happen inside the forward launch, the backward's slab
        # sums inside the backward launch, and every result lands in the caller's buffers (no copies)
        # (walking a > 256 MiB batch in row chunks that fit the infinity cache, so that a chunk's backward re-reads its logits
        # from cache, was built and measured in round 4: 806 us against 570 us on the 1 GiB shape -- 33 launches of
        # 96-MiB chunks cost more in start-up and tail than the cache gives back: LAB_NOTEBOOK.md)
        dlogp = _dlogp_scratch(B, dev)
        # ONE launch, the logits read once (round 5: k_mix_wide_f2l; 415 us against 598 us on the 1 GiB shape) where that form
        # exists -- 64 <= N <= 256 particles per dim, A % 4 == 0, A <= 128, a batch of at least 8 row chunks -- and WIDE_FUSED is on.
        # It needs the device to itself while it runs (its waves wait for each other): the package launches on one stream.
        # A wait that expires (a shared GPU) leaves NaN gradients AND a sticky mark in fault_word(dev): check_faults() raises
        # at the next loss read-back / checkpoint.
        if WIDE_FUSED:
            rc = lib.pfpn_ppo_wide_fused(pl, pa, ploc, pls, *ptrs, B, A, N, float(clip_eps), float(value_coef),
                                         out["logp"].data_ptr(), out["dlogits"].data_ptr(), out["dloc"].data_ptr(),
                                         out["dlogstd"].data_ptr(), out["dvalue"].data_ptr(), dlogp.data_ptr(),
                                         _f32(dbias.view(-1), A * N, name="dbias") if dbias is not None else None,
                                         out["loss3"].data_ptr(), None, fault_word(dev).data_ptr(), ws, wsn, _stream())
            if rc != 1:
                _lib.check(rc, "pfpn_ppo_wide_fused")
                return out
        rc = lib.pfpn_logprob_fwd_ppo(pl, pa, ploc, pls, *ptrs, B, A, N, float(clip_eps), float(value_coef),
                                      out["logp"].data_ptr(), dlogp.data_ptr(), out["dvalue"].data_ptr(),
                                      out["loss3"].data_ptr(), ws, wsn, _stream())
        if rc == 1:        # (A % 4 != 0, very long row chunks: forward and loss head as launches of their own)
            logprob_fwd(logits, action, loc, logstd, out=out["logp"])
            dlogp, _, _ = ppo_loss(out["logp"], logp_old, adv, value, value_old, stats, clip_eps, value_coef,
                                   dvalue=out["dvalue"], loss3=out["loss3"])
        else:
            _lib.check(rc, "pfpn_logprob_fwd_ppo")
        # straight into the caller's buffers; the head's bias gradient inside the backward launch (a third slab plane)
        logprob_bwd(logits, action, loc, logstd, dlogp, out=out, dbias=dbias)
        return out
    _lib.check(rc, "pfpn_ppo_mixture_fwd_bwd")
    return out


# --------------------------------------------------------------------------------------- #
# rollout                                                                                  #
# --------------------------------------------------------------------------------------- #
def _typed(t, dtype, *shape, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError("%s must be a contiguous %s CUDA tensor" % (name, dtype))
    if shape and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t.data_ptr()


def _roll_ws(B, A, N, device):
    lib = _lib.load()
    buf = workspace(lib.pfpn_rollout_workspace_bytes(B, A, N), device)
    return buf.data_ptr(), buf.numel()


def ppo_loss(logp, logp_old, adv, value, value_old, stats, clip_eps=0.2, value_coef=0.5, dvalue=None, loss3=None):
    """-> dlogp [B], dvalue [B], loss3; `dvalue` / `loss3`: tensors to write into."""
    lib = _lib.load()
    B = logp.shape[0]
    dev = logp.device
    dlogp = torch.empty(B, dtype=torch.float32, device=dev)
    dvalue = torch.empty(B, dtype=torch.float32, device=dev) if dvalue is None else dvalue
    loss3 = torch.empty(3, dtype=torch.float32, device=dev) if loss3 is None else loss3
    _f32(dvalue, B, name="dvalue"); _f32(loss3, 3, name="loss3")
    _lib.check(lib.pfpn_ppo_loss(_f32(logp, B), _f32(logp_old, B), _f32(adv, B), _f32(value, B), _f32(value_old, B),
                                 _f32(stats, 4), B, float(clip_eps), float(value_coef), dlogp.data_ptr(),
                                 dvalue.data_ptr(), loss3.data_ptr(), _stream()), "pfpn_ppo_loss")
    return dlogp, dvalue, loss3


def vtrace(logp_target, logp_behavior, value, reward, not_terminal, E, gamma, rho_clip=1.0, pg_rho_clip=1.0,
           seq_len=None, want_mask=False):
    """V-trace value targets and policy-gradient advantages; all inputs flat [E*T] episode-major.
    -> vs, pg_adv (, mask)."""
    lib = _lib.load()
    n = logp_target.shape[0]
    if E < 1 or n % E:
        raise ValueError("V-trace: %d rows do not split into %d episodes" % (n, E))
    T = n // E
    dev = logp_target.device
    vs = torch.empty(n, dtype=torch.float32, device=dev)
    pg = torch.empty(n, dtype=torch.float32, device=dev)
    mask = torch.empty(n, dtype=torch.float32, device=dev) if want_mask else None
    _lib.check(lib.pfpn_vtrace(_f32(logp_target, n), _f32(logp_behavior, n), _f32(value, n), _f32(reward, n),
                               _f32(not_terminal, n),
                               _typed(seq_len, torch.int32, E, name="seq_len") if seq_len is not None else None,
                               E, T, float(gamma), float(rho_clip or 0.0), float(pg_rho_clip or 0.0),
                               vs.data_ptr(), pg.data_ptr(), mask.data_ptr() if want_mask else None, _stream()),
               "pfpn_vtrace")
    return (vs, pg, mask) if want_mask else (vs, pg)


def a2c_loss(logp, adv, value, value_target, stats, value_coef=0.5, mask=None):
    """-> dlogp [B], dvalue [B], loss4 = {loss, policy_loss, value_loss*coef, n_valid}."""
    lib = _lib.load()
    B = logp.shape[0]
    dev = logp.device
    dlogp = torch.empty(B, dtype=torch.float32, device=dev)
    dvalue = torch.empty(B, dtype=torch.float32, device=dev)
    loss4 = torch.empty(4, dtype=torch.float32, device=dev)
    _lib.check(lib.pfpn_a2c_loss(_f32(logp, B), _f32(adv, B), _f32(value, B), _f32(value_target, B), _f32(stats, 4),
                                 _f32(mask, B) if mask is not None else None, B, float(value_coef),
                                 dlogp.data_ptr(), dvalue.data_ptr(), loss4.data_ptr(), _stream()), "pfpn_a2c_loss")
    return dlogp, dvalue, loss4


def sample(logits, loc, logstd, u_cat, eps_sel, max_active=None, sum_active=None, want_logp=True):
    """-> k [B,A] int32, action [B,A], logp [B] (or None); updates max/sum_active in place if given."""
    lib = _lib.load()
    A, N = loc.shape
    B = u_cat.shape[0]
    dev = u_cat.device
    pl = _f32(logits.view(B, A * N), B, A * N, name="logits")
    k = torch.empty(B, A, dtype=torch.int32, device=dev)
    action = torch.empty(B, A, dtype=torch.float32, device=dev)
    logp = torch.empty(B, dtype=torch.float32, device=dev) if want_logp else None
    ws, wsn = _roll_ws(B, A, N, dev)
    _lib.check(lib.pfpn_sample(pl, _f32(loc, A, N), _f32(logstd, A, N), _typed(u_cat, torch.float64, B, A, name="u_cat"),
                               _f32(eps_sel, B, A, name="eps_sel"), B, A, N, k.data_ptr(), action.data_ptr(),
                               logp.data_ptr() if want_logp else None,
                               _f32(max_active, A, N) if max_active is not None else None,
                               _f32(sum_active, A, N) if sum_active is not None else None, ws, wsn, _stream()),
               "pfpn_sample")
    return k, action, logp


def active_stats(logits, max_active, sum_active):
    lib = _lib.load()
    A, N = max_active.shape
    B = logits.shape[0]
    ws, wsn = _roll_ws(B, A, N, logits.device)
    _lib.check(lib.pfpn_active_stats(_f32(logits.view(B, A * N), B, A * N), B, A, N, _f32(max_active, A, N),
                                     _f32(sum_active, A, N), ws, wsn, _stream()), "pfpn_active_stats")


def mean_action(logits, loc, tanh_mode=False):
    lib = _lib.load()
    A, N = loc.shape
    B = logits.shape[0]
    k = torch.empty(B, A, dtype=torch.int32, device=logits.device)
    action = torch.empty(B, A, dtype=torch.float32, device=logits.device)
    _lib.check(lib.pfpn_mean_action(_f32(logits.view(B, A * N), B, A * N), _f32(loc, A, N), B, A, N, int(tanh_mode),
                                    k.data_ptr(), action.data_ptr(), _stream()), "pfpn_mean_action")
    return k, action


def entropy_fwd(logits, A, N):
    lib = _lib.load()
    B = logits.shape[0]
    H = torch.empty(B, A, dtype=torch.float32, device=logits.device)
    _lib.check(lib.pfpn_entropy_fwd(_f32(logits.view(B, A * N), B, A * N), B, A, N, H.data_ptr(), _stream()),
               "pfpn_entropy_fwd")
    return H


def entropy_bwd(logits, dH, A, N, dlogits=None):
    lib = _lib.load()
    B = logits.shape[0]
    acc = dlogits is not None
    if dlogits is None:
        dlogits = torch.empty(B, A * N, dtype=torch.float32, device=logits.device)
    _lib.check(lib.pfpn_entropy_bwd(_f32(logits.view(B, A * N), B, A * N), _f32(dH, B, A), B, A, N, int(acc),
                                    _f32(dlogits.view(B, A * N), B, A * N), _stream()), "pfpn_entropy_bwd")
    return dlogits


def rsample_fwd(logits, loc, logstd, gumbel, eps):
    lib = _lib.load()
    A, N = loc.shape
    B = logits.shape[0]
    dev = logits.device
    k = torch.empty(B, A, dtype=torch.int32, device=dev)
    a = torch.empty(B, A, dtype=torch.float32, device=dev)
    s_ = torch.empty(B, A, dtype=torch.float32, device=dev)
    _lib.check(lib.pfpn_rsample_fwd(_f32(logits.view(B, A * N), B, A * N), _f32(loc, A, N), _f32(logstd, A, N),
                                    _f32(gumbel.view(B, A * N), B, A * N), _f32(eps.view(B, A * N), B, A * N), B, A, N,
                                    k.data_ptr(), a.data_ptr(), s_.data_ptr(), _stream()), "pfpn_rsample_fwd")
    return a, s_, k


def rsample_bwd(logits, loc, logstd, gumbel, eps, da, ds, dlogits=None, dloc=None, dlogstd=None, slabs_ws=None):
    """accumulates into dlogits/dloc/dlogstd when all three are given.  `slabs_ws` (private, rollout_workspace()): the
    particle-grid partial sums stay in it instead of being added to dloc / dlogstd (rsample_bwd_slabs())."""
    lib = _lib.load()
    A, N = loc.shape
    B = logits.shape[0]
    dev = logits.device
    acc = dlogits is not None
    if not acc:
        dlogits = torch.empty(B, A * N, dtype=torch.float32, device=dev)
        dloc = torch.empty(A, N, dtype=torch.float32, device=dev)
        dlogstd = torch.empty(A, N, dtype=torch.float32, device=dev)
    ws, wsn = _roll_ws(B, A, N, dev) if slabs_ws is None else (slabs_ws.data_ptr(), slabs_ws.numel())
    _lib.check(lib.pfpn_rsample_bwd(_f32(logits.view(B, A * N), B, A * N), _f32(loc, A, N), _f32(logstd, A, N),
                                    _f32(gumbel.view(B, A * N), B, A * N), _f32(eps.view(B, A * N), B, A * N),
                                    _f32(da, B, A), _f32(ds, B, A), B, A, N, int(acc),
                                    _f32(dlogits.view(B, A * N), B, A * N), _f32(dloc, A, N), _f32(dlogstd, A, N),
                                    int(slabs_ws is None), ws, wsn, _stream()), "pfpn_rsample_bwd")
    return dlogits, dloc, dlogstd


def gumbel_from_uniform_(u):
    """in place: u in [0,1) -> Gumbel(0,1) noise -log(-log(max(u, tiny))), one launch."""
    lib = _lib.load()
    _lib.check(lib.pfpn_gumbel_from_uniform(_f32(u), u.numel(), _stream()), "pfpn_gumbel_from_uniform")
    return u


def sac_noise_fill_(buf, seed, offset, epoch=None):
    """buf [4, ...]: rows 0,1 <- Gumbel(0,1), rows 2,3 <- N(0,1), drawn on the device (Philox4x32-10 keyed by `seed`,
    block counter (element, `offset`)) -- one launch, no uniform fill.  `epoch` (an int64 device scalar): the offset is read
    from it when the launch runs (the learner step's own draw counter, kernels.sac_loss(noise_epoch=...) advances it)."""
    lib = _lib.load()
    if buf.shape[0] != 4:
        raise ValueError("buf must be [4, ...]")
    if epoch is not None and (epoch.dtype != torch.int64 or not epoch.is_cuda or epoch.numel() != 1):
        raise ValueError("epoch must be a one-element int64 device tensor")
    _lib.check(lib.pfpn_sac_noise_fill(_f32(buf), buf.numel() // 4, int(seed) & 0xFFFFFFFFFFFFFFFF,
                                       int(offset) & 0xFFFFFFFFFFFFFFFF, epoch.data_ptr() if epoch is not None else None,
                                       _stream()), "pfpn_sac_noise_fill")
    return buf


def sac_noise_from_uniform_(buf):
    """buf [4, ...] uniforms in [0,1) -> rows 0,1 Gumbel(0,1), rows 2,3 N(0,1) (Box-Muller pair), in place, one launch."""
    lib = _lib.load()
    if buf.shape[0] != 4:
        raise ValueError("expected a [4, ...] buffer")
    _lib.check(lib.pfpn_sac_noise_from_uniform(_f32(buf), buf[0].numel(), _stream()), "pfpn_sac_noise_from_uniform")
    return buf


def resample(loc, logstd, bias, weight, max_active, sum_active, u_resample, u_noise, threshold=0.0,
             tanh_mode=False, topk=0, fixed_sigma=0.0):
    """In place on loc/logstd/bias/weight.  Returns (dead [A,N] int32, src [A,N] int32).
    topk = 0: categorical source draw (`resample=-1`); topk = k: uniform choice among the k most active (`resample=k`)."""
    lib = _lib.load()
    A, N = loc.shape
    H = weight.shape[0]
    dev = loc.device
    dead = torch.empty(A, N, dtype=torch.int32, device=dev)
    src = torch.empty(A, N, dtype=torch.int32, device=dev)
    _lib.check(lib.pfpn_resample(_f32(loc, A, N), _f32(logstd, A, N), _f32(bias, A * N), _f32(weight, H, A * N), H,
                                 _f32(max_active, A, N), _f32(sum_active, A, N),
                                 _typed(u_resample, torch.float64, A, N, name="u_resample"), _f32(u_noise, A, N),
                                 A, N, float(threshold), int(tanh_mode), int(topk), float(fixed_sigma or 0.0), dead.data_ptr(), src.data_ptr(),
                                 _stream()), "pfpn_resample")
    return dead, src


# --------------------------------------------------------------------------------------- #
# learner utilities                                                                        #
# --------------------------------------------------------------------------------------- #
def _learn_ws(rows, H, device):
    lib = _lib.load()
    buf = workspace(max(lib.pfpn_learner_workspace_bytes(rows, H), 4096 * 4), device)
    return buf.data_ptr(), buf.numel()


def gae(reward, value, offsets, gamma, gae_gamma):
    """reward [Ttot] f32, value [Ttot+E] f32, offsets [E+1] int64 (device) -> adv [Ttot] f32."""
    lib = _lib.load()
    E = offsets.shape[0] - 1
    adv = torch.empty_like(reward)
    _lib.check(lib.pfpn_gae(_f32(reward), _f32(value), _typed(offsets, torch.int64, E + 1, name="offsets"), E,
                            float(gamma), float(gae_gamma), adv.data_ptr(), _stream()), "pfpn_gae")
    return adv


def state_norm(s, mean, std, clip=5.0, out=None):
    lib = _lib.load()
    B, S = s.shape
    if out is None:
        out = torch.empty_like(s)
    _lib.check(lib.pfpn_state_norm(_f32(s, B, S), _f32(mean, S), _f32(std, S), B, S, float(clip or 0.0),
                                   _f32(out, B, S), _stream()), "pfpn_state_norm")
    return out


def state_norm_update(s, step, mean, std, mean_out=None, std_out=None):
    lib = _lib.load()
    B, S = s.shape
    if mean_out is None:
        mean_out, std_out = torch.empty_like(mean), torch.empty_like(std)
    if not isinstance(step, torch.Tensor):
        step = torch.tensor([float(step)], dtype=torch.float32, device=s.device)
    ws, wsn = _learn_ws(B, S, s.device)
    _lib.check(lib.pfpn_state_norm_update(_f32(s, B, S), B, S, _f32(step, 1), _f32(mean, S), _f32(std, S),
                                          _f32(mean_out, S), _f32(std_out, S), ws, wsn, _stream()),
               "pfpn_state_norm_update")
    return mean_out, std_out


def state_prologue(s, mean, std, clip, out_x, want_moments=True, adv=None, normalize_adv=False, adv_stats=None,
                   ws=None, s2=None, out_x2=None):
    """x = clip((s-mean)/std) + float64 column moments of s (for state_norm_ema) + advantage moments, one launch.
    `ws`: private uint8 workspace for the moments (deferred EMA: `job_norm_ema`)."""
    lib = _lib.load()
    B, S = s.shape
    ws, wsn = _learn_ws(B, S, s.device) if ws is None else (ws.data_ptr(), ws.numel())
    _lib.check(lib.pfpn_state_prologue(_f32(s, B, S), _f32(mean, S), _f32(std, S), B, S, float(clip or 0.0),
                                       _f32(out_x, B, S), int(want_moments),
                                       _f32(adv, B) if adv_stats is not None else None, int(normalize_adv),
                                       _f32(adv_stats, 4) if adv_stats is not None else None,
                                       _f32(s2, B, S) if s2 is not None else None,
                                       _f32(out_x2, B, S) if s2 is not None else None, ws, wsn, _stream()),
               "pfpn_state_prologue")
    return out_x


def state_norm_ema(B, S, step, mean, std, mean_out, std_out):
    """EMA update from the moments state_prologue left in the learner workspace of this stream (call it next)."""
    lib = _lib.load()
    ws, _ = _learn_ws(B, S, mean.device)
    _lib.check(lib.pfpn_state_norm_ema(ws, B, S, _f32(step, 1), _f32(mean, S), _f32(std, S), _f32(mean_out, S),
                                       _f32(std_out, S), _stream()), "pfpn_state_norm_ema")


def bias_act(y, bias, relu6=True, group_rows=0):
    lib = _lib.load()
    rows, H = y.shape
    _lib.check(lib.pfpn_bias_act(_f32(y, rows, H), _f32(bias), rows, H, int(group_rows), int(relu6), _stream()),
               "pfpn_bias_act")
    return y


def act_bwd_bias_grad(y, dy, db, relu6=True, group_rows=0):
    lib = _lib.load()
    rows, H = dy.shape
    ws, wsn = _learn_ws(rows, H, dy.device)
    _lib.check(lib.pfpn_act_bwd_bias_grad(_f32(y, rows, H) if y is not None else None, _f32(dy, rows, H), rows, H,
                                          int(group_rows), int(relu6), _f32(db), ws, wsn, _stream()),
               "pfpn_act_bwd_bias_grad")
    return dy, db


def learner_workspace(rows, H, device):
    """a PRIVATE workspace (uint8 tensor) big enough for the chunk slabs of a [rows, H] column reduction; producers
    whose reductions are deferred to `reduce_jobs` each need their own."""
    lib = _lib.load()
    return torch.empty(int(max(lib.pfpn_learner_workspace_bytes(rows, H), 4096 * 4)), dtype=torch.uint8, device=device)


def relu6_mask(rows, H, device):
    """bit-mask buffer of a [rows, H] ReLU6 layer (bias_act_mask writes it, act_bwd_partial(mask=...) reads it)"""
    lib = _lib.load()
    return torch.empty(int(lib.pfpn_relu6_mask_bytes(rows, H)), dtype=torch.uint8, device=device)


def bias_act_mask(y, bias, mask, group_rows=0):
    """y <- relu6(y + bias) and the Relu6Grad mask as bits (1 bit per element) into `mask`."""
    lib = _lib.load()
    rows, H = y.shape
    _lib.check(lib.pfpn_bias_act_mask(_f32(y, rows, H), _f32(bias), rows, H, int(group_rows), mask.data_ptr(),
                                      mask.numel(), _stream()), "pfpn_bias_act_mask")
    return y


def bias_act_value(y, bias, group_rows, value_group, value_w, value_b, v, relu6=True):
    """last trunk layer: y <- act(y + bias) and v = y[value_group] . value_w + value_b in ONE pass.
    value_group = -1: every group has its own head -- value_w [G,H], value_b [G], v [G, group_rows]."""
    lib = _lib.load()
    rows, H = y.shape
    nh = rows // group_rows if value_group < 0 else 1
    bstride = value_b.stride(0) if (value_group < 0 and value_b.dim() == 2) else 1      # [G, pad]: bias at [g, 0]
    if value_b.numel() < (nh - 1) * bstride + 1:
        raise ValueError("value_b too short")
    _lib.check(lib.pfpn_bias_act_value(_f32(y, rows, H), _f32(bias), rows, H, int(group_rows), int(relu6),
                                       int(value_group), _f32(value_w.view(-1), nh * H), _f32(value_b), int(bstride),
                                       _f32(v.view(-1), nh * group_rows), _stream()), "pfpn_bias_act_value")
    return v


class EpilogueJob(ctypes.Structure):
    """mirror of PfpnEpilogueJob (include/pfpn_hip.h): one of the two passes of a paired epilogue launch"""
    _fields_ = [("y", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("rows", ctypes.c_int), ("group_rows", ctypes.c_int),
                ("mask", ctypes.c_void_p), ("mask_bytes", ctypes.c_size_t), ("value_w", ctypes.c_void_p),
                ("value_b", ctypes.c_void_p), ("value_b_stride", ctypes.c_int), ("v", ctypes.c_void_p)]


def epilogue_job(y, bias, group_rows=0, mask=None, value=None):
    """a pass of bias_act_mask_pair (mask: the bit buffer or None) / bias_act_value_pair (value = (w [G,H], b [G(,pad)],
    v [G, group_rows]): a head on every group)."""
    rows, H = y.shape
    j = EpilogueJob(_f32(y, rows, H), _f32(bias), rows, int(group_rows), mask.data_ptr() if mask is not None else None,
                    mask.numel() if mask is not None else 0, None, None, 0, None)
    j._keep = (y, bias, mask, value)
    if value is not None:
        vw, vb, v = value
        nh = rows // group_rows
        bstride = vb.stride(0) if vb.dim() == 2 else 1
        if vb.numel() < (nh - 1) * bstride + 1:
            raise ValueError("value_b too short")
        j.value_w, j.value_b, j.value_b_stride = _f32(vw.view(-1), nh * H), _f32(vb), int(bstride)
        j.v = _f32(v.view(-1), nh * group_rows)
    return j, H


def bias_act_mask_pair(a, b):
    """two BiasAdd + Relu6 passes of one width (each with or without mask bits) in one launch; a, b = epilogue_job(...)"""
    if a[1] != b[1]:
        raise ValueError("the two passes must have one width")
    _lib.check(_lib.load().pfpn_bias_act_mask_pair(ctypes.addressof(a[0]), ctypes.addressof(b[0]), a[1], _stream()),
               "pfpn_bias_act_mask_pair")


def bias_act_value_pair(a, b):
    """two last-layer passes (BiasAdd + Relu6 + a scalar head on every group) in one launch; a, b = epilogue_job(value=...)"""
    if a[1] != b[1]:
        raise ValueError("the two passes must have one width")
    _lib.check(_lib.load().pfpn_bias_act_value_pair(ctypes.addressof(a[0]), ctypes.addressof(b[0]), a[1], _stream()),
               "pfpn_bias_act_value_pair")


class SacLoss(ctypes.Structure):
    """mirror of PfpnSacLoss (include/pfpn_hip.h): the SAC loss head as a rider of the critics' first backward launch"""
    _fields_ = [("q_both", ctypes.c_void_p), ("q_tgt", ctypes.c_void_p), ("logp", ctypes.c_void_p),
                ("logp_t", ctypes.c_void_p), ("reward", ctypes.c_void_p), ("not_terminal", ctypes.c_void_p),
                ("log_alpha", ctypes.c_void_p), ("B", ctypes.c_int), ("gamma", ctypes.c_float),
                ("value_coef", ctypes.c_float), ("target_entropy", ctypes.c_float), ("dq_both", ctypes.c_void_p),
                ("dlogp", ctypes.c_void_p), ("dlog_alpha", ctypes.c_void_p), ("loss3", ctypes.c_void_p),
                ("noise_epoch", ctypes.c_void_p)]


def sac_loss_rider(q_both, q_tgt, logp, logp_t, reward, not_terminal, log_alpha, gamma, value_coef, target_entropy,
                   dq_both, out, noise_epoch=None):
    """sac_loss() for the one-pass critic step (q_both / dq_both [2, 2B]) as an argument of act_bwd_partial(value=(-1, w,
    None, ws_value, rider)): no launch of its own."""
    B = logp.shape[0]
    if noise_epoch is not None and (noise_epoch.dtype != torch.int64 or not noise_epoch.is_cuda or noise_epoch.numel() != 1):
        raise ValueError("noise_epoch must be a one-element int64 device tensor")
    r = SacLoss(_f32(q_both, 2, 2 * B), _f32(q_tgt, 2, B), _f32(logp, B), _f32(logp_t, B), _f32(reward, B),
                _f32(not_terminal, B), _f32(log_alpha.view(-1), 1), B, float(gamma), float(value_coef),
                float(target_entropy), _f32(dq_both, 2, 2 * B), _f32(out["dlogp"], B), _f32(out["dlog_alpha"].view(-1), 1),
                _f32(out["loss3"], 3), noise_epoch.data_ptr() if noise_epoch is not None else None)
    r._keep = (q_both, q_tgt, logp, logp_t, reward, not_terminal, log_alpha, dq_both, out, noise_epoch)
    return r


def act_bwd_partial(y, dy, group_rows, ws, relu6=True, value=None, mask=None):
    """Relu6Grad in place + per-chunk bias-gradient partials left in `ws` (float view [G*cpg, H]); returns cpg.
    value = (group, w [H], dvalue [group_rows], ws_value): fold the critic head's backward into that group.
    value = (-1, w [2, H], None, ws_value, SacLoss): the twin critics' heads with the SAC loss head riding (sac_loss_rider).
    mask (from bias_act_mask, H % 256 == 0): read the mask bits instead of the activations `y`."""
    lib = _lib.load()
    rows, H = dy.shape
    cpg = ctypes.c_int(0)
    if value is not None and len(value) > 4 and value[4] is not None:
        rider = value[4]
        if mask is not None or not relu6 or value[0] >= 0 or rows != 4 * rider.B or group_rows != 2 * rider.B:
            raise ValueError("the SAC loss rides on the twin heads of a [2, 2B]-row pass only")
        if value[3].numel() < 4 * 2 * 1024 * H:
            raise ValueError("value-head partial workspace too small")
        _lib.check(lib.pfpn_act_bwd_partial_sac(_f32(y, rows, H), _f32(dy, rows, H), H, _f32(value[1].view(-1), 2 * H),
                                                value[3].data_ptr(), ws.data_ptr(), ws.numel() // 4,
                                                ctypes.addressof(cpg), ctypes.addressof(rider), _stream()),
                   "pfpn_act_bwd_partial_sac")
        return cpg.value
    if mask is not None:
        if value is not None or not relu6:
            raise ValueError("the mask-bit form has no folded head and is ReLU6 only")
        if mask.numel() < lib.pfpn_relu6_mask_bytes(rows, H):
            raise ValueError("mask buffer too small")
        _lib.check(lib.pfpn_act_bwd_partial_mask(_f32(dy, rows, H), mask.data_ptr(), rows, H, int(group_rows),
                                                 ws.data_ptr(), ws.numel() // 4, ctypes.addressof(cpg), _stream()),
                   "pfpn_act_bwd_partial_mask")
        return cpg.value
    vg, vw, dv, vws = 0, None, None, None
    if value is not None:
        nh = rows // group_rows if value[0] < 0 else 1
        if value[3].numel() < 4 * nh * 1024 * H:
            raise ValueError("value-head partial workspace too small")
        vg, vw, dv, vws = (value[0], _f32(value[1].view(-1), nh * H), _f32(value[2].view(-1), nh * (group_rows or rows)),
                           value[3].data_ptr())
    _lib.check(lib.pfpn_act_bwd_partial(_f32(y, rows, H) if y is not None else None, _f32(dy, rows, H), rows, H,
                                        int(group_rows), int(relu6), int(vg), vw, dv, vws, ws.data_ptr(),
                                        ws.numel() // 4, ctypes.addressof(cpg), _stream()), "pfpn_act_bwd_partial")
    return cpg.value


def ppo_mixture_slabs(B, A, N, want_dbias, logits, dlogits):
    """(nslab, stride) of the slabs ppo_mixture_fwd_bwd(finalize=False) leaves in its workspace."""
    lib = _lib.load()
    ns, sd = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.pfpn_ppo_mixture_slabs(B, A, N, int(want_dbias), logits.data_ptr(), dlogits.data_ptr(),
                                          ctypes.addressof(ns), ctypes.addressof(sd)), "pfpn_ppo_mixture_slabs")
    return ns.value, sd.value


class ReduceJob(ctypes.Structure):
    """mirror of PfpnReduceJob (include/pfpn_hip.h)"""
    _fields_ = [("kind", ctypes.c_int), ("nslab", ctypes.c_int), ("stride", ctypes.c_int), ("n", ctypes.c_int),
                ("src", ctypes.c_void_p), ("out", ctypes.c_void_p), ("out2", ctypes.c_void_p),
                ("in0", ctypes.c_void_p), ("in1", ctypes.c_void_p), ("in2", ctypes.c_void_p),
                ("p0", ctypes.c_float), ("p1", ctypes.c_float), ("norm", ctypes.c_int)]


JOB_COLSUM, JOB_VECSUM, JOB_PPO_LOSS, JOB_NORM_EMA, JOB_SQSUM = 0, 1, 2, 3, 4
MAX_REDUCE_JOBS = 32


def job_colsum(ws, float_offset, nslab, stride, n, out, second=None):
    """out[i] = sum_k ws_f32[float_offset + k*stride + i], i < n.  second = (ws2, float_offset2, nslab2, stride2): plus
    the slab-order sum of a second slab set (out = sum(first) + sum(second))."""
    if out.numel() < n or ws.numel() < 4 * (float_offset + (nslab - 1) * stride + n):
        raise ValueError("column-sum job outside its buffers")
    in0, p0, p1 = None, 0.0, 0.0
    if second is not None:
        ws2, off2, nslab2, stride2 = second
        if ws2.numel() < 4 * (off2 + (nslab2 - 1) * stride2 + n) or nslab2 < 1 or stride2 < n or stride2 >= 1 << 24:
            raise ValueError("second slab set outside its buffer")
        in0, p0, p1 = ws2.data_ptr() + 4 * off2, float(nslab2), float(stride2)
    return ReduceJob(JOB_COLSUM, nslab, stride, n, ws.data_ptr() + 4 * float_offset, _f32(out), None, in0, None, None,
                     p0, p1, 0)


def job_vecsum(vec, out):
    return ReduceJob(JOB_VECSUM, 1, 0, vec.numel(), _f32(vec), _f32(out), None, None, None, None, 0.0, 0.0, 0)


def job_ppo_loss(ws, float_offset, nslab, stride, inv_B, value_coef, loss3):
    return ReduceJob(JOB_PPO_LOSS, nslab, stride, 2, ws.data_ptr() + 4 * float_offset, _f32(loss3, 3), None, None, None,
                     None, float(inv_B), float(value_coef), 0)


def job_norm_ema(ws, B, S, step, mean, std, mean_out, std_out):
    """EMA of the state normaliser from the float64 moments state_prologue(ws=ws) left behind."""
    lib = _lib.load()
    C = lib.pfpn_state_moment_chunks(B)
    if ws.numel() < C * 2 * S * 8:
        raise ValueError("moment workspace too small")
    return ReduceJob(JOB_NORM_EMA, C, B, S, ws.data_ptr(), _f32(mean_out, S), _f32(std_out, S), _f32(step, 1),
                     _f32(mean, S), _f32(std, S), 0.0, 0.0, 0)


def job_sqsum(ptr, n_floats, blocks):
    """sum of squares of a FINISHED gradient segment (address `ptr`, 16-byte aligned, n_floats % 4 == 0), over `blocks`
    blocks; only inside reduce_jobs(..., norm=...)."""
    if ptr % 16 or n_floats % 4 or n_floats < 4 or blocks < 1:
        raise ValueError("sqsum segment must be 16-byte aligned with a multiple of 4 floats")
    return ReduceJob(JOB_SQSUM, int(blocks), 0, int(n_floats), ptr, None, None, None, None, None, 0.0, 0.0, 0)


def reduce_jobs(jobs, norm=None):
    """ONE launch for every deferred fixed-order reduction of the step.
    norm = dict(partial [>= #blocks] f32, beta_pow [G,4] or None, beta1, beta2): the launch also leaves the global-norm
    partial sums (outputs of jobs with .norm = 1 + the JOB_SQSUM segments) and rotates the Adam powers; returns the
    number of partials (for clip_adam(prenorm=...))."""
    lib = _lib.load()
    if not 1 <= len(jobs) <= MAX_REDUCE_JOBS:
        raise ValueError("1..%d reduce jobs per launch" % MAX_REDUCE_JOBS)
    arr = (ReduceJob * len(jobs))(*jobs)
    if norm is None:
        _lib.check(lib.pfpn_reduce_jobs(ctypes.addressof(arr), len(jobs), _stream()), "pfpn_reduce_jobs")
        return 0
    nb = ctypes.c_int(0)
    bp = norm.get("beta_pow")
    G = bp.shape[0] if bp is not None else 0
    _lib.check(lib.pfpn_reduce_jobs_norm(ctypes.addressof(arr), len(jobs), _f32(norm["partial"]), norm["partial"].numel(),
                                         ctypes.addressof(nb), _f32(bp, G, 4) if bp is not None else None, G,
                                         float(norm.get("beta1", 0.9)), float(norm.get("beta2", 0.999)), _stream()),
               "pfpn_reduce_jobs_norm")
    return nb.value


def value_head_fwd(h, w, bias, out=None):
    lib = _lib.load()
    B, H = h.shape
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=h.device)
    _lib.check(lib.pfpn_value_head_fwd(_f32(h, B, H), _f32(w), _f32(bias), B, H, _f32(out, B), _stream()),
               "pfpn_value_head_fwd")
    return out


def value_head_bwd(h, w, dv, dh, dw, db, relu6=True):
    lib = _lib.load()
    B, H = h.shape
    ws, wsn = _learn_ws(B, H, h.device)
    _lib.check(lib.pfpn_value_head_bwd(_f32(h, B, H), _f32(w), _f32(dv, B), B, H, int(relu6), _f32(dh, B, H),
                                       _f32(dw), _f32(db), ws, wsn, _stream()), "pfpn_value_head_bwd")
    return dh, dw, db


def clip_by_global_norm(grads_flat, clip_norm, norm_scale, apply=True):
    lib = _lib.load()
    n = grads_flat.numel()
    ws, wsn = _learn_ws(1, 1, grads_flat.device)
    _lib.check(lib.pfpn_clip_by_global_norm(_f32(grads_flat), n, float(clip_norm or 0.0), int(apply),
                                            _f32(norm_scale, 2), ws, wsn, _stream()), "pfpn_clip_by_global_norm")
    return norm_scale


def tf_adam(params, grads, m, v, beta_pow, lr, beta1=0.9, beta2=0.999, eps=1e-8, advance=True, norm_scale=None,
            grad_mul=1.0, step_counter=None):
    lib = _lib.load()
    n = params.numel()
    _lib.check(lib.pfpn_tf_adam(_f32(params), _f32(grads), _f32(m), _f32(v), n, float(lr), float(beta1), float(beta2),
                                float(eps), _f32(beta_pow, 2), int(advance),
                                _f32(norm_scale, 2) if norm_scale is not None else None, float(grad_mul),
                                _f32(step_counter, 1) if step_counter is not None else None, _stream()),
               "pfpn_tf_adam")


def _soft_args(soft):
    """soft = (target_flat, off, len, tau) or None -> the four trailing C arguments"""
    if soft is None:
        return None, 0, 0, 0.0
    t, off, ln, tau = soft
    return _f32(t), int(off), int(ln), float(tau)


class StatePrologue(ctypes.Structure):
    """PfpnStatePrologue (include/pfpn_hip.h): the next step's first launch as a rider of the optimiser launch"""
    _fields_ = [("s", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("std", ctypes.c_void_p), ("B", ctypes.c_int),
                ("S", ctypes.c_int), ("clip", ctypes.c_float), ("x", ctypes.c_void_p), ("adv", ctypes.c_void_p),
                ("normalize_adv", ctypes.c_int), ("adv_stats", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
                ("workspace_bytes", ctypes.c_size_t), ("s2", ctypes.c_void_p), ("x2", ctypes.c_void_p),
                ("noise", ctypes.c_void_p), ("noise_n", ctypes.c_size_t), ("noise_seed", ctypes.c_uint64),
                ("noise_epoch", ctypes.c_void_p)]
    _key = None          # set by the network: which prologue call this rider stands for (None: no prologue part)


def state_prologue_rider(s, mean, std, clip, out_x, adv, normalize_adv, adv_stats, ws, s2=None, out_x2=None):
    """the arguments of state_prologue(..., want_moments=True, ws=ws[, s2, out_x2]) as a rider for clip_adam(rider=...)"""
    B, S = s.shape
    r = StatePrologue(s=_f32(s, B, S), mean=_f32(mean, S), std=_f32(std, S), B=B, S=S, clip=float(clip or 0.0),
                      x=_f32(out_x, B, S), adv=_f32(adv, B) if adv_stats is not None else None,
                      normalize_adv=int(normalize_adv), adv_stats=_f32(adv_stats, 4) if adv_stats is not None else None,
                      workspace=ws.data_ptr(), workspace_bytes=ws.numel(),
                      s2=_f32(s2, B, S) if s2 is not None else None, x2=_f32(out_x2, B, S) if s2 is not None else None)
    r._keep = (s, mean, std, out_x, adv, adv_stats, ws, s2, out_x2)
    return r


def noise_rider(rider, buf, seed, epoch):
    """add the SAC step's noise draw (sac_noise_fill_(buf, seed, epoch=epoch)) to `rider` (None: a rider of the noise alone)"""
    if rider is None:
        rider = StatePrologue()
        rider._keep = ()
    if buf.dim() != 3 or buf.shape[0] != 4 or not buf.is_contiguous() or epoch.dtype != torch.int64:
        raise ValueError("noise buffer [4, B, A*N] contiguous and an int64 device counter expected")
    rider.noise, rider.noise_n = _f32(buf), buf.shape[1] * buf.shape[2]
    rider.noise_seed, rider.noise_epoch = int(seed) & 0xFFFFFFFFFFFFFFFF, epoch.data_ptr()
    rider._keep = tuple(rider._keep) + (buf, epoch)
    return rider


def clip_adam(params, grads, m, v, group_off, group_len, group_lr, beta_pow, clip_norm, norm_scale, beta1=0.9,
              beta2=0.999, eps=1e-8, step_counter=None, prenorm=None, soft=None, rider=None):
    """One-replica optimiser step in two launches (global-norm clip folded into TF-Adam).  group_* are ctypes
    arrays (host); beta_pow is [G,4] (see include/pfpn_hip.h).
    prenorm = (partial, count) from reduce_jobs(norm=...): the norm's partial sums exist and the powers are rotated
    already -- ONE launch.  rider (with prenorm): a StatePrologue -- the NEXT learner step's prologue (and, SAC, its noise
    draw) runs as extra blocks of this launch (captured steps)."""
    lib = _lib.load()
    n = params.numel()
    G = len(group_lr)
    if rider is not None:
        if prenorm is None:
            raise ValueError("the prologue rider needs the one-launch optimiser (prenorm)")
        _lib.check(lib.pfpn_clip_adam_prenorm_prologue(_f32(params), _f32(grads), _f32(m), _f32(v), n, G,
                                                       ctypes.addressof(group_off), ctypes.addressof(group_len),
                                                       ctypes.addressof(group_lr), float(beta1), float(beta2), float(eps),
                                                       _f32(beta_pow, G, 4), float(clip_norm), _f32(norm_scale, 2),
                                                       _f32(step_counter, 1) if step_counter is not None else None,
                                                       _f32(prenorm[0]), int(prenorm[1]), ctypes.addressof(rider),
                                                       *_soft_args(soft), _stream()),
                   "pfpn_clip_adam_prenorm_prologue")
        return
    if prenorm is not None:
        _lib.check(lib.pfpn_clip_adam_prenorm(_f32(params), _f32(grads), _f32(m), _f32(v), n, G,
                                              ctypes.addressof(group_off), ctypes.addressof(group_len),
                                              ctypes.addressof(group_lr), float(beta1), float(beta2), float(eps),
                                              _f32(beta_pow, G, 4), float(clip_norm), _f32(norm_scale, 2),
                                              _f32(step_counter, 1) if step_counter is not None else None,
                                              _f32(prenorm[0]), int(prenorm[1]), *_soft_args(soft), _stream()),
                   "pfpn_clip_adam_prenorm")
        return
    if soft is not None:
        raise ValueError("the soft target update rides on the one-launch optimiser (prenorm) only")
    ws, wsn = _learn_ws(1, 1, params.device)
    _lib.check(lib.pfpn_clip_adam(_f32(params), _f32(grads), _f32(m), _f32(v), n, G, ctypes.addressof(group_off),
                                  ctypes.addressof(group_len), ctypes.addressof(group_lr), float(beta1), float(beta2),
                                  float(eps), _f32(beta_pow, G, 4), float(clip_norm), _f32(norm_scale, 2),
                                  _f32(step_counter, 1) if step_counter is not None else None, ws, wsn, _stream()),
               "pfpn_clip_adam")


def clip_scale_inplace(grads_flat, clip_norm, norm_scale, beta_pow=None, beta1=0.9, beta2=0.999, prenorm=None):
    """per-replica global-norm clip applied in place (2 launches); rotates the [G,4] Adam powers when given.
    prenorm = (partial, count) from reduce_jobs(norm=...): ONE launch (the powers were rotated there)."""
    lib = _lib.load()
    if prenorm is not None:
        _lib.check(lib.pfpn_clip_scale_inplace_prenorm(_f32(grads_flat), grads_flat.numel(), float(clip_norm),
                                                       _f32(norm_scale, 2), _f32(prenorm[0]), int(prenorm[1]), _stream()),
                   "pfpn_clip_scale_inplace_prenorm")
        return
    ws, wsn = _learn_ws(1, 1, grads_flat.device)
    G = beta_pow.shape[0] if beta_pow is not None else 0
    _lib.check(lib.pfpn_clip_scale_inplace(_f32(grads_flat), grads_flat.numel(), float(clip_norm), _f32(norm_scale, 2),
                                           _f32(beta_pow, G, 4) if beta_pow is not None else None, G, float(beta1),
                                           float(beta2), ws, wsn, _stream()), "pfpn_clip_scale_inplace")


def clip_scale_only(grads_flat, clip_norm, norm_scale, beta_pow=None, beta1=0.9, beta2=0.999, prenorm=None):
    """norm_scale <- {global norm, clip factor} WITHOUT touching the gradients (sharded exchange: the owner of a bucket
    slice applies every replica's factor itself).  prenorm = (partial, count) from reduce_jobs(norm=...): one launch, the
    powers were rotated there; else two launches over the gradients, rotating the [G,4] Adam powers when given."""
    lib = _lib.load()
    if prenorm is not None:
        _lib.check(lib.pfpn_clip_scale_only(None, 0, float(clip_norm), _f32(norm_scale, 2), _f32(prenorm[0]),
                                            int(prenorm[1]), None, 0, 0.0, 0.0, None, 0, _stream()), "pfpn_clip_scale_only")
        return
    ws, wsn = _learn_ws(1, 1, grads_flat.device)
    G = beta_pow.shape[0] if beta_pow is not None else 0
    _lib.check(lib.pfpn_clip_scale_only(_f32(grads_flat), grads_flat.numel(), float(clip_norm), _f32(norm_scale, 2), None, 0,
                                        _f32(beta_pow, G, 4) if beta_pow is not None else None, G, float(beta1),
                                        float(beta2), ws, wsn, _stream()), "pfpn_clip_scale_only")


def shard_adam(params_x, m, v, recv, scales, world, shard_lo, shard_len, n_params, n_bucket, group_off, group_len, group_lr,
               beta_pow, beta1=0.9, beta2=0.999, eps=1e-8, step_counter=None):
    """the owner's part of the sharded exchange (pfpn_shard_adam): mean of the replicas' clipped gradients on the bucket
    slice [shard_lo, shard_lo + shard_len), TF-Adam there, statistics slice -> replica mean."""
    lib = _lib.load()
    if recv.numel() != world * shard_len or scales.numel() != world or params_x.numel() < shard_lo + shard_len or \
            shard_lo + shard_len > world * shard_len or min(m.numel(), v.numel()) < n_params or \
            params_x.numel() < n_bucket:
        raise ValueError("sharded exchange buffers do not match the shard geometry")
    G = len(group_off)
    _lib.check(lib.pfpn_shard_adam(_f32(params_x), _f32(m), _f32(v), _f32(recv), _f32(scales, world), int(world),
                                   int(shard_lo), int(shard_len), int(n_params), int(n_bucket), G,
                                   ctypes.addressof(group_off), ctypes.addressof(group_len), ctypes.addressof(group_lr),
                                   float(beta1), float(beta2), float(eps), _f32(beta_pow, G, 4),
                                   _f32(step_counter, 1) if step_counter is not None else None, _stream()),
               "pfpn_shard_adam")


def adam_groups(params, grads, m, v, group_off, group_len, group_lr, beta_pow, grad_mul=1.0, beta1=0.9, beta2=0.999,
                eps=1e-8, step_counter=None, scale=None, scale_mul=1.0, soft=None):
    """TF-Adam of every parameter group in one launch, on the powers pfpn_clip_scale_inplace / pfpn_clip_adam rotated.
    `scale` (a float tensor): multiplied by `scale_mul` in the same launch (bucket-tail statistics -> replica means)."""
    lib = _lib.load()
    G = len(group_lr)
    _lib.check(lib.pfpn_adam_groups(_f32(params), _f32(grads), _f32(m), _f32(v), params.numel(), G,
                                    ctypes.addressof(group_off), ctypes.addressof(group_len), ctypes.addressof(group_lr),
                                    float(beta1), float(beta2), float(eps), _f32(beta_pow, G, 4), float(grad_mul),
                                    _f32(step_counter, 1) if step_counter is not None else None,
                                    _f32(scale) if scale is not None else None,
                                    scale.numel() if scale is not None else 0, float(scale_mul), *_soft_args(soft),
                                    _stream()), "pfpn_adam_groups")


def gather_columns(srcs, dsts, rows):
    """dsts[k][b] = srcs[k][rows[b]] for up to 8 float32 columns in one launch (a training round's minibatch slicing
    on the device); srcs[k] [n, ...], dsts[k] [B, ...] with equal trailing shapes, rows [B] int64."""
    lib = _lib.load()
    K, B, n = len(srcs), rows.shape[0], srcs[0].shape[0]
    if not (1 <= K <= 8 and len(dsts) == K):
        raise ValueError("1..8 columns")
    if not (rows.is_cuda and rows.dtype == torch.int64 and rows.is_contiguous() and rows.dim() == 1):
        raise ValueError("rows must be a contiguous int64 CUDA vector")
    widths = []
    for s, d in zip(srcs, dsts):
        w = s[0].numel() if s.dim() > 1 else 1
        _f32(s, name="source column")
        _f32(d, name="destination column")
        if s.shape[0] != n or d.shape[0] != B or d.numel() != B * w:
            raise ValueError("column shapes do not match: %s -> %s" % (tuple(s.shape), tuple(d.shape)))
        widths.append(w)
    ps = (ctypes.c_void_p * K)(*[s.data_ptr() for s in srcs])
    pd = (ctypes.c_void_p * K)(*[d.data_ptr() for d in dsts])
    pw = (ctypes.c_int * K)(*widths)
    _lib.check(lib.pfpn_gather_columns(ctypes.cast(ps, ctypes.c_void_p), ctypes.cast(pd, ctypes.c_void_p),
                                       ctypes.cast(pw, ctypes.c_void_p), K, rows.data_ptr(), n, B, _stream()),
               "pfpn_gather_columns")


def soft_update(target, source, tau):
    lib = _lib.load()
    _lib.check(lib.pfpn_soft_update(_f32(target), _f32(source), target.numel(), float(tau), _stream()),
               "pfpn_soft_update")


def _f32_rows(t, rows, cols, name="tensor"):
    """[rows, cols] float32 CUDA view whose rows are contiguous (any row stride >= cols) -> (pointer, row stride)"""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == (rows, cols)
            and t.stride(1) == 1 and t.stride(0) >= cols):
        raise ValueError("%s must be a [%d, %d] float32 CUDA view with contiguous rows" % (name, rows, cols))
    return t.data_ptr(), t.stride(0)


def sac_loss(q_pi, q_run, q_tgt, logp, logp_t, reward, not_terminal, log_alpha, gamma, value_coef, target_entropy,
             out, noise_epoch=None):
    """out: dict(dq_pi [2,B], dq_run [2,B], dlogp [B], dlog_alpha [1], loss3 [3]).  q_pi, q_run, dq_pi, dq_run may be
    row-strided views with one common row stride (the halves of [2, 2B] buffers)."""
    lib = _lib.load()
    B = logp.shape[0]
    ptrs = [_f32_rows(t, 2, B) for t in (q_pi, q_run, out["dq_pi"], out["dq_run"])]
    ld = ptrs[0][1]
    if any(p[1] != ld for p in ptrs):
        raise ValueError("q_pi, q_run, dq_pi, dq_run must share one row stride")
    _lib.check(lib.pfpn_sac_loss(ptrs[0][0], ptrs[1][0], _f32(q_tgt, 2, B), _f32(logp, B),
                                 _f32(logp_t, B), _f32(reward, B), _f32(not_terminal, B), _f32(log_alpha),
                                 B, float(gamma), float(value_coef), float(target_entropy),
                                 ptrs[2][0], ptrs[3][0], _f32(out["dlogp"], B),
                                 _f32(out["dlog_alpha"]), _f32(out["loss3"], 3), int(ld),
                                 noise_epoch.data_ptr() if noise_epoch is not None else None, _stream()), "pfpn_sac_loss")
    return out


def sac_q_inputs(x, x_next, a_pi, a_hist, a_next, xa_pi, xa_hist, xa_next):
    """the three concat([state, action]) critic inputs of a SAC step, one launch."""
    lib = _lib.load()
    B, S = x.shape
    A = a_pi.shape[1]
    _lib.check(lib.pfpn_sac_q_inputs(_f32(x, B, S), _f32(x_next, B, S), _f32(a_pi, B, A), _f32(a_hist, B, A),
                                     _f32(a_next, B, A), B, S, A, _f32(xa_pi, B, S + A), _f32(xa_hist, B, S + A),
                                     _f32(xa_next, B, S + A), _stream()), "pfpn_sac_q_inputs")


def sac_actor_supported(A, N):
    """whether the two-launch actor side of the SAC step (sac_actor_fwd / sac_actor_bwd) covers this particle grid"""
    return bool(_lib.load().pfpn_sac_actor_supported(int(A), int(N)))


def sac_actor_fwd(logits, loc, logstd, gumbel, eps, x=None, a_hist=None, xq_both=None, xq_next=None):
    """reparameterised sample + its log-prob (tanh mode) over the R rows of logits, and -- with x [R, S], a_hist [B, A],
    xq_both [2B, S + A], xq_next [R - B, S + A] -- the three critic inputs, in ONE launch.  -> a, s_, k [R, A], logp [R]."""
    lib = _lib.load()
    A, N = loc.shape
    R = logits.shape[0]
    dev = logits.device
    k = torch.empty(R, A, dtype=torch.int32, device=dev)
    a = torch.empty(R, A, dtype=torch.float32, device=dev)
    s_ = torch.empty(R, A, dtype=torch.float32, device=dev)
    logp = torch.empty(R, dtype=torch.float32, device=dev)
    B = S = 0
    px = pa = pq = pn = None
    if xq_both is not None:
        B, S = a_hist.shape[0], x.shape[1]
        px, pa = _f32(x, R, S), _f32(a_hist, B, A)
        pq, pn = _f32(xq_both, 2 * B, S + A), _f32(xq_next, R - B, S + A)
    _lib.check(lib.pfpn_sac_actor_fwd(_f32(logits.view(R, A * N), R, A * N), _f32(loc, A, N), _f32(logstd, A, N),
                                      _f32(gumbel.view(R, A * N), R, A * N), _f32(eps.view(R, A * N), R, A * N), R, A, N,
                                      k.data_ptr(), a.data_ptr(), s_.data_ptr(), logp.data_ptr(), px, pa, B, S, pq, pn,
                                      _stream()), "pfpn_sac_actor_fwd")
    return a, s_, k, logp


def sac_actor_workspace(B, A, N, device):
    return torch.empty(int(_lib.load().pfpn_sac_actor_workspace_bytes(B, A, N)), dtype=torch.uint8, device=device)


def sac_actor_bwd_slabs(B, A, N):
    """(nslab, stride) of the {dloc | dlogstd | d head bias} partial sums sac_actor_bwd leaves in its workspace"""
    n, st = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.load().pfpn_sac_actor_bwd_slabs(B, A, N, ctypes.addressof(n), ctypes.addressof(st)),
               "pfpn_sac_actor_bwd_slabs")
    return n.value, st.value


def sac_actor_bwd(logits, loc, logstd, gumbel, eps, s_, dlogp, da, dlogits, ws):
    """log-prob backward (at the sample s_, upstream dlogp [B]) + sample backward (upstream da [B, A] of tanh(s_); the
    upstream of s_ is the log-prob's own dx) in ONE launch: dlogits [B, A*N] written, the particle-grid and head-bias
    partial sums left in `ws` (sac_actor_workspace; sac_actor_bwd_slabs)."""
    lib = _lib.load()
    A, N = loc.shape
    B = logits.shape[0]
    _lib.check(lib.pfpn_sac_actor_bwd(_f32(logits.view(B, A * N), B, A * N), _f32(loc, A, N), _f32(logstd, A, N),
                                      _f32(gumbel.view(B, A * N), B, A * N), _f32(eps.view(B, A * N), B, A * N),
                                      _f32(s_, B, A), _f32(dlogp, B), _f32(da, B, A), B, A, N,
                                      _f32(dlogits.view(B, A * N), B, A * N), ws.data_ptr(), ws.numel(), _stream()),
               "pfpn_sac_actor_bwd")
    return dlogits


STREAM_PAIR_MIN_BYTES = 256 << 20          # = the wide backward's streaming regime (dlogits beyond the infinity cache)


def stream_pair(rows, cols, device):
    """Two [rows, cols] float32 buffers for a kernel that streams the first in while it streams the second out (logits ->
    dlogits of the wide mixture backward).  Where the second sits RELATIVE to the first decides how the two streams meet
    in the memory system: measured on the 8192 x 128 x 256 shape, one allocation with the second buffer at
    (a multiple of 256 MiB) + 4 KiB behind the first runs the backward in 379-381 us in every process and on every box
    tried; + 16 KiB, + 16 MiB or + 64 MiB: 445-485 us; two independent allocations land anywhere in between, a different
    place in every process (tools/skew_probe.py, profiles/r03_stream_skew.txt).  Below STREAM_PAIR_MIN_BYTES: two plain
    allocations."""
    n = int(rows) * int(cols)
    if n * 4 < STREAM_PAIR_MIN_BYTES:
        return (torch.empty(rows, cols, dtype=torch.float32, device=device),
                torch.empty(rows, cols, dtype=torch.float32, device=device))
    step = 256 << 20
    off = ((n * 4 + step - 1) // step * step + 4096) // 4
    pool = torch.empty(off + n, dtype=torch.float32, device=device)
    return pool[:n].view(rows, cols), pool[off:off + n].view(rows, cols)


def stream_copy(dst, src, unroll=4, nontemporal=False, blocks=2048):
    """the build's own streaming copy (calibration of the HBM roofline fractions on the box at hand)."""
    lib = _lib.load()
    n = src.numel()
    if dst.numel() != (n if unroll else blocks * 256):
        raise ValueError("stream_copy: dst is src-sized (copy) or [blocks * 256] (unroll = 0: read-only stream)")
    _lib.check(lib.pfpn_stream_copy(_f32(dst), _f32(src), n, int(unroll), int(nontemporal), int(blocks), _stream()),
               "pfpn_stream_copy")
    return dst
