"""Thin tensor-level wrappers over the C ABI (libpfpn_hip.so).

PyTorch is plumbing here: device memory + the current HIP stream.  Every function validates
shapes/dtypes on the host before launching (a faulting kernel can take the whole node down) and
launches on torch's current stream, so calls can be captured in a torch.cuda.CUDAGraph.
"""
import torch

from . import _lib

_ws = {}


def _stream():
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
    key = (device.index if device.index is not None else torch.cuda.current_device())
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


def logprob_fwd(logits, x, loc, logstd, tanh_mode=False, want_pmix=False):
    """logits [B,A,N] (or [B,A*N]), x [B,A], loc/logstd [A,N] -> logp [B] (, pmix [B,A])."""
    lib = _lib.load()
    A, N = loc.shape
    B = x.shape[0]
    pl = _f32(logits.view(B, A * N), B, A * N, name="logits")
    px = _f32(x, B, A, name="x")
    ploc, pls = _f32(loc, A, N, name="loc"), _f32(logstd, A, N, name="logstd")
    logp = torch.empty(B, dtype=torch.float32, device=x.device)
    pmix = torch.empty(B, A, dtype=torch.float32, device=x.device) if want_pmix else None
    ws, wsn = _mix_ws(B, A, N, x.device)
    _lib.check(lib.pfpn_logprob_fwd(pl, px, ploc, pls, B, A, N, int(tanh_mode), logp.data_ptr(),
                                    pmix.data_ptr() if want_pmix else None, ws, wsn, _stream()),
               "pfpn_logprob_fwd")
    return (logp, pmix) if want_pmix else logp


def logprob_bwd(logits, x, loc, logstd, dlogp, tanh_mode=False, want_dx=False):
    """-> dlogits [B,A,N], dloc [A,N], dlogstd [A,N] (, dx [B,A])."""
    lib = _lib.load()
    A, N = loc.shape
    B = x.shape[0]
    pl = _f32(logits.view(B, A * N), B, A * N, name="logits")
    px = _f32(x, B, A, name="x")
    ploc, pls = _f32(loc, A, N, name="loc"), _f32(logstd, A, N, name="logstd")
    pg = _f32(dlogp, B, name="dlogp")
    dlogits = torch.empty(B, A, N, dtype=torch.float32, device=x.device)
    dloc = torch.empty(A, N, dtype=torch.float32, device=x.device)
    dlogstd = torch.empty(A, N, dtype=torch.float32, device=x.device)
    dx = torch.empty(B, A, dtype=torch.float32, device=x.device) if want_dx else None
    ws, wsn = _mix_ws(B, A, N, x.device)
    _lib.check(lib.pfpn_logprob_bwd(pl, px, ploc, pls, pg, B, A, N, int(tanh_mode), dlogits.data_ptr(),
                                    dloc.data_ptr(), dlogstd.data_ptr(), dx.data_ptr() if want_dx else None,
                                    ws, wsn, _stream()), "pfpn_logprob_bwd")
    return (dlogits, dloc, dlogstd, dx) if want_dx else (dlogits, dloc, dlogstd)


def adv_stats(adv, normalize=True, out=None):
    lib = _lib.load()
    B = adv.shape[0]
    pa = _f32(adv, B, name="advantage")
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=adv.device)
    _lib.check(lib.pfpn_adv_stats(pa, B, int(normalize), out.data_ptr(), _stream()), "pfpn_adv_stats")
    return out


def ppo_mixture_fwd_bwd(logits, action, loc, logstd, logp_old, adv, value, value_old, stats,
                        clip_eps=0.2, value_coef=0.5, out=None):
    """Fused learner kernel.  Returns dict(logp, dlogits, dloc, dlogstd, dvalue, loss3)."""
    lib = _lib.load()
    A, N = loc.shape
    B = action.shape[0]
    dev = action.device
    pl = _f32(logits.view(B, A * N), B, A * N, name="logits")
    pa = _f32(action, B, A, name="action")
    ploc, pls = _f32(loc, A, N, name="loc"), _f32(logstd, A, N, name="logstd")
    ptrs = [_f32(logp_old, B, name="log_prob"), _f32(adv, B, name="advantage"), _f32(value, B, name="value"),
            _f32(value_old, B, name="value_old"), _f32(stats, 2, name="adv_stats")]
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
    ws, wsn = _mix_ws(B, A, N, dev)
    _lib.check(lib.pfpn_ppo_mixture_fwd_bwd(pl, pa, ploc, pls, *ptrs, B, A, N, float(clip_eps),
                                            float(value_coef), out["logp"].data_ptr(),
                                            out["dlogits"].data_ptr(), out["dloc"].data_ptr(),
                                            out["dlogstd"].data_ptr(), out["dvalue"].data_ptr(),
                                            out["loss3"].data_ptr(), ws, wsn, _stream()),
               "pfpn_ppo_mixture_fwd_bwd")
    return out
