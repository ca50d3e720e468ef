"""The ctypes wrappers of the dense MFMA families as they stood in pfpn_amd/kernels.py until round 4 (moved out of the
product: see README.md in this directory).  Not importable on its own: it was part of pfpn_amd/kernels.py."""
# --------------------------------------------------------------------------------------- #
# dense layers on the hand-written fp32 MFMA GEMM (fused epilogues)                         #
# --------------------------------------------------------------------------------------- #
def _dense_ws(G, K, N, device):
    lib = _lib.load()
    buf = workspace(lib.pfpn_dense_workspace_bytes(G, K, N), device)
    return buf.data_ptr(), buf.numel()


def dense_fwd(x, w, bias, out, relu6=True):
    """x [rows, ldx] shared by all groups (K = w.shape[1] <= ldx) or [G, rows, K]; w [G,K,N]; bias [G,N]; out [G,rows,N]."""
    lib = _lib.load()
    G, K, N = w.shape
    if x.dim() == 2:
        rows, ldx, gstride = x.shape[0], x.stride(0), 0
    else:
        rows, ldx, gstride = x.shape[1], x.stride(1), x.stride(0)
    _lib.check(lib.pfpn_dense_fwd(x.data_ptr(), gstride, ldx, _f32(w, G, K, N), _f32(bias, G, N), G, rows, K, N,
                                  int(relu6), _f32(out, G, rows, N), _stream()), "pfpn_dense_fwd")
    return out


def dense_dgrad(dy, w, hin, dx):
    """dy [G,rows,N], w [G,K,N], hin [G,rows,K] or None, dx [G,rows,K]."""
    lib = _lib.load()
    G, K, N = w.shape
    rows = dy.shape[1]
    _lib.check(lib.pfpn_dense_dgrad(_f32(dy, G, rows, N), _f32(w, G, K, N),
                                    _f32(hin, G, rows, K) if hin is not None else None, G, rows, K, N,
                                    _f32(dx, G, rows, K), _stream()), "pfpn_dense_dgrad")
    return dx


def dense_wgrad(x, dy, dw, db):
    """x [rows, ldx] shared or [G,rows,K]; dy [G,rows,N]; dw [G,K,N]; db [G,N] or None."""
    lib = _lib.load()
    G, K, N = dw.shape
    if x.dim() == 2:
        rows, ldx, gstride = x.shape[0], x.stride(0), 0
    else:
        rows, ldx, gstride = x.shape[1], x.stride(1), x.stride(0)
    ws, wsn = _dense_ws(G, K, N, dy.device)
    _lib.check(lib.pfpn_dense_wgrad(x.data_ptr(), gstride, ldx, _f32(dy, G, rows, N), G, rows, K, N, _f32(dw, G, K, N),
                                    _f32(db, G, N) if db is not None else None, ws, wsn, _stream()), "pfpn_dense_wgrad")
    return dw, db


# ---- small-row dense layers: one launch per layer and direction (csrc/rowdense_kernels.hip) ---------------------------
class DenseJob(ctypes.Structure):
    """mirror of PfpnDenseJob (include/pfpn_hip.h)"""
    _fields_ = [("kind", ctypes.c_int), ("flags", ctypes.c_int), ("cfg", ctypes.c_int),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("batch", ctypes.c_int),
                ("lda", ctypes.c_int), ("ldb", ctypes.c_int), ("ldc", ctypes.c_int), ("ldh", ctypes.c_int),
                ("hgroup", ctypes.c_int),
                ("sA", ctypes.c_longlong), ("sB", ctypes.c_longlong), ("sC", ctypes.c_longlong),
                ("sH", ctypes.c_longlong), ("sBias", ctypes.c_longlong), ("sAux", ctypes.c_longlong),
                ("sHw", ctypes.c_longlong),
                ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("H", ctypes.c_void_p), ("aux", ctypes.c_void_p), ("hw", ctypes.c_void_p),
                ("p0", ctypes.c_float), ("p1", ctypes.c_float)]


DENSE_NN, DENSE_NT, DENSE_TN, DENSE_HEAD_FINAL, DENSE_RANK1, DENSE_HEAD_WGRAD = 0, 1, 2, 8, 9, 10
DENSE_SLABSUM, DENSE_SQSUM, DENSE_PPO_LOSS, DENSE_NORM_EMA = 11, 12, 13, 14
DENSE_BIAS, DENSE_RELU6, DENSE_MASK, DENSE_COLSUM, DENSE_HEADPART, DENSE_NORM = 1, 2, 4, 8, 16, 32
DENSE_MAX_JOBS = 16


def _mat(t, G, rows, cols, name):
    """(ptr, ld, group stride) of a float32 CUDA matrix view [rows, cols] (shared by the G groups) or [G, rows, cols]
    whose rows are contiguous; shapes are checked here so the kernel never reads outside the view."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise ValueError("%s must be a float32 CUDA tensor" % name)
    if t.dim() == 2:
        if tuple(t.shape) != (rows, cols) or (cols > 1 and t.stride(1) != 1):
            raise ValueError("%s: shape %s / strides %s, expected [%d, %d] with contiguous rows"
                             % (name, tuple(t.shape), t.stride(), rows, cols))
        return t.data_ptr(), t.stride(0), 0
    if t.dim() != 3 or tuple(t.shape) != (G, rows, cols) or (cols > 1 and t.stride(2) != 1):
        raise ValueError("%s: shape %s / strides %s, expected [%d, %d, %d] with contiguous rows"
                         % (name, tuple(t.shape), t.stride(), G, rows, cols))
    return t.data_ptr(), t.stride(1), (t.stride(0) if G > 1 else 0)


def _vec(t, G, n, name):
    """(ptr, group stride) of a float32 CUDA vector [n] (shared) or [G, n] with contiguous elements."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise ValueError("%s must be a float32 CUDA tensor" % name)
    if t.dim() == 1:
        if t.shape[0] != n or (n > 1 and t.stride(0) != 1):
            raise ValueError("%s: shape %s, expected [%d]" % (name, tuple(t.shape), n))
        return t.data_ptr(), 0
    if t.dim() != 2 or tuple(t.shape) != (G, n) or (n > 1 and t.stride(1) != 1):
        raise ValueError("%s: shape %s, expected [%d, %d]" % (name, tuple(t.shape), G, n))
    return t.data_ptr(), (t.stride(0) if G > 1 else 0)


def _groups(*ts):
    return max([t.shape[0] for t in ts if t is not None and t.dim() == 3] + [1])


def dense_job_nn(x, w, out, bias=None, relu6=False, head=None, cfg=-1):
    """out[g] = act(x[g] w[g] + bias[g]): x [M,K] (shared) or [G,M,K], w [K,N] or [G,K,N], out [G,M,N] / [M,N].
    head = (hgroup, hw [N] or [G,N], hpart [M, ceil(N/32)] or [G, M, ceil(N/32)]): the per-32-column partial dot products
    of the output with a scalar head's weights on group `hgroup` (-1: every group) -- finished by dense_job_head_final."""
    G = _groups(x, w, out)
    M, K, N = x.shape[-2], x.shape[-1], w.shape[-1]
    j = DenseJob(kind=DENSE_NN, cfg=int(cfg), M=M, N=N, K=K, batch=G, hgroup=0)
    j.A, j.lda, j.sA = _mat(x, G, M, K, "x")
    j.B, j.ldb, j.sB = _mat(w, G, K, N, "w")
    j.C, j.ldc, j.sC = _mat(out, G, M, N, "out")
    if bias is not None:
        j.bias, j.sBias = _vec(bias, G, N, "bias")
        j.flags |= DENSE_BIAS
    if relu6:
        j.flags |= DENSE_RELU6
    if head is not None:
        hg, hw, hpart = head
        parts = (N + 31) // 32
        if not -1 <= hg < G:
            raise ValueError("head group outside the batch")
        j.hgroup = int(hg)
        j.hw, j.sHw = _vec(hw, G, N, "head weight")
        j.aux, _, j.sAux = _mat(hpart, G, M, parts, "head partials")
        if not hpart.is_contiguous():
            raise ValueError("head partials must be contiguous")
        j.flags |= DENSE_HEADPART
    return j


def dense_job_nt(a, bt, out, mask=None, cfg=-1):
    """out[g] = (a[g] bt[g]^T) * [0 < mask[g] < 6]: a [G,M,K], bt [G,N,K], out [G,M,N], mask like out (or None)."""
    G = _groups(a, bt, out)
    M, K, N = a.shape[-2], a.shape[-1], bt.shape[-2]
    j = DenseJob(kind=DENSE_NT, cfg=int(cfg), M=M, N=N, K=K, batch=G)
    j.A, j.lda, j.sA = _mat(a, G, M, K, "a")
    j.B, j.ldb, j.sB = _mat(bt, G, N, K, "bt")
    j.C, j.ldc, j.sC = _mat(out, G, M, N, "out")
    if mask is not None:
        j.H, j.ldh, j.sH = _mat(mask, G, M, N, "mask")
        j.flags |= DENSE_MASK
    return j


def dense_job_tn(at, b, out, colsum=None, cfg=-1):
    """out[g] = at[g]^T b[g], colsum[g] = column sums of b[g]: at [G,K,M] (or [K,M] shared), b [G,K,N], out [G,M,N],
    colsum [G,N] / [N]."""
    G = _groups(at, b, out)
    K, M, N = at.shape[-2], at.shape[-1], b.shape[-1]
    j = DenseJob(kind=DENSE_TN, cfg=int(cfg), M=M, N=N, K=K, batch=G)
    j.A, j.lda, j.sA = _mat(at, G, K, M, "at")
    j.B, j.ldb, j.sB = _mat(b, G, K, N, "b")
    j.C, j.ldc, j.sC = _mat(out, G, M, N, "out")
    if colsum is not None:
        j.aux, j.sAux = _vec(colsum, G, N, "colsum")
        j.flags |= DENSE_COLSUM
    return j


def dense_job_head_final(part, bias, out):
    """out[g][b] = bias[g] + sum_t part[g][b][t]: part [M,P] / [G,M,P] contiguous, bias [1] / [G,1] (or [G]), out [M] / [G,M]."""
    G = part.shape[0] if part.dim() == 3 else 1
    M, P = part.shape[-2], part.shape[-1]
    if not part.is_contiguous() or not out.is_contiguous():
        raise ValueError("head partials / outputs must be contiguous")
    j = DenseJob(kind=DENSE_HEAD_FINAL, M=M, N=1, K=P, batch=G)
    j.A, _, j.sA = _mat(part, G, M, P, "part")
    b2 = bias.view(G, -1) if G > 1 else bias.view(-1)
    j.bias, j.sBias = _vec(b2[..., :1], G, 1, "head bias")
    if G > 1:
        j.sBias = b2.stride(0)
    j.C, j.sC = _vec(out, G, M, "out")
    return j


def dense_job_rank1(dv, w, h, out):
    """out[g][b][n] = dv[g][b] * w[g][n] * [0 < h[g][b][n] < 6]: dv [M] / [G,M], w [N] / [G,N], h and out [M,N] / [G,M,N]."""
    G = _groups(h, out)
    M, N = h.shape[-2], h.shape[-1]
    j = DenseJob(kind=DENSE_RANK1, M=M, N=N, K=1, batch=G)
    j.A, j.sA = _vec(dv, G, M, "dv")
    j.B, j.sB = _vec(w, G, N, "w")
    j.H, j.ldh, j.sH = _mat(h, G, M, N, "h")
    j.C, j.ldc, j.sC = _mat(out, G, M, N, "out")
    return j


def dense_job_head_wgrad(dv, h, dw, db):
    """dw[g][n] = sum_b dv[g][b] h[g][b][n], db[g] = sum_b dv[g][b]: dv [M] / [G,M], h [M,N] / [G,M,N], dw [N] / [G,N],
    db [1] / [G,1]."""
    G = _groups(h)
    M, N = h.shape[-2], h.shape[-1]
    j = DenseJob(kind=DENSE_HEAD_WGRAD, M=M, N=N, K=1, batch=G)
    j.A, j.sA = _vec(dv, G, M, "dv")
    j.H, j.ldh, j.sH = _mat(h, G, M, N, "h")
    j.C, j.sC = _vec(dw, G, N, "dw")
    d2 = db.view(G, -1) if G > 1 else db.view(-1)
    j.aux, j.sAux = _vec(d2[..., :1], G, 1, "db")
    if G > 1:
        j.sAux = d2.stride(0)
    return j


def dense_job_of_reduce(job):
    """a deferred reduction (ReduceJob of reduce_jobs()) as a job of the small-row family's LAST dense launch, which then
    stands in for the separate reduction launch: column sums of partial slabs, the PPO loss columns, the state-normaliser
    EMA, square sums of finished gradient segments.  Pointers were validated when the ReduceJob was made."""
    if job.kind == JOB_COLSUM:
        j = DenseJob(kind=DENSE_SLABSUM, M=1, N=job.n, K=job.nslab, batch=1, lda=job.stride, ldb=0)
        j.A, j.C = job.src, job.out
        if job.in0:
            j.B, j.M, j.ldb = job.in0, int(job.p0), int(job.p1)
        if job.norm:
            j.flags |= DENSE_NORM
        return j
    if job.kind == JOB_SQSUM:
        j = DenseJob(kind=DENSE_SQSUM, M=1, N=job.n, K=job.nslab, batch=1, flags=DENSE_NORM)
        j.A = job.src
        return j
    if job.kind == JOB_PPO_LOSS:
        j = DenseJob(kind=DENSE_PPO_LOSS, M=1, N=2, K=job.nslab, batch=1, lda=job.stride, p0=job.p0, p1=job.p1)
        j.A, j.C = job.src, job.out
        return j
    if job.kind == JOB_NORM_EMA:
        j = DenseJob(kind=DENSE_NORM_EMA, M=job.stride, N=job.n, K=job.nslab, batch=1)
        j.A, j.C, j.aux, j.bias, j.B, j.H = job.src, job.out, job.out2, job.in0, job.in1, job.in2
        return j
    raise ValueError("reduce job kind %d has no dense-launch form" % job.kind)


def dense_jobs(jobs, norm=None):
    """ONE launch for up to DENSE_MAX_JOBS jobs (products with fused epilogues + scalar-head helpers + reductions).
    norm = dict(partial [>= #blocks] f32, beta_pow [G,4] or None, beta1, beta2): the launch also leaves the global-norm
    partial sums (one per block: squares of the outputs of jobs flagged DENSE_NORM) and rotates the Adam powers; returns
    the number of partials (for clip_adam(prenorm=...))."""
    lib = _lib.load()
    if not 1 <= len(jobs) <= DENSE_MAX_JOBS:
        raise ValueError("1..%d dense jobs per launch" % DENSE_MAX_JOBS)
    arr = (DenseJob * len(jobs))(*jobs)
    if norm is None:
        _lib.check(lib.pfpn_dense_jobs(ctypes.addressof(arr), len(jobs), _stream()), "pfpn_dense_jobs")
        return 0
    nb = ctypes.c_int(0)
    bp = norm.get("beta_pow")
    G = bp.shape[0] if bp is not None else 0
    _lib.check(lib.pfpn_dense_jobs_norm(ctypes.addressof(arr), len(jobs), _f32(norm["partial"]), norm["partial"].numel(),
                                        ctypes.addressof(nb), _f32(bp, G, 4) if bp is not None else None, G,
                                        float(norm.get("beta1", 0.9)), float(norm.get("beta2", 0.999)), _stream()),
               "pfpn_dense_jobs_norm")
    return nb.value
