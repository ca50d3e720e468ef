"""Dense-layer plumbing for the MI355X build: one flat fp32 parameter buffer, GEMMs through
PyTorch-ROCm (fp32 MFMA via hipBLASLt/rocBLAS, `allow_tf32` off), bias/ReLU6 epilogues and their
backward through the HIP kernels of libpfpn_hip.so.

Mirrors reference networks/ops.py (`fc_layer` :82-118, `clip_by_value_with_gradient` :4-9); the
reference's conv/lstm layers are never instantiated by the DeepMimic settings and
are not provided.  No autograd: the backward is written out so every buffer is preallocated, every
gradient lands directly in the flat gradient buffer (one RCCL bucket, one Adam launch), and the whole
step can be captured in a hipGraph.
"""
import math

import torch

from .. import kernels

torch.backends.cuda.matmul.allow_tf32 = False


def truncated_normal_(t, std=0.01, generator=None):
    """tf.truncated_normal_initializer(0, std): resample outside 2 sigma (deepmimic_base.py:7)."""
    with torch.no_grad():
        torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=generator)
    return t


class ParamStore:
    """Named fp32 tensors carved out of ONE flat buffer (+ matching flat grad / Adam m / Adam v).

    Every tensor starts on a 16-byte boundary (sizes padded to 4 floats; padding stays zero) so the
    float4 streaming kernels (clip, Adam, all-reduce bucket) run over the whole buffer in one launch.
    `extra` floats are reserved behind the gradients for tensors that ride in the same all-reduce
    bucket (state-normaliser moments, activation statistics: reference sync_model.py:37-46)."""

    def __init__(self, device):
        self.device = device
        self.specs = []       # (name, shape, offset)
        self.size = 0
        self.flat = self.grad = self.m = self.v = None
        self.extra = 0
        self._views = {}
        self._gviews = {}

    def declare(self, name, shape):
        n = int(math.prod(shape)) if len(shape) else 1
        self.specs.append((name, tuple(shape), self.size))
        self.size += (n + 3) // 4 * 4
        return name

    def allocate(self, extra=0, with_optimizer=True):
        self.extra = (int(extra) + 3) // 4 * 4
        # both buffers are padded to a multiple of 256 floats (zeros): any power-of-two replica count up to 64 then cuts
        # them into equal 16-byte-aligned slices (the sharded exchange: an all-to-all of bucket slices to their owners,
        # an in-place all-gather of the parameter slices), and the parameter buffer has room behind the parameters for
        # the replicated statistics -- the bucket's index space [parameters | statistics | padding] on both sides
        self.npad = (self.size + self.extra + 255) // 256 * 256
        self.flatx = torch.zeros(self.npad, dtype=torch.float32, device=self.device)
        self.flat = self.flatx[:self.size]
        self.flat_tail = self.flatx[self.size:self.size + self.extra]
        self.bucketx = torch.zeros(self.npad, dtype=torch.float32, device=self.device)
        self.bucket = self.bucketx[:self.size + self.extra]
        self.grad = self.bucket[:self.size]
        self.tail = self.bucket[self.size:]
        if with_optimizer:
            # (padded like the parameters: under the sharded exchange a replica keeps only its slice of the Adam slots up
            # to date; `SyncReplicasApplier.gather_optimizer_state` all-gathers them in place before a checkpoint)
            self.mx = torch.zeros(self.npad, dtype=torch.float32, device=self.device)
            self.vx = torch.zeros(self.npad, dtype=torch.float32, device=self.device)
            self.m, self.v = self.mx[:self.size], self.vx[:self.size]
        for name, shape, off in self.specs:
            n = int(math.prod(shape)) if len(shape) else 1
            self._views[name] = self.flat[off:off + n].view(shape)
            self._gviews[name] = self.grad[off:off + n].view(shape)
        return self

    def __getitem__(self, name):
        return self._views[name]

    def views_of(self, flat):
        """named views of ANOTHER flat buffer with this store's layout (Adam m / v, a target network)."""
        class _Views(dict):
            def g(self, name):
                return self[name]
        out = _Views()
        for name, shape, off in self.specs:
            n = int(math.prod(shape)) if len(shape) else 1
            out[name] = flat[off:off + n].view(shape)
        return out

    def g(self, name):
        return self._gviews[name]

    def names(self):
        return [s[0] for s in self.specs]

    def span(self, names):
        """(offset, length) of the contiguous flat range covering `names` (declared adjacently)."""
        offs = {s[0]: (s[2], (int(math.prod(s[1])) + 3) // 4 * 4) for s in self.specs}
        lo = min(offs[n][0] for n in names)
        hi = max(offs[n][0] + offs[n][1] for n in names)
        return lo, hi - lo

    def state_dict(self):
        return {name: self._views[name].detach().clone() for name in self.names()}

    def load_state_dict(self, sd):
        for name in self.names():
            if name in sd:
                self._views[name].copy_(torch.as_tensor(sd[name]).to(self.device).view(self._views[name].shape))


class DenseStack:
    """G parallel MLP trunks with identical layer shapes fed by the SAME input (G=2: the actor and
    critic trunks of the DeepMimic settings, or the twin Q networks of SAC).

    Layer 0 of a multi-layer stack is ONE GEMM for all groups: its weight is stored [in, G*H0] (group g
    owns columns [g*H0, (g+1)*H0)), so x[B,in] @ W gives the activations of every trunk as [B, G*H0] --
    twice the tiles per launch and half the launches of per-trunk GEMMs (measured 42 -> 33 us forward,
    52 -> 37 us weight gradient at 4096 x 197 x 2 x 1024).  Layers i > 0 are one batched GEMM over the
    groups; they read layer 0's output through the strided view [G, B, H0] (batch stride H0, row stride
    G*H0), which hipBLASLt takes as is.

    Reference: build_conv_fc_net / fc_layer (networks/utils.py:17-43, ops.py:82-118):
    y = relu6(x W + b), truncated-normal(0, 0.01) weights, zero bias."""

    def __init__(self, store, prefixes, in_dim, hidden, first_name_index=1):
        self.store = store
        self.G = len(prefixes)
        self.prefixes = prefixes
        self.dims = [in_dim] + list(hidden)
        self.L = len(hidden)
        self.merged0 = self.G > 1 and self.L > 1
        self.wn, self.bn = [], []
        # declare group-major tensors; reference names are views of them
        for i in range(self.L):
            wshape = (self.dims[0], self.G * self.dims[1]) if (i == 0 and self.merged0) else \
                (self.G, self.dims[i], self.dims[i + 1])
            self.wn.append(store.declare("%s|fc%d/weight" % ("+".join(prefixes), i + first_name_index), wshape))
            self.bn.append(store.declare("%s|fc%d/bias" % ("+".join(prefixes), i + first_name_index),
                                         (self.G, self.dims[i + 1])))
        self.first_name_index = first_name_index
        self._bufs = {}
        self._last = None
        self.pinned = set()       # batch sizes a captured graph points at (the network shares its set): never dropped

    MAX_BUFFER_SETS = 64

    def group_view(self, t, layer, g, bias=False):
        """group g's slice of a layer tensor (parameter, gradient or Adam slot) in the reference's shape."""
        if layer == 0 and self.merged0 and not bias:
            H0 = self.dims[1]
            return t[:, g * H0:(g + 1) * H0]
        return t[g]

    def reference_names(self):
        """reference variable name -> (store name, selector: stored tensor -> that variable's view)."""
        out = {}
        for i in range(self.L):
            for g, p in enumerate(self.prefixes):
                out["%s/fc%d/weight" % (p, i + self.first_name_index)] = \
                    (self.wn[i], lambda t, i=i, g=g: self.group_view(t, i, g))
                out["%s/fc%d/bias" % (p, i + self.first_name_index)] = \
                    (self.bn[i], lambda t, i=i, g=g: self.group_view(t, i, g, bias=True))
        return out

    def init_weights(self, generator=None, std=0.01):
        for i in range(self.L):
            truncated_normal_(self.store[self.wn[i]], std, generator)
            self.store[self.bn[i]].zero_()

    def _buffers(self, B, slot):
        key = (B, slot)
        if key in self._bufs:
            return self._bufs[key]
        dev = self.store.device
        G = self.G

        def alloc(i):
            H = self.dims[i + 1]
            if i == 0 and self.merged0:
                flat = torch.empty(B, G * H, dtype=torch.float32, device=dev)
                return flat, flat.view(B, G, H).transpose(0, 1)
            t = torch.empty(G, B, H, dtype=torch.float32, device=dev)
            return t.view(G * B, H), t
        a = [alloc(i) for i in range(self.L)]
        d = [alloc(i) for i in range(self.L)]
        if len(self._bufs) > self.MAX_BUFFER_SETS:
            # ragged batches (episodic V-trace / A2C rounds) would otherwise keep one activation set per size ever seen:
            # drop the oldest set no graph can point at (a step over B rows may run its stacks over 2B rows: SAC)
            for old in list(self._bufs):
                b0 = old[0]
                if isinstance(b0, int) and old != key and b0 not in self.pinned and \
                        not (b0 % 2 == 0 and b0 // 2 in self.pinned) and \
                        not any(v.get("prefix_of") is self._bufs[old] for v in self._bufs.values() if isinstance(v, dict)):
                    del self._bufs[old]
                    break
        self._bufs[key] = {"act2d": [t[0] for t in a], "act": [t[1] for t in a],
                           "dact2d": [t[0] for t in d], "dact": [t[1] for t in d], "x": None, "B": B, "ws": None,
                           "mask": [None] * self.L, "mask_ok": [False] * self.L}
        return self._bufs[key]

    def share_prefix(self, B_big, B_small, slot_big, slot_small):
        """(one group) register the first `B_small` rows of the activation set (B_big, slot_big) as the set
        (B_small, slot_small): after ONE forward over B_big rows a backward can run over the leading rows only (SAC: the
        actor evaluated on [s; s'] in one pass, differentiated through s alone).  Views, no copies; the mask bits of a
        row-major [rows, H] layer are a prefix too (H / 4 a multiple of 64)."""
        if self.G != 1:
            raise ValueError("prefix sharing is built for single-group stacks")
        big = self._bufs[(B_big, slot_big)]
        key = (B_small, slot_small)
        v = self._bufs.get(key)
        if v is None or v.get("prefix_of") is not big:
            v = self._bufs[key] = {
                "act2d": [a[:B_small] for a in big["act2d"]], "act": [a[:, :B_small] for a in big["act"]],
                "dact2d": [a[:B_small] for a in big["dact2d"]], "dact": [a[:, :B_small] for a in big["dact"]],
                "x": None, "B": B_small, "ws": None, "mask": big["mask"], "mask_ok": big["mask_ok"], "prefix_of": big}
        v["x"] = big["x"][:B_small]
        return v

    def _slab_workspaces(self, buf):
        """private partial-sum workspaces (one per layer + one for a folded critic head): their reductions are
        deferred to the step's single `kernels.reduce_jobs` launch, so no two producers may share slabs."""
        if buf["ws"] is None:
            B = buf["B"]
            buf["ws"] = [kernels.learner_workspace(self.G * B, a.shape[1], self.store.device) for a in buf["act2d"]]
            buf["ws_value"] = kernels.learner_workspace(self.G * B, self.G * self.dims[-1], self.store.device)
        return buf["ws"], buf["ws_value"]

    @property
    def act(self):
        return self._last["act"]

    def _group_rows(self, i, B):
        """`group_rows` of the epilogue kernels: layer 0 merged is one [B, G*H] matrix with a [G*H] bias."""
        return 0 if (i == 0 and self.merged0) else B

    def forward(self, x, slot=0, params=None, value_head=None, masks=False):
        """x: [B, in], shared by all groups.  Returns act[-1]  [G, B, H_last].
        `slot` selects an activation set (a stack evaluated on several inputs in one step keeps one
        set per input for the backward); `params` = another ParamStore with the same layout (SAC
        target network).  `value_head` = (group, w [H_last], b [1], out [B]): the scalar head on that group's
        output is evaluated inside the last layer's epilogue pass (no launch of its own).
        `masks`: the hidden layers below the last also leave their Relu6Grad masks as bits (1 bit per element), which the
        deferred backward then reads instead of the activations."""
        B = x.shape[0]
        st = params if params is not None else self.store
        buf = self._buffers(B, slot)
        buf["x"] = x
        self._last = buf
        act, act2d = buf["act"], buf["act2d"]
        for i in range(self.L):
            w, b = st[self.wn[i]], st[self.bn[i]]
            if i == 0 and self.merged0:
                torch.mm(x, w, out=act2d[0])
            elif i == 0:
                for g in range(self.G):
                    torch.mm(x, w[g], out=act[0][g])
            else:
                torch.bmm(act[i - 1], w, out=act[i])
            if value_head is not None and i == self.L - 1 and self._group_rows(i, B) and B % 4 == 0:
                g, vw, vb, vout = value_head
                kernels.bias_act_value(act2d[i], b, B, g, vw, vb, vout, relu6=True)
                value_head = None
            elif masks and i < self.L - 1 and act2d[i].shape[1] % 256 == 0:
                if buf["mask"][i] is None:
                    buf["mask"][i] = kernels.relu6_mask(act2d[i].shape[0], act2d[i].shape[1], self.store.device)
                kernels.bias_act_mask(act2d[i], b, buf["mask"][i], group_rows=self._group_rows(i, B))
                buf["mask_ok"][i] = True
                continue
            else:
                kernels.bias_act(act2d[i], b, relu6=True, group_rows=self._group_rows(i, B))
            buf["mask_ok"][i] = False
        if value_head is not None:
            g, vw, vb, vout = value_head
            if g < 0:
                for gi in range(self.G):
                    kernels.value_head_fwd(act[-1][gi], vw[gi].view(-1), vb[gi].view(-1), out=vout[gi])
            else:
                kernels.value_head_fwd(act[-1][g], vw.view(-1), vb, out=vout)
        return act[-1]

    def forward_pair(self, x_a, slot_a, head_a, x_b, slot_b, params_b, head_b):
        """Two forward passes of this stack in lock step: pass a with the live parameters and mask bits (a backward
        follows), pass b with `params_b` (SAC: the target critics) -- GEMM a, GEMM b, then ONE epilogue launch for both,
        layer by layer; head_* = (w [G,H], b [G(,pad)], out [G,B]): a scalar head on every group inside the last layer's
        epilogue.  Same results, bit for bit, as forward(x_a, slot_a, value_head=(-1,) + head_a, masks=True) and
        forward(x_b, slot_b, params=params_b, value_head=(-1,) + head_b); two launches fewer per layer pair.
        Returns False (nothing done) when a layer does not fit the paired kernels."""
        Ba, Bb = x_a.shape[0], x_b.shape[0]
        if self.L < 2 or Ba % 4 or Bb % 4 or any(self.dims[i + 1] % 256 for i in range(self.L - 1)) or self.dims[-1] % 4:
            return False
        bufs = [self._buffers(Ba, slot_a), self._buffers(Bb, slot_b)]
        bufs[0]["x"], bufs[1]["x"] = x_a, x_b
        self._last = bufs[0]
        stores = (self.store, params_b)
        xs = (x_a, x_b)
        Bs = (Ba, Bb)
        for i in range(self.L):
            for buf, st, x in zip(bufs, stores, xs):
                w = st[self.wn[i]]
                if i == 0 and self.merged0:
                    torch.mm(x, w, out=buf["act2d"][0])
                elif i == 0:
                    for g in range(self.G):
                        torch.mm(x, w[g], out=buf["act"][0][g])
                else:
                    torch.bmm(buf["act"][i - 1], w, out=buf["act"][i])
            if i < self.L - 1:
                a2d = bufs[0]["act2d"][i]
                if bufs[0]["mask"][i] is None:
                    bufs[0]["mask"][i] = kernels.relu6_mask(a2d.shape[0], a2d.shape[1], self.store.device)
                kernels.bias_act_mask_pair(
                    kernels.epilogue_job(a2d, stores[0][self.bn[i]], self._group_rows(i, Ba), mask=bufs[0]["mask"][i]),
                    kernels.epilogue_job(bufs[1]["act2d"][i], stores[1][self.bn[i]], self._group_rows(i, Bb)))
                bufs[0]["mask_ok"][i], bufs[1]["mask_ok"][i] = True, False
            else:
                kernels.bias_act_value_pair(
                    kernels.epilogue_job(bufs[0]["act2d"][i], stores[0][self.bn[i]], Ba, value=head_a),
                    kernels.epilogue_job(bufs[1]["act2d"][i], stores[1][self.bn[i]], Bb, value=head_b))
                bufs[0]["mask_ok"][i] = bufs[1]["mask_ok"][i] = False
        return True

    def backward(self, dlast, need_dx=False, slot=0, need_dw=True, B=None, deferred=None, value=None,
                 dw_rows=None, dx_rows=None):
        """dlast: [G, B, H_last] gradient w.r.t. the (post-ReLU6) output (the stack's own dact[-1] buffer
        or any contiguous tensor); masked in place.
        `deferred` (a list): the bias-gradient column sums are left as per-chunk partials and a reduce job per
        tensor is appended instead of being launched (the caller runs `kernels.reduce_jobs(deferred)` once).
        `value` = (group, w, dvalue [B], grad_w, grad_b) (with `deferred` only): the backward of a scalar head on that
        group's output is folded into the last layer's pass -- dlast[group] is formed as dvalue (x) w, not read.
        A sixth entry (kernels.sac_loss_rider) makes that launch ALSO the loss head that produces dvalue.
        group = -1: a head on EVERY group (w [G,H], dvalue [G,B], grad_w [G,H], grad_b [G,1]; twin Q networks).
        With need_dw=False the partial sums are left unreduced (no jobs; grad_w / grad_b may be None).
        `dw_rows` = (r0, r1) (with `deferred`): only these rows of the batch enter the weight / bias / head gradients --
        the others are carried through the dX chain alone (SAC: Q(s, a_pi) and Q(s, a_hist) evaluated as ONE batch of 2B
        rows; the value loss reaches the critic through the second half, the policy loss only passes through the first
        half to the action).  r0 and r1 must fall on the row chunks of the partial sums.  `dx_rows` = (p0, p1): the
        rows of the input gradient to return.
        Weight/bias gradients are written into the store's flat gradient buffer (need_dw).  Everything runs on the
        current stream (weight-gradient GEMMs on a second stream measured 8 % slower: LAB_NOTEBOOK.md)."""
        if B is None:
            B = dlast.shape[1]
        buf = self._bufs[(B, slot)]
        act, act2d, dact, dact2d, x = buf["act"], buf["act2d"], buf["dact"], buf["dact2d"], buf["x"]
        d, d2d = dlast, dlast.view(self.G * B, -1)
        dx = None

        if value is not None and (deferred is None or not self._group_rows(self.L - 1, B)):
            raise ValueError("a folded value head needs deferred reductions on a batched last layer")
        if dw_rows is not None and deferred is None:
            raise ValueError("a row subset for the weight gradients needs deferred reductions")
        r0, r1 = dw_rows if dw_rows is not None else (0, B)
        rsl = slice(r0, r1)
        for i in reversed(range(self.L)):
            w = self.store[self.wn[i]]
            if deferred is not None:
                # need_dw=False (gradient w.r.t. the input only): the partials are simply never reduced
                ws, ws_value = self._slab_workspaces(buf)
                grows, H = self._group_rows(i, B), act2d[i].shape[1]
                fold = None
                if value is not None and i == self.L - 1:
                    fold = (value[0], value[1], value[2], ws_value) + tuple(value[5:6])     # (+ a loss-head rider)
                if fold is None and buf["mask_ok"][i]:
                    cpg = kernels.act_bwd_partial(None, d2d, grows, ws[i], mask=buf["mask"][i])
                else:
                    cpg = kernels.act_bwd_partial(act2d[i], d2d, grows, ws[i], relu6=True, value=fold)
                # the partial sums are per chunk of `clen` rows: rows [r0, r1) are the slabs [c0, c0 + nc) of a group
                clen = (B + cpg - 1) // cpg
                c0, nc = r0 // clen, (r1 - r0 + clen - 1) // clen
                if r0 % clen or (r1 % clen and r1 != B):
                    raise ValueError("dw_rows %s do not fall on the %d-row chunks of the partial sums" % (dw_rows, clen))
                gb = self.store.g(self.bn[i])
                if need_dw and grows:
                    for g in range(self.G):
                        deferred.append(kernels.job_colsum(ws[i], (g * cpg + c0) * H, nc, H, H, gb[g]))
                elif need_dw:
                    deferred.append(kernels.job_colsum(ws[i], c0 * H, nc, H, H, gb.view(-1)))
                if fold is not None and need_dw:
                    if value[0] < 0:        # a head per group: grad_w [G,H], grad_b [G,>=1], dvalue [G,B]
                        for g in range(self.G):
                            deferred.append(kernels.job_colsum(ws_value, (g * cpg + c0) * H, nc, H, H,
                                                               value[3][g].view(-1)))
                            deferred.append(kernels.job_vecsum(value[2][g][rsl], value[4][g].view(-1)[:1]))
                    else:
                        deferred.append(kernels.job_colsum(ws_value, c0 * H, nc, H, H, value[3].view(-1)))
                        deferred.append(kernels.job_vecsum(value[2][rsl], value[4].view(-1)))
            else:
                kernels.act_bwd_bias_grad(act2d[i], d2d,
                                          self.store.g(self.bn[i]) if need_dw else self._scratch_bias(i),
                                          relu6=True, group_rows=self._group_rows(i, B))
            gw = self.store.g(self.wn[i])
            psl = slice(None) if dx_rows is None else slice(dx_rows[0], dx_rows[1])
            if i == 0:
                xt = x[rsl].t()
                if self.merged0:
                    if need_dw:
                        torch.mm(xt, d2d[rsl], out=gw)
                    if need_dx:
                        # need_dx = (c0, c1): the gradient w.r.t. input columns [c0, c1) only (SAC: the action part)
                        wx = w if need_dx is True else w[need_dx[0]:need_dx[1]]
                        dx = torch.mm(d2d[psl], wx.t())
                else:
                    if need_dw:
                        for g in range(self.G):
                            torch.mm(xt, d[g][rsl], out=gw[g])
                    if need_dx:
                        rows = slice(None) if need_dx is True else slice(need_dx[0], need_dx[1])
                        dx = torch.mm(d[0][psl], w[0][rows].t())
                        for g in range(1, self.G):
                            dx.addmm_(d[g][psl], w[g][rows].t())
            else:
                if need_dw:
                    torch.bmm(act[i - 1][:, rsl].transpose(1, 2), d[:, rsl], out=gw)
                torch.bmm(d, w.transpose(1, 2), out=dact[i - 1])
                d, d2d = dact[i - 1], dact2d[i - 1]
        return dx

    def _scratch_bias(self, i):
        k = ("sb", i)
        if k not in self._bufs:
            self._bufs[k] = torch.empty(self.G, self.dims[i + 1], dtype=torch.float32, device=self.store.device)
        return self._bufs[k]


def clip_by_value_with_gradient(x, l=-1.0, u=1.0):
    """reference ops.py:4-9: forward clip, identity gradient (the backward is written by hand here,
    so this is just the forward)."""
    return torch.clamp(x, l, u)
