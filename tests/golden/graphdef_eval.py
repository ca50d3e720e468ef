#!/usr/bin/env python3
"""A NumPy interpreter for the TF-1.x GraphDef the reference ships in its `model.ckpt-*.meta` files.

TEST INFRASTRUCTURE, build container only (the .meta lives under /root/reference and never travels).
Purpose: produce golden OUTPUT vectors from the reference's own saved dataflow graph -- forward, `tf.gradients`
sub-graph, `clip_by_global_norm`, `ApplyAdam`, the state-normaliser update, the activation statistics and the
`cond/...` resample sub-graph -- without TensorFlow.  The graph (which node feeds which, every constant, every
axis / shape argument, the autodiff expansion TF generated) is the reference's; what this file supplies is the
per-op arithmetic of the ~95 primitive ops the graph uses, written from their published semantics:

* elementwise ops are IEEE fp32 NumPy; reductions use NumPy's order (TF/Eigen order differs -> fp32 tolerance);
* `Softmax` multiplies by the reciprocal of the row sum like TF's CPU functor (softmax_op_functor.h);
* `ApplyAdam` is the TF training_ops kernel: alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)*(1-b1);
  v += (g*g-v)*(1-b2); var -= m*alpha/(sqrt(v)+eps);
* the three random ops (`Multinomial`, `RandomStandardNormal`, `RandomUniform`) are NOT implemented: Philox streams
  cannot be reproduced, so their outputs must be fed (exactly the tensors the product takes as explicit noise inputs);
* accumulators / queues of SyncReplicasOptimizer are not executed: the driver feeds `AccumulatorTakeGradient_i`
  with the mean over workers of the tensors that the graph routes into `AccumulatorApplyGradient_i`.

Variables follow TF-1 ref semantics: a read (`Identity` of a `VariableV2`) aliases the live buffer, stateful ops
(`Assign`, `AssignAdd`, `ApplyAdam`, `Scatter*Update`) write it in place, control inputs (`^name`) run first.
Control flow: `Switch`/`Merge` with dead-tensor propagation; `Enter/Merge/Switch/NextIteration/Exit` while frames
are run iteration by iteration; TensorArrays are Python lists.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_meta_golden as wire  # noqa: E402  (protobuf wire decoder)


class _Dead:
    def __repr__(self):
        return "DEAD"


DEAD = _Dead()

DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_, 7: object}


class Ref:
    """Handle of a ref-typed variable output."""

    def __init__(self, name):
        self.name = name


class TensorArray:
    def __init__(self, size):
        self.items = [None] * int(size)


def parse_attr_full(buf):
    """AttrValue incl. list(i/f/s/type/shape) -- the golden extractor's parser skips lists."""
    out = wire.parse_attr(buf)
    for fn, wt, v in wire.fields(buf):
        if fn == 1:  # ListValue
            lst = {"i": [], "f": [], "s": [], "type": [], "shape": [], "b": []}
            for f2, w2, v2 in wire.fields(v):
                if f2 == 3:
                    if w2 == 2:
                        j = 0
                        while j < len(v2):
                            x, j = wire.varint(v2, j)
                            lst["i"].append(x if x < (1 << 63) else x - (1 << 64))
                    else:
                        lst["i"].append(v2 if v2 < (1 << 63) else v2 - (1 << 64))
                elif f2 == 6:
                    if w2 == 2:
                        j = 0
                        while j < len(v2):
                            x, j = wire.varint(v2, j)
                            lst["type"].append(x)
                    else:
                        lst["type"].append(v2)
                elif f2 == 7:
                    lst["shape"].append(wire.parse_shape(v2))
                elif f2 == 2:
                    lst["s"].append(bytes(v2))
            out["list"] = lst
    return out


def load_graph(path):
    buf = memoryview(open(path, "rb").read())
    nodes = {}
    order = []
    for fn, wt, v in wire.fields(buf):
        if fn != 2:
            continue
        for f2, w2, v2 in wire.fields(v):
            if f2 != 1:
                continue
            node = {"name": None, "op": None, "input": [], "attr": {}}
            for f3, w3, v3 in wire.fields(v2):
                if f3 == 1:
                    node["name"] = bytes(v3).decode()
                elif f3 == 2:
                    node["op"] = bytes(v3).decode()
                elif f3 == 3:
                    node["input"].append(bytes(v3).decode())
                elif f3 == 5:
                    k = val = None
                    for f4, w4, v4 in wire.fields(v3):
                        if f4 == 1:
                            k = bytes(v4).decode()
                        elif f4 == 2:
                            val = parse_attr_full(v4)
                    node["attr"][k] = val
            nodes[node["name"]] = node
            order.append(node["name"])
    return nodes, order


def _split(spec):
    if spec.startswith("^"):
        return spec[1:], None
    if ":" in spec:
        n, i = spec.rsplit(":", 1)
        return n, int(i)
    return spec, 0


def _axes(ax):
    ax = np.asarray(ax).reshape(-1)
    return tuple(int(a) for a in ax)


def _strided_slice(x, begin, end, strides, a):
    bm, em = a.get("begin_mask", {}).get("i", 0), a.get("end_mask", {}).get("i", 0)
    sm = a.get("shrink_axis_mask", {}).get("i", 0)
    assert a.get("ellipsis_mask", {}).get("i", 0) == 0 and a.get("new_axis_mask", {}).get("i", 0) == 0
    idx = []
    for d in range(len(begin)):
        b, e, s = int(begin[d]), int(end[d]), int(strides[d])
        if sm >> d & 1:
            idx.append(b)
            continue
        idx.append(slice(None if bm >> d & 1 else b, None if em >> d & 1 else e, s))
    return np.asarray(x)[tuple(idx)]


class Evaluator:
    def __init__(self, nodes, variables):
        self.nodes = nodes
        self.vars = variables          # name -> ndarray (live state, mutated in place)
        self.consumers = None
        self.frames = {}
        self.executed = []             # names of non-trivial nodes in execution order (for the fixture's provenance)

    # ------------------------------------------------------------------ driver
    def run(self, fetches, feeds=None, cache=None):
        """Evaluate `fetches` (list of 'name' / 'name:i').  `cache` may be passed to continue a previous run
        (same session step): values already computed are reused, so the caller controls the order of reads and
        in-place writes where the graph leaves it open."""
        self.feeds = {}
        for k, v in (feeds or {}).items():
            n, i = _split(k)
            self.feeds[(n, i or 0)] = v
        self.cache = {} if cache is None else cache
        self.iter_cache = None
        self.frame_nodes = frozenset()
        out = []
        for f in fetches:
            n, i = _split(f)
            v = self.value(n, i or 0)
            if isinstance(v, Ref):
                v = self.vars[v.name]
            out.append(np.array(v, copy=True) if isinstance(v, np.ndarray) else v)
        return out

    def value(self, name, idx=0):
        if (name, idx) in self.feeds:
            return self.feeds[(name, idx)]
        outs = self.outputs(name)
        if outs is DEAD:
            return DEAD
        return outs[idx]

    def outputs(self, name):
        c = self.iter_cache if (self.iter_cache is not None and name in self.frame_nodes) else self.cache
        if name in c:
            return c[name]
        node = self.nodes[name]
        op = node["op"]
        if op == "Exit":
            self._run_frame(node)
            return self.cache[name]
        ins = []
        dead = False
        for spec in node["input"]:
            n, i = _split(spec)
            if i is None:
                if self.outputs_or_feed(n) is DEAD:
                    dead = True
                continue
            v = self.value(n, i)
            if v is DEAD:
                dead = True
            ins.append(v)
        if op == "Merge":
            live = [(k, v) for k, v in enumerate(ins) if v is not DEAD]
            res = DEAD if not live else (live[0][1], np.int32(live[0][0]))
        elif dead:
            res = DEAD
        else:
            res = getattr(self, "op_" + op)(node, *ins)
            if not isinstance(res, tuple):
                res = (res,)
            if op not in ("Const", "Identity", "VariableV2", "Placeholder", "Shape", "StridedSlice", "Pack", "NoOp"):
                self.executed.append(name)
        c[name] = res
        return res

    def outputs_or_feed(self, name):
        if (name, 0) in self.feeds:
            return (self.feeds[(name, 0)],)
        return self.outputs(name)

    # ------------------------------------------------------------------ while frames
    def _frame_of(self, exit_node):
        sw = self.nodes[_split(exit_node["input"][0])[0]]
        merge = self.nodes[_split(sw["input"][0])[0]]
        enter = next(self.nodes[_split(s)[0]] for s in merge["input"] if self.nodes[_split(s)[0]]["op"] == "Enter")
        return enter["attr"]["frame_name"]["s"]

    def _run_frame(self, exit_node):
        fname = self._frame_of(exit_node)
        if self.consumers is None:
            self.consumers = {}
            for n in self.nodes.values():
                for s in n["input"]:
                    self.consumers.setdefault(_split(s)[0], []).append(n["name"])
        enters = [n for n in self.nodes.values() if n["op"] in ("Enter", "RefEnter")
                  and n["attr"]["frame_name"]["s"] == fname]
        members, stack = set(), [n["name"] for n in enters]
        while stack:
            x = stack.pop()
            for cns in self.consumers.get(x, []):
                if cns not in members and self.nodes[cns]["op"] != "Exit":
                    members.add(cns)
                    stack.append(cns)
        exits = [n for n in self.nodes.values() if n["op"] == "Exit"
                 and _split(n["input"][0])[0] in members]
        merges = [m for m in members if self.nodes[m]["op"] == "Merge"]
        loopcond = next(m for m in members if self.nodes[m]["op"] == "LoopCond")
        assert self.iter_cache is None, "nested frames not needed by this graph"
        state = {}
        for m in merges:
            ent = next(_split(s)[0] for s in self.nodes[m]["input"] if self.nodes[_split(s)[0]]["op"] == "Enter")
            state[m] = self.value(ent)
        if any(val is DEAD for val in state.values()):          # the frame sits in an untaken cond branch
            for e in exits:
                self.cache[e["name"]] = DEAD
            return
        self.frame_nodes = frozenset(members)
        n_iter = 0
        while True:
            self.iter_cache = {m: (state[m], np.int32(0)) for m in merges}
            pred = bool(self.value(loopcond))
            if not pred:
                for e in exits:
                    sw = self.nodes[_split(e["input"][0])[0]]
                    self.cache[e["name"]] = (self.value(_split(sw["input"][0])[0]),)
                break
            new = {}
            for m in merges:
                nxt = next(_split(s)[0] for s in self.nodes[m]["input"]
                           if self.nodes[_split(s)[0]]["op"] == "NextIteration")
                new[m] = self.value(nxt)
            state = new
            n_iter += 1
            assert n_iter < 100000
        self.iter_cache = None
        self.frame_nodes = frozenset()

    # ------------------------------------------------------------------ ops: plumbing
    def op_Const(self, n):
        return n["attr"]["value"]["tensor"]

    def op_Placeholder(self, n):
        raise KeyError("placeholder %s must be fed" % n["name"])

    def op_PlaceholderWithDefault(self, n, x):
        return x

    def op_VariableV2(self, n):
        if n["name"] not in self.vars:
            raise KeyError("variable %s has no value" % n["name"])
        return Ref(n["name"])

    def _deref(self, x):
        return self.vars[x.name] if isinstance(x, Ref) else x

    def op_Identity(self, n, x):
        return self._deref(x)        # aliases the live buffer, like TF's Identity on a ref input

    op_StopGradient = op_Identity
    op_Enter = op_Identity
    op_NextIteration = op_Identity
    op_LoopCond = op_Identity

    def op_RefEnter(self, n, x):
        return x

    def op_IdentityN(self, n, *xs):
        return tuple(self._deref(x) for x in xs)

    def op_NoOp(self, n):
        return None

    def op_Switch(self, n, data, pred):
        return (DEAD, data) if bool(pred) else (data, DEAD)

    op_RefSwitch = op_Switch

    def op_Assign(self, n, ref, val):
        val = np.asarray(self._deref(val))
        cur = self.vars[ref.name]
        assert cur.shape == val.shape, (n["name"], cur.shape, val.shape)
        cur[...] = val
        return ref

    def op_AssignAdd(self, n, ref, val):
        self.vars[ref.name][...] += np.asarray(val)
        return ref

    def op_ApplyAdam(self, n, var, m, v, b1p, b2p, lr, b1, b2, eps, g):
        assert not n["attr"]["use_nesterov"]["b"]
        var, m, v = self.vars[var.name], self.vars[m.name], self.vars[v.name]
        one = np.float32(1)
        b1p, b2p, lr, b1, b2, eps = (np.float32(self._deref(t)) for t in (b1p, b2p, lr, b1, b2, eps))
        alpha = lr * np.sqrt(one - b2p) / (one - b1p)
        m += (g - m) * (one - b1)
        v += (g * g - v) * (one - b2)
        var -= (m * alpha) / (np.sqrt(v) + eps)
        return Ref(n["input"][0])

    def op_ScatterUpdate(self, n, ref, idx, upd):
        self.vars[ref.name][np.asarray(idx)] = upd      # duplicates: last write wins, like the CPU kernel's loop
        return ref

    def op_ScatterNdUpdate(self, n, ref, idx, upd):
        idx = np.asarray(idx)
        tgt = self.vars[ref.name]
        for r in range(idx.shape[0]):
            tgt[tuple(idx[r])] = upd[r]
        return ref

    # ------------------------------------------------------------------ ops: shapes
    def op_Shape(self, n, x):
        return np.asarray(np.shape(self._deref(x)), dtype=DT[n["attr"]["out_type"]["type"]])

    def op_Size(self, n, x):
        return np.asarray(np.size(x), dtype=DT[n["attr"]["out_type"]["type"]])

    def op_Reshape(self, n, x, shape):
        return np.reshape(self._deref(x), _axes(shape))

    def op_ExpandDims(self, n, x, d):
        return np.expand_dims(x, int(d))

    def op_Squeeze(self, n, x):
        dims = n["attr"].get("squeeze_dims", {}).get("list", {}).get("i", [])
        return np.squeeze(x, axis=tuple(dims) if dims else None)

    def op_StridedSlice(self, n, x, b, e, s):
        return _strided_slice(x, np.asarray(b).reshape(-1), np.asarray(e).reshape(-1), np.asarray(s).reshape(-1),
                              n["attr"])

    def op_Pack(self, n, *xs):
        return np.stack([np.asarray(x) for x in xs], axis=n["attr"].get("axis", {}).get("i", 0))

    def op_ConcatV2(self, n, *xs):
        return np.concatenate([np.atleast_1d(x) for x in xs[:-1]], axis=int(xs[-1]))

    def op_Fill(self, n, dims, v):
        return np.full(_axes(dims), v, dtype=np.asarray(v).dtype)

    def op_Range(self, n, a, b, d):
        return np.arange(a, b, d, dtype=np.asarray(a).dtype)

    def op_Tile(self, n, x, m):
        return np.tile(x, _axes(m))

    def op_Transpose(self, n, x, perm):
        return np.transpose(x, _axes(perm))

    def op_ZerosLike(self, n, x):
        return np.zeros_like(x)

    def op_Cast(self, n, x):
        return np.asarray(x).astype(DT[n["attr"]["DstT"]["type"]])

    def op_Split(self, n, axis, x):
        return tuple(np.split(x, n["attr"]["num_split"]["i"], axis=int(axis)))

    def op_SplitV(self, n, x, sizes, axis):
        sizes = [int(s) for s in np.asarray(sizes).reshape(-1)]
        if -1 in sizes:
            sizes[sizes.index(-1)] = x.shape[int(axis)] - (sum(sizes) + 1)
        return tuple(np.split(x, np.cumsum(sizes)[:-1], axis=int(axis)))

    def op_BroadcastGradientArgs(self, n, s0, s1):
        s0, s1 = [int(v) for v in s0], [int(v) for v in s1]
        r = max(len(s0), len(s1))
        p0, p1 = [1] * (r - len(s0)) + s0, [1] * (r - len(s1)) + s1
        r0 = [d for d in range(r) if p0[d] == 1 and p1[d] != 1 or d < r - len(s0)]
        r1 = [d for d in range(r) if p1[d] == 1 and p0[d] != 1 or d < r - len(s1)]
        # TF also lists dims where BOTH are 1 for both sides
        both = [d for d in range(r) if p0[d] == 1 and p1[d] == 1]
        r0 = sorted(set(r0) | set(both))
        r1 = sorted(set(r1) | set(both))
        return np.asarray(r0, np.int32), np.asarray(r1, np.int32)

    def op_DynamicStitch(self, n, *xs):
        k = n["attr"]["N"]["i"]
        idx, data = xs[:k], xs[k:]
        size = max(int(np.max(i)) for i in idx if np.size(i)) + 1
        first = np.asarray(data[0])
        out = np.zeros((size,) + first.shape[np.ndim(idx[0]):], dtype=first.dtype)
        for i, d in zip(idx, data):
            out[np.asarray(i)] = d
        return out

    # ------------------------------------------------------------------ ops: arithmetic (fp32 stays fp32)
    def op_Add(self, n, a, b):
        return np.add(a, b)

    def op_AddN(self, n, *xs):
        out = xs[0]
        for x in xs[1:]:
            out = np.add(out, x)
        return out

    def op_Sub(self, n, a, b):
        return np.subtract(a, b)

    def op_Mul(self, n, a, b):
        return np.multiply(self._deref(a), self._deref(b))

    def op_RealDiv(self, n, a, b):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.divide(a, b)

    def op_FloorDiv(self, n, a, b):
        return np.floor_divide(a, b)

    def op_FloorMod(self, n, a, b):
        return np.mod(a, b)

    def op_Neg(self, n, a):
        return np.negative(a)

    def op_Square(self, n, a):
        return np.multiply(a, a)

    def op_SquaredDifference(self, n, a, b):
        d = np.subtract(a, b)
        return d * d

    def op_Sqrt(self, n, a):
        return np.sqrt(a)

    def op_Exp(self, n, a):
        return np.exp(self._deref(a))

    def op_Log(self, n, a):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.log(a)

    def op_Reciprocal(self, n, a):
        with np.errstate(divide="ignore"):
            return np.float32(1) / a

    def op_Maximum(self, n, a, b):
        return np.maximum(a, b)

    def op_Minimum(self, n, a, b):
        return np.minimum(a, b)

    def op_Relu6(self, n, a):
        return np.minimum(np.maximum(a, np.float32(0)), np.float32(6))

    def op_Relu6Grad(self, n, g, f):
        return np.where((f > 0) & (f < 6), g, np.float32(0)).astype(np.float32)

    def op_Softmax(self, n, x):
        e = np.exp(x - x.max(axis=-1, keepdims=True))
        return e * (np.float32(1) / e.sum(axis=-1, keepdims=True))

    def op_L2Loss(self, n, x):
        return np.float32(np.sum(x * x) * np.float32(0.5))

    def op_MatMul(self, n, a, b):
        a, b = self._deref(a), self._deref(b)
        if n["attr"]["transpose_a"]["b"]:
            a = a.T
        if n["attr"]["transpose_b"]["b"]:
            b = b.T
        return np.matmul(a, b)

    def _reduce(self, fn, n, x, ax):
        return fn(x, axis=_axes(ax), keepdims=bool(n["attr"]["keep_dims"]["b"]))

    def op_Sum(self, n, x, ax):
        return self._reduce(np.sum, n, np.asarray(x), ax)

    def op_Mean(self, n, x, ax):
        return self._reduce(np.mean, n, np.asarray(x), ax)

    def op_Max(self, n, x, ax):
        return self._reduce(np.max, n, np.asarray(x), ax)

    def op_Prod(self, n, x, ax):
        return self._reduce(np.prod, n, np.asarray(x), ax)

    # ------------------------------------------------------------------ ops: logic / indexing
    def op_Less(self, n, a, b):
        return np.less(a, b)

    def op_LessEqual(self, n, a, b):
        return np.less_equal(a, b)

    def op_GreaterEqual(self, n, a, b):
        return np.greater_equal(self._deref(a), b)

    def op_Equal(self, n, a, b):
        return np.equal(a, b)

    def op_LogicalAnd(self, n, a, b):
        return np.logical_and(a, b)

    def op_LogicalOr(self, n, a, b):
        return np.logical_or(a, b)

    def op_LogicalNot(self, n, a):
        return np.logical_not(a)

    def op_IsNan(self, n, a):
        return np.isnan(a)

    def op_IsInf(self, n, a):
        return np.isinf(a)

    def op_IsFinite(self, n, a):
        return np.isfinite(a)

    def op_Select(self, n, c, t, e):
        c = np.asarray(c)
        if c.ndim == 1 and np.ndim(t) > 1:
            c = c.reshape((-1,) + (1,) * (np.ndim(t) - 1))
        return np.where(c, t, e)

    def op_Where(self, n, c):
        return np.argwhere(c).astype(np.int64)

    def op_OneHot(self, n, idx, depth, on, off):
        assert n["attr"].get("axis", {}).get("i", -1) == -1
        idx = np.asarray(idx)
        out = np.full(idx.shape + (int(depth),), off, dtype=np.asarray(on).dtype)
        np.put_along_axis(out, idx[..., None].astype(np.int64), on, axis=-1)
        return out

    def op_GatherV2(self, n, p, idx, axis):
        return np.take(self._deref(p), np.asarray(idx), axis=int(axis))

    def op_GatherNd(self, n, p, idx):
        idx = np.asarray(idx)
        return np.asarray(p)[tuple(idx[..., k] for k in range(idx.shape[-1]))]

    def op_UniqueWithCounts(self, n, x):
        x = np.asarray(x)
        seen, y, idx, cnt = {}, [], np.empty(x.shape, DT[n["attr"]["out_idx"]["type"]]), []
        for i, v in enumerate(x.tolist()):
            if v not in seen:
                seen[v] = len(y)
                y.append(v)
                cnt.append(0)
            idx[i] = seen[v]
            cnt[seen[v]] += 1
        return np.asarray(y, x.dtype), idx, np.asarray(cnt, idx.dtype)

    # ------------------------------------------------------------------ ops: tensor arrays
    def op_TensorArrayV3(self, n, size):
        return TensorArray(size), np.float32(0)

    def op_TensorArrayScatterV3(self, n, h, idx, val, flow):
        for k, i in enumerate(np.asarray(idx).tolist()):
            h.items[i] = np.asarray(val)[k]
        return np.float32(0)

    def op_TensorArrayReadV3(self, n, h, i, flow):
        return h.items[int(i)]

    def op_TensorArrayWriteV3(self, n, h, i, val, flow):
        h.items[int(i)] = val
        return np.float32(0)

    def op_TensorArraySizeV3(self, n, h, flow):
        return np.int32(len(h.items))

    def op_TensorArrayGatherV3(self, n, h, idx, flow):
        return np.stack([h.items[i] for i in np.asarray(idx).tolist()])

    # ------------------------------------------------------------------ ops that must be fed
    def _must_feed(self, n, *a):
        raise KeyError("random op %s (%s) must be fed" % (n["name"], n["op"]))

    op_Multinomial = op_RandomStandardNormal = op_RandomUniform = op_TruncatedNormal = _must_feed
