"""TensorFlow checkpoint (tensor_bundle) files <-> the MI355X build's parameter buffers.

The reference saves and restores through `tf.train.Saver` (models/distributed_model.py:436-496, :608-619):
`<prefix>.index` is a LevelDB-style table (prefix-compressed sorted keys, 16-entry restart points, one
BundleEntryProto per variable, a BundleHeaderProto under the empty key, 5-byte block trailers with a masked
CRC-32C, 48-byte footer), `<prefix>.data-00000-of-00001` the raw little-endian tensors back to back in key
order.  TensorFlow is not needed on either side: this module parses and writes the format itself, so a
reference-trained policy drops into `ParticleFilteringClipPPONetwork` (and a policy trained here can be
restored by the reference).  Variable names are the reference graph's (`global_net/actor/fc1/weight`,
`.../Adam`, `.../Adam_1`, `beta1_power`, `global_net/state_normalizer/mean`, `global_net/max_active_degree`,
`step/global_step`, ...: tests/golden/meta_constants.json).

CRC-32C runs in libpfpn_hip.so (`pfpn_crc32c`, host code).
"""
import os
import struct
from collections import OrderedDict

import numpy as np

from . import _lib

TABLE_MAGIC = 0xdb4775248b80fb57
BLOCK_SIZE = 262144                  # tensorflow/core/lib/io/table_options.h default
RESTART_INTERVAL = 16
# tensorflow/core/framework/types.proto
DT = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 6: np.int8, 9: np.int64, 10: np.bool_}
DT_OF = {np.dtype(v): k for k, v in DT.items()}


# --------------------------------------------------------------------------------------- #
# primitives                                                                               #
# --------------------------------------------------------------------------------------- #
def crc32c(data, crc=0):
    data = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    return int(_lib.load().pfpn_crc32c(bytes(data), len(data), crc)) & 0xffffffff


def mask_crc(crc):
    """tensorflow/core/lib/hash/crc32c.h Mask(): rotate right 15, add a constant."""
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, i):
    r = s = 0
    while True:
        c = buf[i]
        i += 1
        r |= (c & 0x7f) << s
        s += 7
        if c < 0x80:
            return r, i


class BundleEntry(object):
    """tensorflow/core/protobuf/tensor_bundle.proto BundleEntryProto (no slices)."""

    def __init__(self, dtype, shape, shard_id=0, offset=0, size=0, crc=0):
        self.dtype, self.shape, self.shard_id, self.offset, self.size, self.crc = dtype, tuple(shape), shard_id, offset, size, crc

    def encode(self):
        dims = b"".join(b"\x12" + _varint(len(_varint(d)) + 1) + b"\x08" + _varint(d) for d in self.shape)
        out = b"\x08" + _varint(self.dtype) + b"\x12" + _varint(len(dims)) + dims
        if self.shard_id:
            out += b"\x18" + _varint(self.shard_id)
        if self.offset:
            out += b"\x20" + _varint(self.offset)
        if self.size:
            out += b"\x28" + _varint(self.size)
        return out + b"\x35" + struct.pack("<I", self.crc)

    @classmethod
    def decode(cls, buf):
        e = cls(0, ())
        i = 0
        while i < len(buf):
            tag, i = _read_varint(buf, i)
            f, wt = tag >> 3, tag & 7
            if wt == 0:
                v, i = _read_varint(buf, i)
                if f == 1: e.dtype = v
                elif f == 3: e.shard_id = v
                elif f == 4: e.offset = v
                elif f == 5: e.size = v
            elif wt == 5:
                v = struct.unpack_from("<I", buf, i)[0]
                i += 4
                if f == 6: e.crc = v
            elif wt == 2:
                ln, i = _read_varint(buf, i)
                sub = buf[i:i + ln]
                i += ln
                if f == 2:                                  # TensorShapeProto: repeated Dim dim = 2 {int64 size = 1}
                    dims, j = [], 0
                    while j < len(sub):
                        t2, j = _read_varint(sub, j)
                        l2, j = _read_varint(sub, j)
                        d, k = 0, j
                        while k < j + l2:
                            t3, k = _read_varint(sub, k)
                            v3, k = _read_varint(sub, k)
                            if t3 >> 3 == 1: d = v3
                        j += l2
                        if t2 >> 3 == 2: dims.append(d)
                    e.shape = tuple(dims)
                elif f == 7:
                    raise NotImplementedError("sliced (partitioned) variables are not used by the reference")
            else:
                raise ValueError("BundleEntryProto: unexpected wire type %d" % wt)
        return e


# --------------------------------------------------------------------------------------- #
# table reader                                                                             #
# --------------------------------------------------------------------------------------- #
def _read_block(buf, off, size, verify):
    body = buf[off:off + size]
    if verify:
        want = struct.unpack_from("<I", buf, off + size + 1)[0]
        got = mask_crc(crc32c(buf[off:off + size + 1]))
        if want != got:
            raise ValueError("table block at %d: checksum mismatch" % off)
    if buf[off + size] != 0:
        raise NotImplementedError("compressed table blocks (TensorFlow writes none)")
    nr = struct.unpack_from("<I", body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * nr
    i, key, out = 0, b"", []
    while i < end:
        sh, i = _read_varint(body, i)
        ns, i = _read_varint(body, i)
        vl, i = _read_varint(body, i)
        key = key[:sh] + body[i:i + ns]
        i += ns
        out.append((key, body[i:i + vl]))
        i += vl
    return out


def read_index(path, verify=True):
    """<prefix>.index -> (header bytes, OrderedDict name -> BundleEntry), keys in file order."""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow table file" % path)
    foot = buf[-48:]
    _, i = _read_varint(foot, 0)
    _, i = _read_varint(foot, i)
    ix_off, i = _read_varint(foot, i)
    ix_size, i = _read_varint(foot, i)
    header, entries = None, OrderedDict()
    for _, handle in _read_block(buf, ix_off, ix_size, verify):
        off, j = _read_varint(handle, 0)
        size, j = _read_varint(handle, j)
        for key, val in _read_block(buf, off, size, verify):
            if key == b"":
                header = val
            else:
                entries[key.decode()] = BundleEntry.decode(val)
    return header, entries


class TFCheckpointReader(object):
    """read access to <prefix>.index / <prefix>.data-*: names, shapes, tensors."""

    def __init__(self, prefix, verify=True):
        self.prefix = prefix
        self.verify = verify
        self.header, self.entries = read_index(prefix + ".index", verify)
        self._shards = {}

    def names(self):
        return list(self.entries)

    def shape(self, name):
        return self.entries[name].shape

    def has_data(self):
        return os.path.exists(self._shard_path(0))

    def _shard_path(self, shard):
        return "%s.data-%05d-of-%05d" % (self.prefix, shard, 1)

    def get(self, name):
        e = self.entries[name]
        if e.shard_id not in self._shards:
            self._shards[e.shard_id] = open(self._shard_path(e.shard_id), "rb")
        f = self._shards[e.shard_id]
        f.seek(e.offset)
        raw = f.read(e.size)
        if len(raw) != e.size:
            raise ValueError("%s: data shard is truncated" % name)
        if self.verify and mask_crc(crc32c(raw)) != e.crc:
            raise ValueError("%s: tensor checksum mismatch" % name)
        return np.frombuffer(raw, dtype=DT[e.dtype]).reshape(e.shape).copy()


# --------------------------------------------------------------------------------------- #
# table writer                                                                             #
# --------------------------------------------------------------------------------------- #
class _BlockBuilder(object):
    def __init__(self, restart_interval):
        self.ri = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b""

    def add(self, key, value):
        shared = 0
        if self.count < self.ri:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _shortest_separator(a, b):
    """leveldb BytewiseComparator::FindShortestSeparator."""
    n = min(len(a), len(b))
    i = 0
    while i < n and a[i] == b[i]:
        i += 1
    if i < n and a[i] < 0xff and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def _short_successor(a):
    """leveldb BytewiseComparator::FindShortSuccessor."""
    for i, c in enumerate(a):
        if c != 0xff:
            return a[:i] + bytes([c + 1])
    return a


def encode_index(header, entries, block_size=BLOCK_SIZE):
    """(header bytes, ordered {name: BundleEntry}) -> bytes of the .index table, as TensorFlow's TableBuilder
    lays it out (uncompressed blocks)."""
    out = bytearray()
    index = _BlockBuilder(1)
    pending = None                                    # (last key of the flushed block, handle)

    def write_block(body):
        off = len(out)
        out.extend(body)
        out.append(0)                                 # kNoCompression
        out.extend(struct.pack("<I", mask_crc(crc32c(bytes(body) + b"\x00"))))
        return _varint(off) + _varint(len(body))

    data = _BlockBuilder(RESTART_INTERVAL)
    items = [(b"", header)] + [(k.encode(), e.encode()) for k, e in sorted(entries.items())]
    for key, val in items:
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), pending[1])
            pending = None
        data.add(key, val)
        if data.size_estimate() >= block_size:
            pending = (key, write_block(data.finish()))
            data = _BlockBuilder(RESTART_INTERVAL)
    if not data.empty():
        pending = (data.last, write_block(data.finish()))
    meta_handle = write_block(_BlockBuilder(RESTART_INTERVAL).finish())
    if pending is not None:
        index.add(_short_successor(pending[0]), pending[1])
    index_handle = write_block(index.finish())
    foot = meta_handle + index_handle
    out.extend(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", TABLE_MAGIC))
    return bytes(out)


BUNDLE_HEADER = b"\x08\x01\x1a\x02\x08\x01"          # num_shards 1, little endian, version.producer 1


def write_checkpoint(prefix, tensors):
    """tensors: {variable name: numpy array} -> <prefix>.index, <prefix>.data-00000-of-00001 and the
    `checkpoint` state file tf.train.latest_checkpoint reads."""
    entries, off = {}, 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name], order="C")      # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in DT_OF:
                raise TypeError("%s: dtype %s has no TensorFlow counterpart here" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            f.write(raw)
            entries[name] = BundleEntry(DT_OF[a.dtype], a.shape, 0, off, len(raw), mask_crc(crc32c(raw)))
            off += len(raw)
    with open(prefix + ".index", "wb") as f:
        f.write(encode_index(BUNDLE_HEADER, entries))
    base = os.path.basename(prefix)
    with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return entries


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by <directory>/checkpoint."""
    p = os.path.join(directory, "checkpoint")
    if not os.path.exists(p):
        return None
    for line in open(p):
        if line.startswith("model_checkpoint_path:"):
            return os.path.join(directory, line.split('"')[1])
    return None


# --------------------------------------------------------------------------------------- #
# network <-> reference variable names                                                     #
# --------------------------------------------------------------------------------------- #
SCOPE = "global_net/"


def network_tensors(net, optimizer=None, n_workers=0, total_samples=0, episodes=0):
    """Everything the reference's Saver stores for a DPPO-PFPN run, under the reference's names.
    `n_workers` > 0 also writes the per-worker local copies (`worker_k/...`) the reference graph expects."""
    cpu = lambda t: t.detach().cpu().numpy()
    from . import kernels
    kernels.check_faults()          # never write the parameters of a run a waiting kernel has poisoned (kernels.DeviceFault)
    out = OrderedDict()
    var = net.variables()
    for name, t in var.items():
        out[SCOPE + name] = cpu(t).astype(np.float32)
    if optimizer is not None and net.store.m is not None:
        if hasattr(optimizer, "gather_optimizer_state"):
            optimizer.gather_optimizer_state(net)        # (sharded exchange: every replica takes part in the all-gather)
        st = net.store
        for slot, buf in (("Adam", st.m), ("Adam_1", st.v)):
            for name, t in net.variables(st.views_of(buf)).items():
                out[SCOPE + name + "/" + slot] = cpu(t).astype(np.float32)
        for gi, g in enumerate(optimizer.groups):           # one Adam per group: TF suffixes the later ones _1, _2
            bp = cpu(g["beta_pow"])
            sfx = "" if gi == 0 else "_%d" % gi
            out["beta1_power" + sfx], out["beta2_power" + sfx] = np.float32(bp[0]), np.float32(bp[1])
    if hasattr(net, "target_variables"):
        # SAC target critics (sac.py:133-135 builds them under variable_scope('target_net')).  The reference ships no
        # SAC checkpoint, so unlike the DPPO names above this scope name is not pinned by reference data.
        for name, t in net.target_variables().items():
            out[SCOPE + "target_net/" + name] = cpu(t).astype(np.float32)
    if getattr(net, "target_norm", None) is not None:
        # TD3 keeps a FULL target network, its normaliser statistics included (ddpg.py:56-60: the target net is built by
        # the same init() under variable_scope('target_net') and soft-updated variable by variable)
        out[SCOPE + "target_net/state_normalizer/mean"] = cpu(net.target_norm[0])
        out[SCOPE + "target_net/state_normalizer/std"] = cpu(net.target_norm[1])
    if getattr(net, "state_normalizer", None) is not None:
        out[SCOPE + "state_normalizer/mean"] = cpu(net.state_normalizer.mean)
        out[SCOPE + "state_normalizer/std"] = cpu(net.state_normalizer.std)
    if getattr(net, "max_active", None) is not None:
        out[SCOPE + "max_active_degree"] = cpu(net.max_active)
        out[SCOPE + "sum_active_degree"] = cpu(net.sum_active)
        out[SCOPE + "resample/train_flag"] = np.float32(net.train_flag)
    out[SCOPE + "samples/samples"] = np.int64(total_samples)
    out[SCOPE + "episode/episode"] = np.int32(episodes)
    out["step/global_step"] = np.int64(net.global_step)
    for k in range(n_workers):
        for name, t in var.items():
            out["worker_%d/%s" % (k, name)] = out[SCOPE + name]
    return out


def save(prefix, net, optimizer=None, **kw):
    return write_checkpoint(prefix, network_tensors(net, optimizer, **kw))


def restore(prefix, net, optimizer=None, strict=True):
    """Load a reference (or own) checkpoint into `net` (and the Adam state into `optimizer`)."""
    import torch
    rd = TFCheckpointReader(prefix)
    var = net.variables()
    missing = [n for n in var if SCOPE + n not in rd.entries]
    if missing and strict:
        raise KeyError("checkpoint %s lacks %s" % (prefix, missing[:4]))
    net.load_variables({n: rd.get(SCOPE + n) for n in var if SCOPE + n in rd.entries})
    if optimizer is not None and net.store.m is not None and SCOPE + next(iter(var)) + "/Adam" in rd.entries:
        st = net.store
        for slot, buf in (("Adam", st.m), ("Adam_1", st.v)):
            net.load_variables({n: rd.get(SCOPE + n + "/" + slot) for n in var}, st.views_of(buf))
        for gi, g in enumerate(optimizer.groups):
            sfx = "" if gi == 0 or ("beta1_power_%d" % gi) not in rd.entries else "_%d" % gi
            g["beta_pow"].copy_(torch.tensor([float(rd.get("beta1_power" + sfx)), float(rd.get("beta2_power" + sfx))]))
    if hasattr(net, "target_variables"):
        tv = net.target_variables()
        if all(SCOPE + "target_net/" + n in rd.entries for n in tv):
            with torch.no_grad():
                for n, t in tv.items():
                    t.copy_(torch.from_numpy(rd.get(SCOPE + "target_net/" + n)).to(t.device).view(t.shape))
            net._target_initialised = True
    if getattr(net, "state_normalizer", None) is not None and SCOPE + "state_normalizer/mean" in rd.entries:
        net.state_normalizer.mean.copy_(torch.from_numpy(rd.get(SCOPE + "state_normalizer/mean")))
        net.state_normalizer.std.copy_(torch.from_numpy(rd.get(SCOPE + "state_normalizer/std")))
    if getattr(net, "target_norm", None) is not None:
        # the target network's own statistics; a checkpoint without them (written before they were stored) starts the
        # target normaliser from the live statistics instead of from mean 0 / std 1
        key = SCOPE + "target_net/state_normalizer/"
        if key + "mean" in rd.entries:
            net.target_norm[0].copy_(torch.from_numpy(rd.get(key + "mean")))
            net.target_norm[1].copy_(torch.from_numpy(rd.get(key + "std")))
        elif getattr(net, "state_normalizer", None) is not None:
            net.target_norm[0].copy_(net.state_normalizer.mean)
            net.target_norm[1].copy_(net.state_normalizer.std)
    if getattr(net, "max_active", None) is not None and SCOPE + "max_active_degree" in rd.entries:
        net.max_active.copy_(torch.from_numpy(rd.get(SCOPE + "max_active_degree")))
        net.sum_active.copy_(torch.from_numpy(rd.get(SCOPE + "sum_active_degree")))
        net.train_flag = int(rd.get(SCOPE + "resample/train_flag"))
    if "step/global_step" in rd.entries:
        net.global_step = int(rd.get("step/global_step"))
        if hasattr(net, "step_dev"):
            net.step_dev.fill_(float(net.global_step))
        if getattr(net, "noise_epoch", None) is not None:      # SAC: the learner's noise draws continue, they do not restart
            net.noise_epoch.fill_(int(net.global_step))
    return rd
