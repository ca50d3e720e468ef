"""TensorBoard event files without TensorFlow.

What the reference logs (models/utils.py:17-54: `build_summaries` turns the network's `summaries` dict into scalar /
histogram summaries, `Summary(tf.summary.FileWriter)` adds `add_scalar` / `add_histogram(values, bins=1000)`; the workers
write them every `log_interval` steps).  Here the same `events.out.tfevents.*` file format is written directly:
TFRecord framing (length, masked CRC-32C of the length, payload, masked CRC-32C of the payload) around `Event`
protobufs {wall_time = 1 (double), step = 2 (int64), file_version = 3 (string), summary = 5}, `Summary.Value`
{tag = 1, simple_value = 2 (float), histo = 5}, `HistogramProto` {min = 1, max = 2, num = 3, sum = 4, sum_squares = 5,
bucket_limit = 6 (packed double), bucket = 7 (packed double)} -- so TensorBoard reads the runs of both code bases.
"""
import os
import socket
import struct
import time

import numpy as np

from .checkpoint import _varint, crc32c, mask_crc


def _key(field, wire):
    return _varint((field << 3) | wire)


def _bytes(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _double(field, v):
    return _key(field, 1) + struct.pack("<d", float(v))


def _float(field, v):
    return _key(field, 5) + struct.pack("<f", float(v))


def _int(field, v):
    return _key(field, 0) + _varint(int(v) & ((1 << 64) - 1))


def _packed_doubles(field, values):
    return _bytes(field, struct.pack("<%dd" % len(values), *[float(v) for v in values]))


def histogram_proto(values, bins=1000):
    """reference models/utils.py:34-47: np.histogram with `bins` buckets; limits are the RIGHT bin edges."""
    values = np.asarray(values, np.float64).reshape(-1)
    counts, edges = np.histogram(values, bins=bins)
    return (_double(1, values.min()) + _double(2, values.max()) + _double(3, values.size) + _double(4, values.sum())
            + _double(5, np.square(values).sum()) + _packed_doubles(6, edges[1:]) + _packed_doubles(7, counts))


class Summary(object):
    """`Summary(logdir)` with the reference writer's surface: add_scalar / add_histogram / flush / close."""

    def __init__(self, logdir, filename_suffix=""):
        os.makedirs(logdir, exist_ok=True)
        name = "events.out.tfevents.%010d.%s%s" % (int(time.time()), socket.gethostname(), filename_suffix)
        self.path = os.path.join(logdir, name)
        self._f = open(self.path, "wb")
        self._record(_double(1, time.time()) + _bytes(3, b"brain.Event:2"))

    def _record(self, event):
        head = struct.pack("<Q", len(event))
        self._f.write(head + struct.pack("<I", mask_crc(crc32c(head))) + event + struct.pack("<I", mask_crc(crc32c(event))))

    def _value(self, value_payload, step):
        self._record(_double(1, time.time()) + _int(2, step) + _bytes(5, _bytes(1, value_payload)))

    def add_scalar(self, tag, value, step):
        self._value(_bytes(1, tag.encode()) + _float(2, value), step)

    def add_histogram(self, tag, values, step, bins=1000):
        self._value(_bytes(1, tag.encode()) + _bytes(5, histogram_proto(values, bins)), step)

    def add_summaries(self, summaries, step, prefix=""):
        """a network's (possibly nested) `summaries` dict: scalars as scalars, arrays as histograms
        (reference build_summaries, models/utils.py:17-32)."""
        for key, val in summaries.items():
            tag = "%s/%s" % (prefix, key) if prefix else key
            if isinstance(val, dict):
                self.add_summaries(val, step, tag)
                continue
            val = val.detach().cpu().numpy() if hasattr(val, "detach") else np.asarray(val)
            if val.ndim > 0 and val.size > 1:
                self.add_histogram(tag, val, step)
            else:
                self.add_scalar(tag, float(val.reshape(-1)[0]), step)

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path):
    """-> [(step, tag, value or histogram dict)] of an event file; verifies every record's checksums."""
    from .checkpoint import _read_varint
    buf = open(path, "rb").read()
    out, i = [], 0

    def fields(b):
        j = 0
        while j < len(b):
            key, j = _read_varint(b, j)
            f, w = key >> 3, key & 7
            if w == 0:
                v, j = _read_varint(b, j)
            elif w == 1:
                v, j = b[j:j + 8], j + 8
            elif w == 5:
                v, j = b[j:j + 4], j + 4
            else:
                n, j = _read_varint(b, j)
                v, j = b[j:j + n], j + n
            yield f, w, v
    while i < len(buf):
        n = struct.unpack("<Q", buf[i:i + 8])[0]
        assert struct.unpack("<I", buf[i + 8:i + 12])[0] == mask_crc(crc32c(buf[i:i + 8])), "length checksum"
        ev = buf[i + 12:i + 12 + n]
        assert struct.unpack("<I", buf[i + 12 + n:i + 16 + n])[0] == mask_crc(crc32c(ev)), "payload checksum"
        i += 16 + n
        step, values = 0, []
        for f, w, v in fields(ev):
            if f == 2:
                step = v
            elif f == 5:
                for f2, _, v2 in fields(v):
                    if f2 == 1:
                        tag, val = None, None
                        for f3, w3, v3 in fields(v2):
                            if f3 == 1:
                                tag = bytes(v3).decode()
                            elif f3 == 2:
                                val = struct.unpack("<f", v3)[0]
                            elif f3 == 5:
                                h = {}
                                for f4, w4, v4 in fields(v3):
                                    name = {1: "min", 2: "max", 3: "num", 4: "sum", 5: "sum_squares", 6: "bucket_limit", 7: "bucket"}[f4]
                                    h[name] = struct.unpack("<d", v4)[0] if w4 == 1 else np.frombuffer(v4, "<f8")
                                val = h
                        values.append((tag, val))
        out += [(step, t, v) for t, v in values]
    return out
