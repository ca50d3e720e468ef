"""CPU: the TensorFlow tensor_bundle reader / writer (pfpn_amd/checkpoint.py) against the reference's own
checkpoint index (tests/golden/ref_walk_model.ckpt-78000.index: DATA file of the reference's shipped
DeepMimicWalk DPPO-PFPN checkpoint; the tensor blob itself is not shipped) and the variable inventory decoded
from the .meta graph (tests/golden/meta_constants.json)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF_INDEX = os.path.join(HERE, "golden", "ref_walk_model.ckpt-78000.index")


def test_reference_index_parses_and_reencodes_byte_exact():
    from pfpn_amd import checkpoint as ck
    header, entries = ck.read_index(REF_INDEX, verify=True)           # block checksums verified
    assert header == ck.BUNDLE_HEADER
    raw = open(REF_INDEX, "rb").read()
    assert ck.encode_index(header, entries) == raw                    # the writer reproduces TensorFlow's bytes
    # tensors lie back to back in key order, sizes follow from dtype x shape
    off = 0
    for name in sorted(entries, key=lambda s: s.encode()):
        e = entries[name]
        assert e.offset == off and e.shard_id == 0
        assert e.size == int(np.prod(e.shape, dtype=np.int64)) * np.dtype(ck.DT[e.dtype]).itemsize
        off += e.size
    # known answer: beta1_power = 0.9^78000 underflows to 0.0f, so its stored (masked) CRC-32C is that of 4 zero bytes
    assert entries["beta1_power"].crc == ck.mask_crc(ck.crc32c(b"\x00\x00\x00\x00"))
    assert entries["global_net/episode/episode"].crc == entries["beta1_power"].crc
    assert entries["global_net/episode/episode"].dtype == 3
    assert ck.crc32c(b"123456789") == 0xE3069283


def test_reference_index_matches_meta_graph_inventory():
    from pfpn_amd import checkpoint as ck
    _, entries = ck.read_index(REF_INDEX)
    meta = json.load(open(os.path.join(HERE, "golden", "meta_constants.json")))["walk"]
    trainables = [n for n in entries if n.startswith("global_net/") and "/Adam" not in n and
                  n.split("/")[1] in ("actor", "critic")]
    assert sum(int(np.prod(entries[n].shape)) for n in trainables) == 2104517          # SURVEY: trainable count
    assert entries["global_net/actor/samples"].shape == (36, 35)
    assert entries["global_net/actor/fc_policy/weight"].shape == (512, 1260)
    assert entries["global_net/state_normalizer/mean"].shape == (197,)
    assert entries["step/global_step"].dtype == 9 and entries["step/global_step"].shape == ()
    for n in trainables:                                                # every trainable has both Adam slots
        assert entries[n + "/Adam"].shape == entries[n].shape == entries[n + "/Adam_1"].shape
    # 8 workers' local copies: the 14 trainables + the local statistics
    w0 = sorted(n[len("worker_0/"):] for n in entries if n.startswith("worker_0/"))
    assert all(("actor/" + t.split("/actor/")[1]) in w0 for t in trainables if "/actor/" in t)
    assert sum(1 for n in entries if n.startswith("worker_")) == 8 * len(w0)
    # every variable the .meta graph declares (decoded independently from the GraphDef) is in the index, same shape
    checked = 0
    for name, shape in meta["variables"].items():
        if name in entries:
            assert tuple(shape) == entries[name].shape, name
            checked += 1
    assert checked >= len(trainables)


def test_write_read_roundtrip(tmp_path):
    from pfpn_amd import checkpoint as ck
    rng = np.random.default_rng(0)
    tensors = {"global_net/actor/fc1/weight": rng.normal(size=(197, 64)).astype(np.float32),
               "global_net/actor/fc1/bias": rng.normal(size=64).astype(np.float32),
               "beta1_power": np.float32(0.5), "step/global_step": np.int64(123),
               "global_net/episode/episode": np.int32(7), "empty": np.zeros((0, 3), np.float32)}
    for i in range(40):                                                 # > 16 keys: several restart points
        tensors["worker_%d/x" % i] = rng.normal(size=(3, i + 1)).astype(np.float32)
    prefix = str(tmp_path / "model.ckpt-123")
    ck.write_checkpoint(prefix, tensors)
    assert ck.latest_checkpoint(str(tmp_path)) == prefix
    rd = ck.TFCheckpointReader(prefix)
    assert sorted(rd.names()) == sorted(tensors)
    for k, v in tensors.items():
        got = rd.get(k)
        assert got.dtype == np.asarray(v).dtype and got.shape == np.asarray(v).shape and np.array_equal(got, v)
    # corruption is detected
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(5); f.write(b"\xff")
    with pytest.raises(ValueError):
        [ck.TFCheckpointReader(prefix).get(k) for k in tensors]
    # multi-block tables (tiny block size) parse back
    header, entries = ck.read_index(prefix + ".index")
    blob = ck.encode_index(header, entries, block_size=300)
    p2 = str(tmp_path / "small.index")
    open(p2, "wb").write(blob)
    h2, e2 = ck.read_index(p2)
    assert list(e2) == list(entries) and all(e2[k].encode() == entries[k].encode() for k in entries)


def test_tensorboard_event_file_round_trip(tmp_path):
    """pfpn_amd/summary.py writes the TFRecord / Event / Summary wire format of the reference's `Summary` writer
    (models/utils.py:28-54): scalars and 1000-bin histograms read back with intact checksums."""
    from pfpn_amd.summary import Summary, read_events
    w = Summary(str(tmp_path))
    rng = np.random.default_rng(0)
    vals = rng.normal(0, 1, 5000).astype(np.float32)
    w.add_scalar("loss/value_loss", 0.25, 7)
    w.add_histogram("model/advantage", vals, 7)
    w.add_summaries({"loss": {"policy_loss": np.float32(-1.5)}, "model": {"value": vals[:64]}}, 8)
    w.close()
    ev = read_events(w.path)
    assert [(s, t) for s, t, _ in ev] == [(7, "loss/value_loss"), (7, "model/advantage"), (8, "loss/policy_loss"),
                                          (8, "model/value")]
    assert ev[0][2] == 0.25 and ev[2][2] == -1.5
    h = ev[1][2]
    counts, edges = np.histogram(vals.astype(np.float64), bins=1000)
    assert h["num"] == 5000 and np.array_equal(h["bucket"], counts) and np.allclose(h["bucket_limit"], edges[1:])
    assert h["min"] == vals.min() and h["max"] == vals.max() and abs(h["sum"] - vals.astype(np.float64).sum()) < 1e-9
