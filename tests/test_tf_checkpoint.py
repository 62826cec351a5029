"""CPU: the TensorFlow V2 checkpoint reader / writer against the formats' own known answers and by round trip
(no TensorFlow here: see the STATUS note in drs_amd/tf_checkpoint.py)."""
import os
import struct

import numpy as np
import pytest

from drs_amd import tf_checkpoint as C


def test_primitives_known_answers():
    assert C.crc32c(b"123456789") == 0xE3069283                    # the CRC-32C check value
    assert C.crc32c(b"") == 0 and C.crc32c(b"\x00" * 32) == 0x8A9136AA          # RFC 3720 B.4
    blob = os.urandom(300001)
    assert C.crc32c_fast(blob) == C.crc32c(blob) and C.crc32c_fast(blob[:70000], lanes=64) == C.crc32c(blob[:70000])
    for v in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert C.unmask_crc(C.mask_crc(v)) == v and C.mask_crc(v) != v
    assert C._put_varint(300) == b"\xac\x02" and C._get_varint(b"\xac\x02", 0) == (300, 2)
    assert C.snappy_decompress(bytes([10, 0x00, 0x61, 0x15, 0x01])) == b"a" * 10           # literal + overlapping copy
    assert C.snappy_decompress(bytes([5, 0x10]) + b"hello") == b"hello"
    e = C.decode_entry(C.encode_entry(C.DT_FLOAT, (3, 4), 96, 48, 0xDEADBEEF))
    assert e == dict(dtype=1, shape=[3, 4], shard_id=0, offset=96, size=48, crc32c=0xDEADBEEF)
    # a hand-assembled BundleEntryProto: dtype=9 (int64), scalar shape, offset omitted, size 8
    raw = bytes([0x08, 9, 0x12, 0, 0x28, 8])
    assert C.decode_entry(raw)["dtype"] == 9 and C.decode_entry(raw)["shape"] == [] and C.decode_entry(raw)["size"] == 8


def test_table_layout_and_checksums():
    items = [(("key%04d" % i).encode(), os.urandom(i % 50)) for i in range(500)] + [(b"", b"header")]
    buf = C.write_table(items, block_size=512)
    assert struct.unpack("<Q", buf[-8:])[0] == 0xdb4775248b80fb57 and len(buf[-48:]) == 48
    got = C.read_table(buf)
    assert got == sorted(items) and got[0][0] == b""
    bad = bytearray(buf)
    bad[10] ^= 0xFF
    with pytest.raises(ValueError):
        C.read_table(bytes(bad))
    with pytest.raises(ValueError):
        C.read_table(buf[:-1] + b"\x00")


def test_bundle_roundtrip_and_net_mapping(tmp_path):
    rng = np.random.default_rng(0)
    t = {"conv1/weights": rng.normal(size=(5, 5, 3, 64)).astype(np.float32), "conv1/biases": np.full(64, 0.1, np.float32),
         "conv1/moving_mean": rng.normal(size=64).astype(np.float32), "conv1/weights/Momentum": np.zeros((5, 5, 3, 64), np.float32),
         "main_global_step": np.array(484000, dtype=np.int32), "aux64": np.arange(6, dtype=np.int64).reshape(2, 3)}
    prefix = str(tmp_path / "model-484000")
    C.write_bundle(prefix, t)
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(a.nbytes for a in t.values())
    back = C.read_bundle(prefix)
    assert set(back) == set(t)
    for k in t:
        assert back[k].dtype == t[k].dtype and back[k].shape == t[k].shape
        np.testing.assert_array_equal(back[k], t[k])
    with open(prefix + ".data-00000-of-00001", "r+b") as f:           # flip one tensor byte: the per-tensor CRC must catch it
        f.seek(100)
        b = f.read(1)
        f.seek(100)
        f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ValueError):
        C.read_bundle(prefix)

    class FakeNet(object):                                             # the DilatedNet surface the mapping uses
        class plan:
            offsets = {"conv1/weights": None, "conv1/biases": None}
        global_step = 7

        def __init__(self):
            self.v = {"conv1/weights": t["conv1/weights"], "conv1/biases": t["conv1/biases"], "conv1/moving_mean": t["conv1/moving_mean"],
                      "conv1/moving_variance": np.ones(64, np.float32)}
            self.m = {k: np.full_like(self.v[k], 0.5) for k in ("conv1/weights", "conv1/biases")}

        def variable_names(self):
            return list(self.v)

        def get_variable(self, n, slot=None):
            return (self.m if slot else self.v)[n]

        def set_variable(self, n, val, slot=None):
            (self.m if slot else self.v)[n] = np.asarray(val)
    a, b = FakeNet(), FakeNet()
    p2 = str(tmp_path / "model-7")
    C.save_tf_checkpoint(a, p2)
    b.v = {k: np.zeros_like(v) for k, v in b.v.items()}
    b.global_step = 0
    extra = C.load_tf_checkpoint(b, p2)
    assert b.global_step == 7 and extra == ["main_global_step"]
    for k in a.v:
        np.testing.assert_array_equal(a.v[k], b.v[k])
    np.testing.assert_array_equal(b.m["conv1/weights"], a.m["conv1/weights"])
