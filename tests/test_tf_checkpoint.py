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


# ---------------------------------------------------------------------------------------------------------------------
# the format halves against INDEPENDENT implementations present in the build container (skipped where absent) --
# Arrow's Snappy codec for the block compression, google.protobuf for BundleEntryProto / BundleHeaderProto / TensorShapeProto
# (messages declared here from tensorflow/core/protobuf/tensor_bundle.proto and tensor_shape.proto: field numbers and wire types as
# published).  Not a TensorFlow-written file -- f1 stays "partial" -- but no longer this module checked against itself.
def _snappy_corpus():
    rng = np.random.default_rng(7)
    yield b""
    yield b"a"
    yield b"a" * 70000                                                   # one long run: overlapping copies, 2-byte offsets
    yield bytes(rng.integers(0, 256, size=100000, dtype=np.uint8))       # incompressible: long literals (60..63 length forms)
    yield bytes(rng.integers(0, 4, size=50000, dtype=np.uint8))          # low entropy: many short copies (1-byte-offset form)
    yield (b"conv%d/weights/Momentum" * 40) % tuple(range(40)) * 30      # what an .index block holds: repeated key prefixes
    blob = bytes(rng.integers(0, 256, size=3000, dtype=np.uint8))
    yield blob + b"\x00" * 80000 + blob                                  # a copy reaching back more than 64 KB (4-byte-offset form)
    for n in (59, 60, 61, 255, 256, 257, 65535, 65536, 65537):           # literal-length form boundaries
        yield bytes(rng.integers(0, 256, size=n, dtype=np.uint8))


def test_snappy_decoder_against_arrows_codec():
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("snappy"):
        pytest.skip("this pyarrow build has no snappy codec")
    tags = set()
    for raw in _snappy_corpus():
        comp = pa.compress(raw, codec="snappy", asbytes=True)            # raw Snappy block format, as LevelDB / TensorFlow tables store it
        assert C.snappy_decompress(comp) == raw
        assert pa.decompress(comp, decompressed_size=len(raw), codec="snappy", asbytes=True) == raw
        pos = C._get_varint(comp, 0)[1]
        while pos < len(comp):                                           # which element kinds Arrow's compressor emitted (coverage, below)
            tag = comp[pos]
            kind = tag & 3
            tags.add(kind if kind else ("lit", min(tag >> 2, 60)))
            if kind == 0:
                ln = tag >> 2
                nb = ln - 59 if ln >= 60 else 0
                ln = int.from_bytes(comp[pos + 1:pos + 1 + nb], "little") if nb else ln
                pos += 1 + nb + ln + 1
            else:
                pos += {1: 2, 2: 3, 3: 5}[kind]
    assert {1, 2} <= tags and ("lit", 60) in tags                        # 1- and 2-byte-offset copies and the long-literal form all occurred
    with pytest.raises(ValueError):
        C.snappy_decompress(pa.compress(bytes(range(200)), codec="snappy", asbytes=True)[:-10])      # a truncated stream is a length mismatch


def test_table_reader_takes_blocks_compressed_by_arrows_codec():
    """a table whose data blocks are Snappy-compressed (type byte 1) by an independent compressor, as TensorFlow's table builder
    writes them when compression helps: read_table must return the same pairs as for the uncompressed table"""
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("snappy"):
        pytest.skip("this pyarrow build has no snappy codec")
    items = sorted([(("conv%d/weights/Momentum" % i).encode(), C.encode_entry(C.DT_FLOAT, (3, 3, 64 + i, 128), 1000 * i, 4 * 9 * (64 + i) * 128, i))
                    for i in range(300)] + [(b"", C.encode_header(1))])
    out, index = bytearray(), []

    def emit(block, compress):
        body = pa.compress(block, codec="snappy", asbytes=True) if compress else block
        off = len(out)
        out.extend(body)
        out.append(1 if compress else 0)
        out.extend(struct.pack("<I", C.mask_crc(C.crc32c(body + bytes([1 if compress else 0])))))
        return off, len(body)
    for i in range(0, len(items), 40):
        chunk = items[i:i + 40]
        off, sz = emit(C._build_block(chunk), compress=True)
        index.append((chunk[-1][0], C._put_varint(off) + C._put_varint(sz)))
    moff, msz = emit(C._build_block([]), compress=False)
    ioff, isz = emit(C._build_block(index, restart_interval=1), compress=True)
    foot = C._put_varint(moff) + C._put_varint(msz) + C._put_varint(ioff) + C._put_varint(isz)
    out.extend(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", C.MAGIC))
    assert C.read_table(bytes(out)) == items == C.read_table(C.write_table(items))
    assert len(out) < len(C.write_table(items))                           # (the blocks really were compressed)


def _bundle_messages():
    """BundleHeaderProto, BundleEntryProto (tensor_bundle.proto), TensorShapeProto (tensor_shape.proto), VersionDef (versions.proto) as
    google.protobuf dynamic messages: field numbers / types as TensorFlow publishes them"""
    pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="drs_test_tensor_bundle.proto", package="drs_test_tf", syntax="proto3")

    def msg(name, fields, parent=None):
        m = (parent.nested_type if parent is not None else fd.message_type).add(name=name)
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = tname
        return m
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    shape = msg("TensorShapeProto", [("dim", 2, F.TYPE_MESSAGE, REP, ".drs_test_tf.TensorShapeProto.Dim"), ("unknown_rank", 3, F.TYPE_BOOL, OPT, None)])
    msg("Dim", [("size", 1, F.TYPE_INT64, OPT, None), ("name", 2, F.TYPE_STRING, OPT, None)], parent=shape)
    msg("VersionDef", [("producer", 1, F.TYPE_INT32, OPT, None), ("min_consumer", 2, F.TYPE_INT32, OPT, None), ("bad_consumers", 3, F.TYPE_INT32, REP, None)])
    msg("BundleHeaderProto", [("num_shards", 1, F.TYPE_INT32, OPT, None), ("endianness", 2, F.TYPE_INT32, OPT, None),      # (an enum on the wire is a varint)
                              ("version", 3, F.TYPE_MESSAGE, OPT, ".drs_test_tf.VersionDef")])
    msg("BundleEntryProto", [("dtype", 1, F.TYPE_INT32, OPT, None), ("shape", 2, F.TYPE_MESSAGE, OPT, ".drs_test_tf.TensorShapeProto"),
                             ("shard_id", 3, F.TYPE_INT32, OPT, None), ("offset", 4, F.TYPE_INT64, OPT, None), ("size", 5, F.TYPE_INT64, OPT, None),
                             ("crc32c", 6, F.TYPE_FIXED32, OPT, None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("drs_test_tf." + n))
    return get("BundleHeaderProto"), get("BundleEntryProto")


ENTRY_CASES = [(C.DT_FLOAT, (5, 5, 3, 64), 0, 19200, 0xDEADBEEF), (C.DT_FLOAT, (64,), 19200, 256, 1), (C.DT_INT32, (), 123456789012, 4, 0xFFFFFFFF),
               (C.DT_INT64, (2, 3), 1 << 40, 48, 0x80000000), (C.DT_DOUBLE, (1, 1, 256, 6), 7, 12288, 0)]


def test_entry_and_header_protos_against_google_protobuf():
    Header, Entry = _bundle_messages()
    for dtype, shape, offset, size, crc in ENTRY_CASES:
        # ours -> protobuf's parser
        m = Entry()
        m.ParseFromString(C.encode_entry(dtype, shape, offset, size, crc))
        assert (m.dtype, [d.size for d in m.shape.dim], m.shard_id, m.offset, m.size, m.crc32c) == (dtype, list(shape), 0, offset, size, crc)
        assert not m.shape.unknown_rank
        # protobuf's serialiser -> ours (shard_id set as well: TensorFlow writes it for sharded savers)
        m2 = Entry(dtype=dtype, shard_id=3, offset=offset, size=size, crc32c=crc)
        for d in shape:
            m2.shape.dim.add(size=d)
        if not shape:
            m2.shape.SetInParent()
        e = C.decode_entry(m2.SerializeToString())
        assert e == dict(dtype=dtype, shape=list(shape), shard_id=3, offset=offset, size=size, crc32c=crc)
        # and byte for byte where the field sets coincide (proto3 omits zero scalars exactly as encode_entry does)
        m3 = Entry(dtype=dtype, offset=offset, size=size, crc32c=crc)
        for d in shape:
            m3.shape.dim.add(size=d)
        m3.shape.SetInParent()
        assert m3.SerializeToString(deterministic=True) == C.encode_entry(dtype, shape, offset, size, crc)
    h = Header()
    h.ParseFromString(C.encode_header(1))
    assert (h.num_shards, h.endianness, h.version.producer, h.version.min_consumer) == (1, 0, 1, 0)
    assert Header(num_shards=1, version=dict(producer=1)).SerializeToString(deterministic=True) == C.encode_header(1)


def test_written_index_parses_with_google_protobuf(tmp_path):
    """write_bundle -> .index -> every value of the table through protobuf's own parser: offsets tile the data file, sizes and shapes
    are the arrays', the stored CRC is the masked CRC-32C of the tensor's bytes"""
    Header, Entry = _bundle_messages()
    rng = np.random.default_rng(3)
    t = {"conv1/weights": rng.normal(size=(5, 5, 3, 64)).astype(np.float32), "conv1/biases": np.full(64, 0.1, np.float32),
         "conv_classifier/weights": rng.normal(size=(1, 1, 256, 6)).astype(np.float32), "main_global_step": np.array(7, dtype=np.int32),
         "conv1/weights/Momentum": np.zeros((5, 5, 3, 64), np.float32)}
    prefix = str(tmp_path / "model-7")
    C.write_bundle(prefix, t)
    pairs = C.read_table(open(prefix + ".index", "rb").read())
    assert [k for k, _ in pairs] == sorted([b""] + [n.encode() for n in t])
    h = Header()
    h.ParseFromString(pairs[0][1])
    assert h.num_shards == 1 and h.endianness == 0 and h.version.producer == 1
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    end = 0
    for k, v in pairs[1:]:
        m = Entry()
        m.ParseFromString(v)
        a = t[k.decode()]
        assert m.dtype == {np.dtype(np.float32): 1, np.dtype(np.int32): 3}[a.dtype] and tuple(d.size for d in m.shape.dim) == a.shape
        assert m.shard_id == 0 and m.offset == end and m.size == a.nbytes
        assert C.unmask_crc(m.crc32c) == C.crc32c(data[m.offset:m.offset + m.size]) and data[m.offset:m.offset + m.size] == a.tobytes()
        end += m.size
    assert end == len(data)
