"""TensorFlow V2 checkpoint ("tensor bundle") import / export without TensorFlow  (SURVEY.md 8f-1).

The reference persists its model with `tf.train.Saver` (/root/reference/isprs_dilated_random.py:1693-1695, 1798, 1835;
restore :1715): `<output_path>model-<step>.index` + `.data-00000-of-00001`.  Variables (TF scopes of `_conv_layer`
isprs:705-707, contrib batch_norm under the same scope, `MomentumOptimizer` slots, the step counter):

    <scope>/weights  <scope>/biases  <scope>/moving_mean  <scope>/moving_variance
    <scope>/weights/Momentum  <scope>/biases/Momentum  main_global_step (isprs:1685) | global_step (coffee:1191)

File formats, restated from TensorFlow's published sources (tensorflow/core/util/tensor_bundle/tensor_bundle.cc,
tensorflow/core/protobuf/tensor_bundle.proto, tensorflow/core/lib/io/table_format.txt = the LevelDB table format):
  * `.index`  an immutable sorted string table: prefix-compressed data blocks (each followed by a 1-byte compression type
    and a masked CRC-32C), an index block, an empty metaindex block and a 48-byte footer ending in the magic
    0xdb4775248b80fb57; key "" -> BundleHeaderProto, key <variable name> -> BundleEntryProto
    (dtype, shape, shard_id, offset, size, crc32c of the bytes);
  * `.data-00000-of-00001`  the tensors' raw little-endian bytes back to back.
The reader accepts Snappy-compressed blocks (TensorFlow's table builder compresses when it helps); the writer emits
uncompressed blocks, which every TensorFlow reader accepts.

STATUS: there is no TensorFlow in this environment, so this module is not verified against a TensorFlow-written file.  It is
verified by round trip, by the formats' own checksums and known answers, and (since round 6) half by half against independent
implementations present in the build container: Snappy against Arrow's codec (pyarrow), the BundleEntryProto / BundleHeaderProto /
TensorShapeProto encodings against google.protobuf, the table reader on blocks compressed by Arrow (tests/test_tf_checkpoint.py).
"""
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_NP_OF = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_INT64: np.int64}
_DT_OF = {np.dtype(v): k for k, v in _NP_OF.items()}

# ------------------------------------------------------------------------------------------- CRC-32C (Castagnoli)
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in bytes(data):
        crc = tab[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _zeros_operator(nbytes):
    """32x32 GF(2) matrix (as 32 column words) that advances a finalised CRC-32C over `nbytes` zero bytes
    (zlib's crc32_combine construction with the Castagnoli polynomial)."""
    odd = [0x82F63B78] + [1 << n for n in range(31)]              # one zero bit
    even = [_gf2_times(odd, odd[n]) for n in range(32)]           # two
    odd = [_gf2_times(even, even[n]) for n in range(32)]          # four
    res = [1 << n for n in range(32)]
    while nbytes:
        even = [_gf2_times(odd, odd[n]) for n in range(32)]       # 8 bits on the first pass, then x4 per pass
        if nbytes & 1:
            res = [_gf2_times(even, res[n]) for n in range(32)]
        nbytes >>= 1
        if not nbytes:
            break
        odd = [_gf2_times(even, even[n]) for n in range(32)]
        if nbytes & 1:
            res = [_gf2_times(odd, res[n]) for n in range(32)]
        nbytes >>= 1
    return res


def crc32c_fast(data, lanes=2048):
    """CRC-32C of a large buffer: `lanes` equal chunks are hashed side by side with numpy table lookups, then folded
    left to right with the zero-advance operator (a CRC is linear over GF(2))."""
    data = bytes(data)
    n = len(data)
    L = n // lanes
    if L < 64:
        return crc32c(data)
    body = np.frombuffer(data, dtype=np.uint8, count=L * lanes).reshape(lanes, L)
    tab = np.asarray(_CRC_TABLE, dtype=np.uint32)
    reg = np.full(lanes, 0xFFFFFFFF, dtype=np.uint32)
    for j in range(L):
        reg = tab[(reg ^ body[:, j]) & 0xFF] ^ (reg >> np.uint32(8))
    parts = (reg ^ np.uint32(0xFFFFFFFF)).tolist()
    op = _zeros_operator(L)
    total = parts[0]
    for c in parts[1:]:
        total = _gf2_times(op, total) ^ c
    tail = data[L * lanes:]
    if tail:
        total = _gf2_times(_zeros_operator(len(tail)), total) ^ crc32c(tail)
    return total


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - 0xa282ead8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------- varints / protobuf wire format
def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _pb_fields(buf):
    """yield (field number, wire type, value) of one protobuf message."""
    pos = 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _pb_varint_field(f, v):
    return _put_varint(f << 3) + _put_varint(v)


def _pb_bytes_field(f, b):
    return _put_varint((f << 3) | 2) + _put_varint(len(b)) + bytes(b)


def encode_entry(dtype, shape, offset, size, crc):
    """BundleEntryProto{dtype=1, shape=2{dim=2{size=1}}, shard_id=3, offset=4, size=5, crc32c=6 (fixed32)}."""
    shp = b"".join(_pb_bytes_field(2, _pb_varint_field(1, d)) for d in shape)
    msg = _pb_varint_field(1, dtype) + _pb_bytes_field(2, shp)
    if offset:
        msg += _pb_varint_field(4, offset)
    msg += _pb_varint_field(5, size)
    if crc:                                  # (proto3 omits a zero fixed32, as TensorFlow's serialiser does)
        msg += _put_varint((6 << 3) | 5) + struct.pack("<I", crc)
    return msg


def decode_entry(buf):
    # proto3: a scalar field that holds its default (0) is not on the wire -- a (masked) crc32c of 0 is an ABSENT field, and TensorFlow's
    # reader checks the tensor against entry.crc32c() == 0 then; so does read_bundle (found against google.protobuf's serialiser, r06)
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=0)
    for f, wt, v in _pb_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:
                    d = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            d = v3
                    e["shape"].append(d)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
    return e


def encode_header(num_shards=1):
    """BundleHeaderProto{num_shards=1, endianness=2 (LITTLE = 0, default -> omitted), version=3{producer=1}}."""
    return _pb_varint_field(1, num_shards) + _pb_bytes_field(3, _pb_varint_field(1, 1))


# ------------------------------------------------------------------------------------------- Snappy (decode only)
def snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        for _ in range(ln):                      # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch")
    return bytes(out)


# ------------------------------------------------------------------------------------------- sorted string table
def _read_block(buf, offset, size, verify=True):
    raw = buf[offset:offset + size]
    ctype = buf[offset + size]
    stored = struct.unpack("<I", buf[offset + size + 1:offset + size + 5])[0]
    if verify and unmask_crc(stored) != crc32c(buf[offset:offset + size + 1]):
        raise ValueError("table block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        raw = snappy_decompress(raw)
    elif ctype != 0:
        raise ValueError("unknown block compression %d" % ctype)
    return raw


def _block_entries(block):
    nrestart = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        unshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(buf, verify=True):
    """all (key, value) pairs of an SSTable held in `buf`."""
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != MAGIC:
        raise ValueError("not a TensorFlow/LevelDB table (bad magic)")
    foot = buf[-48:]
    _, p = _get_varint(foot, 0)            # metaindex handle
    _, p = _get_varint(foot, p)
    ioff, p = _get_varint(foot, p)
    isize, p = _get_varint(foot, p)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0)
        bsize, q = _get_varint(handle, q)
        out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


def _build_block(items, restart_interval=16):
    body, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(body))
        else:
            m = min(len(k), len(last))
            while shared < m and k[shared] == last[shared]:
                shared += 1
        body += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def write_table(items, block_size=4096):
    """SSTable bytes for sorted (key, value) pairs; uncompressed blocks."""
    items = sorted(items)
    out = bytearray()
    index = []

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)                                                  # kNoCompression
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return off, len(block)

    cur, cur_bytes = [], 0
    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 8
        if cur_bytes >= block_size:
            off, sz = emit(_build_block(cur))
            index.append((cur[-1][0], _put_varint(off) + _put_varint(sz)))
            cur, cur_bytes = [], 0
    if cur:
        off, sz = emit(_build_block(cur))
        index.append((cur[-1][0], _put_varint(off) + _put_varint(sz)))
    moff, msz = emit(_build_block([]))
    ioff, isz = emit(_build_block(index, restart_interval=1))
    foot = _put_varint(moff) + _put_varint(msz) + _put_varint(ioff) + _put_varint(isz)
    out.extend(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC))
    return bytes(out)


# ------------------------------------------------------------------------------------------- bundles
def write_bundle(prefix, tensors):
    """tensors: name -> numpy array (float32 / float64 / int32 / int64).  Writes <prefix>.index and .data-00000-of-00001."""
    items = [(b"", encode_header(1))]
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        for name in sorted(tensors):
            a = np.asarray(tensors[name])
            if not a.flags.c_contiguous:                    # (np.ascontiguousarray would turn a scalar into shape (1,))
                a = a.copy(order="C")
            if a.dtype not in _DT_OF:
                raise ValueError("unsupported dtype %s for %s" % (a.dtype, name))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            data.write(raw)
            items.append((name.encode(), encode_entry(_DT_OF[a.dtype], a.shape, offset, len(raw), mask_crc(crc32c_fast(raw)))))
            offset += len(raw)
    with open(prefix + ".index", "wb") as f:
        f.write(write_table(items))


def read_bundle(prefix, verify=True):
    """name -> numpy array for every tensor of a V2 checkpoint."""
    with open(prefix + ".index", "rb") as f:
        entries = dict(read_table(f.read(), verify))
    header = entries.pop(b"", None)
    nshards = 1
    if header is not None:
        for fno, _, v in _pb_fields(header):
            if fno == 1:
                nshards = v
            if fno == 2 and v != 0:
                raise ValueError("big-endian bundles are not supported")
    shards = {}
    out = {}
    for key, val in entries.items():
        e = decode_entry(val)
        if e["dtype"] not in _NP_OF:
            continue                                        # e.g. string tensors of a Saver; not model state
        sid = e["shard_id"]
        if sid not in shards:
            with open("%s.data-%05d-of-%05d" % (prefix, sid, nshards), "rb") as f:
                shards[sid] = f.read()
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if verify and unmask_crc(e["crc32c"]) != crc32c_fast(raw):
            raise ValueError("tensor checksum mismatch for " + key.decode())
        out[key.decode()] = np.frombuffer(raw, dtype=np.dtype(_NP_OF[e["dtype"]]).newbyteorder("<")).reshape(tuple(e["shape"])).copy()
    return out


# ------------------------------------------------------------------------------------------- net <-> checkpoint
def save_tf_checkpoint(net, prefix, global_step_name="main_global_step"):
    """What `saver.save(sess, output_path + 'model', global_step=step)` stores (isprs:1798), readable by tf.train.Saver."""
    t = {}
    for n in net.variable_names():
        t[n] = net.get_variable(n)
    for n in net.plan.offsets:
        t[n + "/Momentum"] = net.get_variable(n, "Momentum")
    t[global_step_name] = np.array(net.global_step, dtype=np.int32)          # tf.Variable(0) is int32
    write_bundle(prefix, t)


def load_tf_checkpoint(net, prefix, strict=True):
    """`saver_restore.restore(sess, former_model_path)` (isprs:1715) from a TensorFlow-written (or save_tf_checkpoint) bundle."""
    t = read_bundle(prefix)
    missing = [n for n in net.variable_names() if n not in t]
    if missing and strict:
        raise KeyError("checkpoint lacks " + ", ".join(missing[:5]) + (" ..." if len(missing) > 5 else ""))
    for n in net.variable_names():
        if n in t:
            net.set_variable(n, t[n])
    for n in net.plan.offsets:
        if n + "/Momentum" in t:
            net.set_variable(n, t[n + "/Momentum"], "Momentum")
    for gs in ("main_global_step", "global_step"):
        if gs in t:
            net.global_step = int(t[gs])
    return sorted(set(t) - set(net.variable_names()) - {n + "/Momentum" for n in net.plan.offsets})
