"""Reader for TensorFlow-1 checkpoints (the "tensor bundle" written by tf.train.Saver), without TensorFlow.

The reference ships its trained policies as `model.ckpt-N.{index,data-00000-of-00001,meta}` under
`rex_gym/policies/<env>/<signal>/` and restores them with `tf.train.Saver` inside `SimplePPOPolicy`
(agents/ppo/simple_ppo_agent.py:29-63, playground/policy_player.py:22-56).  TensorFlow is not part of this
stack, so the two files that matter are read directly:

* `*.index` is an SSTable in the LevelDB table format (blocks of prefix-compressed key/value entries, a trailing
  index block, a 48-byte footer with magic 0xdb4775248b80fb57).  Keys are variable names; values are serialized
  `BundleEntryProto` messages (dtype, shape, shard, byte offset and size, masked crc32c).  The empty key holds the
  `BundleHeaderProto`.
* `*.data-SSSSS-of-NNNNN` holds the raw little-endian tensor bytes at those offsets.

Only what a checkpoint of dense float/int variables needs is implemented (no string tensors, no slices); snappy
block compression -- not used by TensorFlow's bundle writer, but legal in the table format -- is decoded as well.
Every tensor's crc32c is verified on load.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}


class CheckpointError(ValueError):
    pass


def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise CheckpointError("varint too long")


def _snappy(src):
    """Raw snappy block format: varint uncompressed length, then literal / copy elements."""
    n, pos = _varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            length = tag >> 2
            if length >= 60:
                nb = length - 59
                length = int.from_bytes(src[pos:pos + nb], "little")
                pos += nb
            length += 1
            out += src[pos:pos + length]
            pos += length
            continue
        if kind == 1:
            length = ((tag >> 2) & 7) + 4
            offset = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            length = (tag >> 2) + 1
            offset = int.from_bytes(src[pos:pos + 2], "little")
            pos += 2
        else:
            length = (tag >> 2) + 1
            offset = int.from_bytes(src[pos:pos + 4], "little")
            pos += 4
        if offset == 0 or offset > len(out):
            raise CheckpointError("bad snappy copy offset")
        for _ in range(length):
            out.append(out[-offset])
    if len(out) != n:
        raise CheckpointError("snappy length mismatch")
    return bytes(out)


def _make_crc_table():
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)
    return np.asarray(table, np.uint32)


_CRC_TABLE = _make_crc_table()


def crc32c(data):
    """CRC-32C (Castagnoli), slicing one byte at a time over numpy lookups in chunks."""
    crc = 0xFFFFFFFF
    table = _CRC_TABLE.tolist()
    for b in memoryview(data).cast("B"):
        crc = table[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    """The masking LevelDB / TensorFlow store: rotate right by 15 and add a constant."""
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _block(buf, offset, size):
    raw = buf[offset:offset + size]
    if len(raw) != size or offset + size + 5 > len(buf):
        raise CheckpointError("block handle out of range")
    kind = buf[offset + size]
    stored, = struct.unpack_from("<I", buf, offset + size + 1)
    if masked_crc32c(buf[offset:offset + size + 1]) != stored:
        raise CheckpointError("block checksum mismatch")
    if kind == 0:
        return raw
    if kind == 1:
        return _snappy(raw)
    raise CheckpointError(f"unknown block compression {kind}")


def _entries(block):
    """(key, value) pairs of one table block."""
    if len(block) < 4:
        raise CheckpointError("block too small")
    num_restarts, = struct.unpack_from("<I", block, len(block) - 4)
    limit = len(block) - 4 - 4 * num_restarts
    if limit < 0:
        raise CheckpointError("bad restart array")
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        value_len, pos = _varint(block, pos)
        if shared > len(key) or pos + non_shared + value_len > limit:
            raise CheckpointError("corrupt block entry")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + value_len]
        pos += value_len


def _fields(msg):
    """(field number, wire type, value) triples of one protobuf message."""
    pos = 0
    while pos < len(msg):
        tag, pos = _varint(msg, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            value, pos = _varint(msg, pos)
        elif wire == 1:
            value = msg[pos:pos + 8]
            pos += 8
        elif wire == 2:
            n, pos = _varint(msg, pos)
            value = msg[pos:pos + n]
            pos += n
        elif wire == 5:
            value = msg[pos:pos + 4]
            pos += 4
        else:
            raise CheckpointError(f"unsupported protobuf wire type {wire}")
        yield field, wire, value


def _entry(msg):
    """BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto)."""
    e = dict(dtype=0, shape=[], shard=0, offset=0, size=0, crc=None, sliced=False)
    for field, _, value in _fields(msg):
        if field == 1:
            e["dtype"] = value
        elif field == 2:                                  # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
            for f2, _, dim in _fields(value):
                if f2 == 2:
                    size = 0
                    for f3, _, v in _fields(dim):
                        if f3 == 1:
                            size = v
                    e["shape"].append(size)
        elif field == 3:
            e["shard"] = value
        elif field == 4:
            e["offset"] = value
        elif field == 5:
            e["size"] = value
        elif field == 6:
            e["crc"], = struct.unpack("<I", value)
        elif field == 7:
            e["sliced"] = True
    return e


class Checkpoint:
    """`Checkpoint(prefix)` with prefix = '.../model.ckpt-2000000': names, shapes and tensors of a TF-1 checkpoint."""

    def __init__(self, prefix):
        self.prefix = prefix
        with open(prefix + ".index", "rb") as f:
            buf = f.read()
        if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
            raise CheckpointError(f"{prefix}.index is not a TensorFlow checkpoint index (bad magic)")
        footer = buf[-48:]
        pos = 0
        _, pos = _varint(footer, pos)                      # metaindex handle (unused)
        _, pos = _varint(footer, pos)
        index_offset, pos = _varint(footer, pos)
        index_size, pos = _varint(footer, pos)
        self.entries = {}
        self.num_shards = 1
        for _, handle in _entries(_block(buf, index_offset, index_size)):
            off, p = _varint(handle, 0)
            size, _ = _varint(handle, p)
            for key, value in _entries(_block(buf, off, size)):
                if key == b"":                             # BundleHeaderProto: num_shards = 1, endianness = 2
                    for field, _, v in _fields(value):
                        if field == 1:
                            self.num_shards = v
                        elif field == 2 and v != 0:
                            raise CheckpointError("big-endian checkpoints are not supported")
                    continue
                self.entries[key.decode()] = _entry(value)
        self._shards = {}

    def names(self):
        return sorted(self.entries)

    def shape(self, name):
        return tuple(self.entries[name]["shape"])

    def _shard(self, k):
        if k not in self._shards:
            path = f"{self.prefix}.data-{k:05d}-of-{self.num_shards:05d}"
            if not os.path.exists(path):
                raise CheckpointError(f"checkpoint data shard missing: {path}")
            self._shards[k] = np.memmap(path, dtype=np.uint8, mode="r")
        return self._shards[k]

    def tensor(self, name, verify=True):
        if name not in self.entries:
            raise KeyError(f"{name!r} not in checkpoint {self.prefix}; have {self.names()[:8]} ...")
        e = self.entries[name]
        if e["sliced"] or e["dtype"] not in _DTYPES:
            raise CheckpointError(f"{name}: unsupported entry (dtype {e['dtype']}, sliced {e['sliced']})")
        raw = bytes(self._shard(e["shard"])[e["offset"]:e["offset"] + e["size"]])
        dtype = np.dtype(_DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if len(raw) != e["size"] or count * dtype.itemsize != e["size"]:
            raise CheckpointError(f"{name}: size {e['size']} does not match shape {e['shape']} of {dtype}")
        if verify and e["crc"] is not None and masked_crc32c(raw) != e["crc"]:
            raise CheckpointError(f"{name}: tensor checksum mismatch")
        return np.frombuffer(raw, dtype.newbyteorder("<")).astype(dtype).reshape(tuple(e["shape"]))


# ---------------------------------------------------------------------------------------------------------------------
# Writer: the same two files, so that a policy trained here goes back to the reference's tooling (`rex-gym policy`,
# SimplePPOPolicy's tf.train.Saver.restore: agents/ppo/simple_ppo_agent.py:62-63, agents/scripts/utility.py:79-95).
# Byte layout as TensorFlow's BundleWriter produces it for the reference's shipped checkpoints (tests/test_agents_policy_player.py
# rewrites rex_gym/policies/walk/ik/model.ckpt-2000000 and compares both files byte for byte): tensors in key order, back to
# back, in one data shard; one uncompressed table block of prefix-compressed entries with a restart point every 16 keys
# (a new block once one reaches BLOCK_SIZE), an empty metaindex block, an index block of (shortest separator, block handle)
# and the 48-byte footer.
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}
BLOCK_SIZE = 262144          # tensorflow/core/lib/io/table_options.h
RESTART_INTERVAL = 16


def _put_varint(out, v):
    v = int(v)
    if v < 0:
        raise CheckpointError("negative varint")
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)


def _entry_proto(dtype_code, shape, offset, size, crc):
    """BundleEntryProto, fields in number order; zero-valued scalars are not written (proto3), the shape message always is."""
    m = bytearray()
    m.append(0x08)
    _put_varint(m, dtype_code)
    dims = bytearray()
    for d in shape:
        dim = bytearray()
        if d:
            dim.append(0x08)
            _put_varint(dim, d)
        dims.append(0x12)
        _put_varint(dims, len(dim))
        dims += dim
    m.append(0x12)
    _put_varint(m, len(dims))
    m += dims
    if offset:
        m.append(0x20)
        _put_varint(m, offset)
    if size:
        m.append(0x28)
        _put_varint(m, size)
    m.append(0x35)
    m += struct.pack("<I", crc)
    return bytes(m)


class _BlockBuilder:
    def __init__(self, restart_interval=RESTART_INTERVAL):
        self.buf, self.restarts, self.count, self.last, self.interval = bytearray(), [0], 0, b"", restart_interval

    def add(self, key, value):
        shared = 0
        if self.count % self.interval == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        _put_varint(self.buf, shared)
        _put_varint(self.buf, len(key) - shared)
        _put_varint(self.buf, len(value))
        self.buf += key[shared:] + value
        self.last, self.count = key, self.count + 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + struct.pack(f"<{len(self.restarts)}I", *self.restarts) + struct.pack("<I", len(self.restarts))


def _separator(a, b):
    """LevelDB's BytewiseComparator::FindShortestSeparator: a short key k with a <= k < b."""
    n = min(len(a), len(b))
    d = 0
    while d < n and a[d] == b[d]:
        d += 1
    if d < n and a[d] < 0xFF and a[d] + 1 < b[d]:
        return a[:d] + bytes([a[d] + 1])
    return a


def _successor(a):
    """FindShortSuccessor: the key of the last block's index entry."""
    for i, c in enumerate(a):
        if c != 0xFF:
            return a[:i] + bytes([c + 1])
    return a


def write_checkpoint(prefix, tensors, block_size=BLOCK_SIZE):
    """Write `tensors` (name -> array; float32 / int32 / ... of _DTYPES) as `<prefix>.index` + `<prefix>.data-00000-of-00001`."""
    names = sorted(tensors, key=lambda s: s.encode())
    if any(not n for n in names):
        raise CheckpointError("empty variable name")
    data = bytearray()
    entries = [(b"", b"\x08\x01\x1a\x02\x08\x01")]            # BundleHeaderProto: num_shards 1, little endian (0, unwritten), version { producer 1 }
    for name in names:
        a = np.asarray(tensors[name])
        if a.dtype not in _DTYPE_CODES:
            raise CheckpointError(f"{name}: dtype {a.dtype} has no TensorFlow code here")
        raw = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"))).tobytes()
        entries.append((name.encode(), _entry_proto(_DTYPE_CODES[a.dtype], a.shape, len(data), len(raw), masked_crc32c(raw))))
        data += raw
    out = bytearray()

    def emit(block):
        handle = bytearray()
        _put_varint(handle, len(out))
        _put_varint(handle, len(block))
        out.extend(block + b"\x00" + struct.pack("<I", masked_crc32c(block + b"\x00")))
        return bytes(handle)

    index, blk, pending = _BlockBuilder(1), _BlockBuilder(), None      # (index blocks restart at every key: table_builder.cc)
    for key, value in entries:
        if pending is not None:
            index.add(_separator(pending[0], key), pending[1])
            pending = None
        blk.add(key, value)
        if blk.size() >= block_size:
            pending = (key, emit(blk.finish()))
            blk = _BlockBuilder()
    if blk.count:
        pending = (blk.last, emit(blk.finish()))
    if pending is not None:
        index.add(_successor(pending[0]), pending[1])
    meta = emit(_BlockBuilder().finish())
    idx = emit(index.finish())
    footer = meta + idx
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    return prefix
