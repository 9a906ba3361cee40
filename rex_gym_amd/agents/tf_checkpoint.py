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
