"""TensorFlow "bundle v2" checkpoint reader/writer without TensorFlow.

The reference restores its variables *by name* with ``tf.train.Saver.restore``
(reference ``borealisflows/NoiseFlowWrapper.py:67,77`` and
``train_noise_flow.py:322-328,358-367``).  TensorFlow is not available on the
MI355X box, so this module parses the two on-disk files directly:

``<prefix>.index``
    a LevelDB-style sorted string table: data blocks of prefix-compressed
    ``(shared, non_shared, value_len, key_delta, value)`` varint entries, each
    block followed by a restart array and a 5-byte trailer (compression type +
    masked crc32c); a 48-byte footer holds the metaindex / index block handles
    and the magic ``0xdb4775248b80fb57``.  Key ``""`` maps to a
    ``BundleHeaderProto``; every other key is a variable name mapping to a
    ``BundleEntryProto`` {1: dtype, 2: TensorShapeProto, 3: shard_id,
    4: offset, 5: size, 6: crc32c}.
``<prefix>.data-00000-of-00001``
    raw little-endian row-major tensor bytes at ``offset``.

Only what the Noise Flow checkpoints use is supported: uncompressed blocks,
a single shard, float32 / int32 / int64 tensors.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Tuple

import numpy as np

_TABLE_MAGIC = 0xDB4775248B80FB57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8")}
_DTYPE_CODES = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9}


# ----------------------------------------------------------------------------
# crc32c (Castagnoli) + the LevelDB/TF "mask"
# ----------------------------------------------------------------------------
def _make_crc_table() -> List[int]:
    poly = 0x82F63B78
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        table.append(c)
    return table


_CRC_TABLE = _make_crc_table()


def crc32c(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in data:
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ----------------------------------------------------------------------------
# varint / protobuf helpers
# ----------------------------------------------------------------------------
def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _write_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf: bytes) -> Dict[int, list]:
    """Minimal protobuf wire parser: {field: [values]} (varint→int, len→bytes,
    fixed32/64→int)."""
    fields: Dict[int, list] = {}
    pos = 0
    n = len(buf)
    while pos < n:
        tag, pos = _read_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            val = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        fields.setdefault(field, []).append(val)
    return fields


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for d in _parse_proto(buf).get(2, []):
        size = _parse_proto(d).get(1, [0])[0]
        dims.append(int(size))
    return tuple(dims)


# ----------------------------------------------------------------------------
# table reader
# ----------------------------------------------------------------------------
def _read_block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    block = buf[offset:offset + size]
    trailer = buf[offset + size:offset + size + 5]
    if len(block) != size or len(trailer) != 5:
        raise ValueError("truncated table block")
    if trailer[0] != 0:
        raise ValueError("compressed table blocks are not supported (type %d)" % trailer[0])
    if verify:
        want = struct.unpack("<I", trailer[1:5])[0]
        got = masked_crc32c(block + trailer[:1])
        if want != got:
            raise ValueError("table block crc mismatch")
    return block


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos = 0
    key = b""
    out = []
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(path: str, verify_crc: bool = True) -> Dict[str, dict]:
    """Parse ``<prefix>.index`` → ``{name: {dtype, shape, shard, offset, size, crc}}``."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 48:
        raise ValueError("index file too small")
    footer = buf[-48:]
    if struct.unpack("<Q", footer[40:])[0] != _TABLE_MAGIC:
        raise ValueError("bad table magic in %s" % path)
    pos = 0
    _, pos = _read_varint(footer, pos)  # metaindex offset
    _, pos = _read_varint(footer, pos)  # metaindex size
    idx_off, pos = _read_varint(footer, pos)
    idx_size, pos = _read_varint(footer, pos)
    entries: Dict[str, dict] = {}
    for _, handle in _block_entries(_read_block(buf, idx_off, idx_size, verify_crc)):
        boff, p = _read_varint(handle, 0)
        bsize, p = _read_varint(handle, p)
        for key, val in _block_entries(_read_block(buf, boff, bsize, verify_crc)):
            if key == b"":
                continue  # BundleHeaderProto
            f = _parse_proto(val)
            entries[key.decode("utf-8")] = {
                "dtype": f.get(1, [0])[0],
                "shape": _parse_shape(f[2][0]) if 2 in f else (),
                "shard": f.get(3, [0])[0],
                "offset": f.get(4, [0])[0],
                "size": f.get(5, [0])[0],
                "crc": f.get(6, [None])[0],
            }
    return entries


def load_checkpoint(prefix: str, verify_crc: bool = True) -> Dict[str, np.ndarray]:
    """Read every tensor of a TF bundle checkpoint: ``{variable_name: ndarray}``.

    ``prefix`` is what the reference passes to ``Saver.restore``
    (e.g. ``models/NoiseFlow/ckpt/model.ckpt.best``).
    """
    index = read_index(prefix + ".index", verify_crc)
    shards = sorted({e["shard"] for e in index.values()})
    if shards not in ([], [0]):
        raise ValueError("multi-shard checkpoints are not supported")
    data_path = prefix + ".data-00000-of-00001"
    with open(data_path, "rb") as f:
        data = f.read()
    out: Dict[str, np.ndarray] = {}
    for name, e in index.items():
        if e["dtype"] not in _DTYPES:
            raise ValueError("unsupported dtype %d for %s" % (e["dtype"], name))
        dt = _DTYPES[e["dtype"]]
        raw = data[e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError("truncated data for %s" % name)
        n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if n * dt.itemsize != e["size"]:
            raise ValueError("size/shape mismatch for %s" % name)
        if verify_crc and e["crc"] is not None and masked_crc32c(raw) != e["crc"]:
            raise ValueError("tensor crc mismatch for %s" % name)
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
    return out


# ----------------------------------------------------------------------------
# writer (same format; lets downstream TF tools restore what this repo saves)
# ----------------------------------------------------------------------------
def _proto_varint_field(field: int, v: int) -> bytes:
    return _write_varint((field << 3) | 0) + _write_varint(v)


def _proto_bytes_field(field: int, b: bytes) -> bytes:
    return _write_varint((field << 3) | 2) + _write_varint(len(b)) + b


def _build_block(items: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out = bytearray()
    restarts = []
    last = b""
    for i, (k, v) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            m = min(len(last), len(k))
            while shared < m and last[shared] == k[shared]:
                shared += 1
        out += _write_varint(shared) + _write_varint(len(k) - shared) + _write_varint(len(v))
        out += k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def save_checkpoint(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write ``tensors`` as a single-shard TF bundle (``.index`` + ``.data-…``)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    data = bytearray()
    items: List[Tuple[bytes, bytes]] = []
    header = _proto_varint_field(1, 1) + _proto_bytes_field(3, _proto_varint_field(1, 1))
    items.append((b"", header))
    for name in names:
        arr = np.asarray(tensors[name])  # (ascontiguousarray would promote 0-d to 1-d)
        if np.dtype(arr.dtype.name) not in _DTYPE_CODES:
            raise ValueError("unsupported dtype %s for %s" % (arr.dtype, name))
        raw = arr.astype(arr.dtype.newbyteorder("<"), order="C", copy=False).tobytes()
        shape = b"".join(_proto_bytes_field(2, _proto_varint_field(1, int(d))) for d in arr.shape)
        entry = _proto_varint_field(1, _DTYPE_CODES[np.dtype(arr.dtype.name)])
        entry += _proto_bytes_field(2, shape)
        if len(data):
            entry += _proto_varint_field(4, len(data))
        entry += _proto_varint_field(5, len(raw))
        entry += _write_varint((6 << 3) | 5) + struct.pack("<I", masked_crc32c(raw))
        items.append((name.encode("utf-8"), entry))
        data += raw
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))

    def emit(buf: bytearray, block: bytes) -> Tuple[int, int]:
        off = len(buf)
        buf += block
        buf += b"\x00" + struct.pack("<I", masked_crc32c(block + b"\x00"))
        return off, len(block)

    table = bytearray()
    d_off, d_size = emit(table, _build_block(items))
    m_off, m_size = emit(table, _build_block([]))
    handle = _write_varint(d_off) + _write_varint(d_size)
    # index key: any key >= the last key of the data block
    i_off, i_size = emit(table, _build_block([(items[-1][0] + b"\xff", handle)], 1))
    footer = _write_varint(m_off) + _write_varint(m_size) + _write_varint(i_off) + _write_varint(i_size)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC)
    table += footer
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(table))
