"""TensorFlow checkpoint ("tensor bundle", the V2 format tf.train.Saver writes) reader / writer in pure Python.

The reference saves and restores `<train_dir>/<ModelName>-<global_step>.{index,data-00000-of-00001}` through
tf.train.Saver / pywrap_tensorflow.NewCheckpointReader (helper/trainer.py:408-410, common/model_loader.py:87-165).
TensorFlow is not importable here, so the container format is restated from its published layout:

  <prefix>.index                an SSTable (the LevelDB table format): key = variable name, value = a serialised
                                BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}; the empty key holds the
                                BundleHeaderProto {num_shards, endianness, version}.  Blocks of prefix-compressed entries
                                with restart points, a 5-byte trailer (compression type + masked CRC-32C) per block, an index
                                block, an (empty) metaindex block and a 48-byte footer ending in the table magic.
  <prefix>.data-00000-of-00001  the tensors' raw little-endian bytes back to back, in key order.
  checkpoint                    text CheckpointState: `model_checkpoint_path` + `all_model_checkpoint_paths`
                                (what tf.train.latest_checkpoint reads).

Parity note: written from the format's specification, never compared with a file produced by TensorFlow itself in this
image ("parity unpinned"); the reader also accepts Snappy-compressed blocks, which TF's table builder may emit.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
HEADER_KEY = b""

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_NP_OF_DT = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DT_OF_KIND = {(v.kind, v.itemsize): k for k, v in _NP_OF_DT.items()}


# ---- CRC-32C (Castagnoli), two bytes per table step ---------------------------------------------------------------------
def _make_tables():
    t8 = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
        t8[i] = c
    idx = np.arange(65536, dtype=np.uint32)
    lo = t8[idx & 0xff] ^ (idx >> 8)                 # one byte step applied to the 16-bit value
    t16 = t8[lo & 0xff] ^ (lo >> 8)                   # second byte step
    return [int(x) for x in t8], [int(x) for x in t16]


_T8, _T16 = _make_tables()


def crc32c(data, crc: int = 0) -> int:
    mv = memoryview(data).cast("B")
    c = crc ^ 0xffffffff
    n2 = len(mv) // 2 * 2
    if n2:
        t16 = _T16
        for w in np.frombuffer(mv[:n2], dtype="<u2").tolist():
            c = t16[(c ^ w) & 0xffff] ^ (c >> 16)
    if len(mv) != n2:
        c = _T8[(c ^ mv[-1]) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def mask_crc(c: int) -> int:
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xffffffff


def unmask_crc(m: int) -> int:
    r = (m - _MASK_DELTA) & 0xffffffff
    return ((r >> 17) | (r << 15)) & 0xffffffff


# ---- varints / minimal protobuf -----------------------------------------------------------------------------------------
def _put_varint(out: bytearray, v: int):
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)


def _get_varint(buf, pos: int) -> Tuple[int, int]:
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7f) << shift
        if b < 0x80:
            return v, pos
        shift += 7


def _pb_fields(buf) -> Iterable[Tuple[int, int, object]]:
    """(field number, wire type, value) of a serialised message; value = int (varint / fixed) or bytes (length-delimited)."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _get_varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield f, wt, v


def _encode_entry(dtype: int, shape: Tuple[int, ...], offset: int, size: int, crc_masked: int) -> bytes:
    sh = bytearray()
    for d in shape:                       # TensorShapeProto.dim = 2 { size = 1 }
        dim = bytearray()
        dim.append(0x08)
        _put_varint(dim, int(d))
        sh.append(0x12)
        _put_varint(sh, len(dim))
        sh += dim
    e = bytearray()
    e.append(0x08)
    _put_varint(e, dtype)                 # dtype = 1
    e.append(0x12)
    _put_varint(e, len(sh))
    e += sh                               # shape = 2 (present even for scalars)
    if offset:
        e.append(0x20)
        _put_varint(e, offset)            # offset = 4 (shard_id = 3 stays 0)
    e.append(0x28)
    _put_varint(e, size)                  # size = 5
    e.append(0x35)
    e += struct.pack("<I", crc_masked)    # crc32c = 6 (fixed32)
    return bytes(e)


def _decode_entry(buf) -> dict:
    out = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for f, _wt, v in _pb_fields(buf):
        if f == 1:
            out["dtype"] = v
        elif f == 2:
            dims = []
            for f2, _w2, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _w3, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                    dims.append(size)
            out["shape"] = tuple(dims)
        elif f == 3:
            out["shard_id"] = v
        elif f == 4:
            out["offset"] = v
        elif f == 5:
            out["size"] = v
        elif f == 6:
            out["crc32c"] = v
        elif f == 7:
            out["sliced"] = True
    return out


_HEADER_PROTO = bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])       # num_shards = 1, endianness LITTLE (default), version.producer = 1


# ---- SSTable ------------------------------------------------------------------------------------------------------------
class _BlockBuilder:
    def __init__(self, restart_interval: int):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last_key = b""

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count % self.interval == 0:
            if self.count:
                self.restarts.append(len(self.buf))
        else:
            m = min(len(key), len(self.last_key))
            while shared < m and key[shared] == self.last_key[shared]:
                shared += 1
        _put_varint(self.buf, shared)
        _put_varint(self.buf, len(key) - shared)
        _put_varint(self.buf, len(value))
        self.buf += key[shared:]
        self.buf += value
        self.last_key = key
        self.count += 1

    def size(self) -> int:
        return len(self.buf) + 4 * (len(self.restarts) + 1)

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _write_table(path: str, items: List[Tuple[bytes, bytes]], block_size: int = 262144):
    """items: (key, value) sorted by key, bytewise."""
    out = bytearray()

    def emit(block: bytes) -> bytes:                    # -> encoded BlockHandle
        off = len(out)
        out.extend(block)
        out.append(0)                                   # kNoCompression
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        h = bytearray()
        _put_varint(h, off)
        _put_varint(h, len(block))
        return bytes(h)

    index = _BlockBuilder(1)
    data = _BlockBuilder(16)
    for key, value in items:
        if data.count and key <= data.last_key:
            raise ValueError("table keys must be strictly increasing")
        data.add(key, value)
        if data.size() >= block_size:
            index.add(data.last_key, emit(data.finish()))
            data = _BlockBuilder(16)
    if data.count:
        index.add(data.last_key, emit(data.finish()))
    meta_handle = emit(_BlockBuilder(16).finish())      # empty metaindex block
    index_handle = emit(index.finish())
    footer = bytearray(meta_handle + index_handle)
    footer += b"\x00" * (40 - len(footer))
    footer += struct.pack("<Q", TABLE_MAGIC)
    out += footer
    with open(path, "wb") as fh:
        fh.write(out)


def _snappy_decompress(src: bytes) -> bytes:
    n, pos = _get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            ln = 1 + (tag >> 2)
            off = src[pos] | (src[pos + 1] << 8)
            pos += 2
        else:
            ln = 1 + (tag >> 2)
            off = int.from_bytes(src[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                             # (copies may overlap their own output)
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("corrupt snappy block (length)")
    return bytes(out)


def _read_block(buf: bytes, off: int, size: int, verify: bool) -> bytes:
    body, ctype = buf[off:off + size], buf[off + size]
    if verify:
        want = struct.unpack_from("<I", buf, off + size + 1)[0]
        if unmask_crc(want) != crc32c(buf[off:off + size + 1]):
            raise ValueError("table block checksum mismatch")
    if ctype == 0:
        return body
    if ctype == 1:
        return _snappy_decompress(body)
    raise ValueError(f"unsupported table block compression {ctype}")


def _block_entries(block: bytes) -> Iterable[Tuple[bytes, bytes]]:
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 * (nrestart + 1)
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _read_table(path: str, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    with open(path, "rb") as fh:
        buf = fh.read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not an SSTable (bad magic)")
    foot = buf[len(buf) - 48:]
    _mo, p = _get_varint(foot, 0)
    _ms, p = _get_varint(foot, p)
    io, p = _get_varint(foot, p)
    isz, p = _get_varint(foot, p)
    out = []
    for _k, handle in _block_entries(_read_block(buf, io, isz, verify)):
        bo, q = _get_varint(handle, 0)
        bs, q = _get_varint(handle, q)
        out.extend(_block_entries(_read_block(buf, bo, bs, verify)))
    return out


# ---- public API ---------------------------------------------------------------------------------------------------------
def data_path(prefix: str) -> str:
    return f"{prefix}.data-00000-of-00001"


def write_checkpoint(prefix: str, tensors: Dict[str, np.ndarray], block_size: int = 262144):
    """`saver.save(session, prefix)`: writes <prefix>.index and <prefix>.data-00000-of-00001 (float32/float64/int32/int64)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items: List[Tuple[bytes, bytes]] = [(HEADER_KEY, _HEADER_PROTO)]
    offset = 0
    with open(data_path(prefix) + ".tmp", "wb") as fh:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name])
            code = _DT_OF_KIND.get((a.dtype.kind, a.dtype.itemsize))
            if code is None:
                raise TypeError(f"{name}: dtype {a.dtype} is not supported by this bundle writer")
            raw = np.ascontiguousarray(a, dtype=_NP_OF_DT[code]).tobytes()
            fh.write(raw)
            items.append((name.encode(), _encode_entry(code, tuple(a.shape), offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    os.replace(data_path(prefix) + ".tmp", data_path(prefix))
    _write_table(prefix + ".index.tmp", items, block_size)
    os.replace(prefix + ".index.tmp", prefix + ".index")       # the index appears last: a watcher never sees a half-written checkpoint


class CheckpointReader:
    """pywrap_tensorflow.NewCheckpointReader(prefix): has_tensor / get_tensor / get_variable_to_shape_map."""

    def __init__(self, prefix: str, verify: bool = True):
        self.prefix, self.verify = prefix, verify
        self.entries: Dict[str, dict] = {}
        for key, value in _read_table(prefix + ".index", verify):
            if key == HEADER_KEY:
                hdr = {f: v for f, _w, v in _pb_fields(value)}
                if hdr.get(1, 1) != 1:
                    raise ValueError(f"{prefix}: {hdr.get(1)} data shards; only single-shard bundles are supported")
                if hdr.get(2, 0) != 0:
                    raise ValueError(f"{prefix}: big-endian bundle")
                continue
            e = _decode_entry(value)
            if e["sliced"]:
                raise ValueError(f"{prefix}: {key.decode()} is a partitioned variable; not supported")
            self.entries[key.decode()] = e
        self._data = None

    def has_tensor(self, name: str) -> bool:
        return name in self.entries

    def get_variable_to_shape_map(self) -> Dict[str, List[int]]:
        return {k: list(e["shape"]) for k, e in self.entries.items()}

    def get_tensor(self, name: str) -> np.ndarray:
        e = self.entries[name]
        if e["dtype"] not in _NP_OF_DT:
            raise TypeError(f"{name}: tensor dtype enum {e['dtype']} is not supported")
        if self._data is None:
            with open(data_path(self.prefix), "rb") as fh:
                self._data = fh.read()
        raw = self._data[e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError(f"{name}: data file is truncated")
        if self.verify and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError(f"{name}: tensor checksum mismatch")
        return np.frombuffer(raw, dtype=_NP_OF_DT[e["dtype"]]).reshape(e["shape"]).copy()


def read_checkpoint(prefix: str) -> Dict[str, np.ndarray]:
    r = CheckpointReader(prefix)
    return {k: r.get_tensor(k) for k in r.entries}


# ---- CheckpointState ("checkpoint" file) --------------------------------------------------------------------------------
def update_checkpoint_state(directory: str, latest: str, all_paths: List[str]):
    lines = [f'model_checkpoint_path: "{latest}"'] + [f'all_model_checkpoint_paths: "{p}"' for p in all_paths]
    tmp = os.path.join(directory, "checkpoint.tmp")
    with open(tmp, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    os.replace(tmp, os.path.join(directory, "checkpoint"))


def read_checkpoint_state(directory: str) -> Tuple[Optional[str], List[str]]:
    path = os.path.join(directory, "checkpoint")
    if not os.path.exists(path):
        return None, []
    latest, every = None, []
    with open(path) as fh:
        for line in fh:
            k, _, v = line.partition(":")
            v = v.strip().strip('"')
            if k.strip() == "model_checkpoint_path":
                latest = v
            elif k.strip() == "all_model_checkpoint_paths":
                every.append(v)
    return latest, every


def latest_checkpoint(directory: str) -> Optional[str]:
    """tf.train.latest_checkpoint: the prefix named by the `checkpoint` state file, if its index exists."""
    latest, _ = read_checkpoint_state(directory)
    if latest is None:
        return None
    prefix = latest if os.path.isabs(latest) else os.path.join(directory, latest)
    return prefix if os.path.exists(prefix + ".index") else None
