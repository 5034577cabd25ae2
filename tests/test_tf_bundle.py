"""Checkpoint container (common/tf_bundle.py): known answers of CRC-32C (RFC 3720 B.4), the LevelDB-style CRC mask, Snappy
streams written out by hand, the byte layout of a table / BundleEntryProto spelled out from the format specification
independently of the writer, multi-block round trips and corruption handling.  No TensorFlow-written file exists in this image
(DESIGN.md section 2): these tests pin the codec to the published formats, not to TF's own output."""
import os
import struct

import numpy as np
import pytest

from tcresnet_amd.common import tf_bundle as B


# ---- CRC-32C ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("data,want", [
    (b"123456789", 0xE3069283),                      # the "check" value of CRC-32C (Castagnoli)
    (bytes(32), 0x8A9136AA),                         # RFC 3720 B.4: 32 bytes of zeroes
    (b"\xff" * 32, 0x62A8AB43),                      #              32 bytes of ones
    (bytes(range(32)), 0x46DD794E),                  #              32 bytes incrementing
    (bytes(range(31, -1, -1)), 0x113FDB5C),          #              32 bytes decrementing
    (b"", 0x00000000),
    (b"a", 0xC1D04330),
])
def test_crc32c_known_answers(data, want):
    assert B.crc32c(data) == want


def test_crc32c_is_incremental_and_handles_odd_lengths():
    rng = np.random.default_rng(0)
    blob = rng.integers(0, 256, 1001, dtype=np.uint8).tobytes()
    whole = B.crc32c(blob)
    for cut in (0, 1, 2, 3, 500, 999, 1000, 1001):
        assert B.crc32c(blob[cut:], B.crc32c(blob[:cut])) == whole
    # bitwise reference implementation
    c = 0xffffffff
    for b in blob:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
    assert whole == c ^ 0xffffffff


def test_crc_mask_is_the_leveldb_rotation():
    assert B.mask_crc(0) == 0xa282ead8                              # rotate right by 15, add kMaskDelta
    assert B.mask_crc(0xE3069283) == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xa282ead8) & 0xffffffff)
    for c in (0, 1, 0x80000000, 0xffffffff, 0xE3069283, 0x8A9136AA):
        assert B.unmask_crc(B.mask_crc(c)) == c
        assert B.mask_crc(c) != c


# ---- Snappy (reader only) ---------------------------------------------------------------------------------------------
def test_snappy_streams_written_by_hand():
    assert B._snappy_decompress(bytes([5, 0x10]) + b"hello") == b"hello"                     # one literal
    assert B._snappy_decompress(bytes([8, 0x04]) + b"ab" + bytes([0x09, 0x02])) == b"abababab"   # literal + overlapping 1-byte-offset copy
    # 2-byte-offset copy: literal "0123456789", copy length 5 from offset 10
    assert B._snappy_decompress(bytes([15, (10 - 1) << 2]) + b"0123456789" + bytes([((5 - 1) << 2) | 2, 10, 0])) == b"012345678901234"
    # long literal (length byte follows the tag)
    lit = bytes(range(200))
    assert B._snappy_decompress(bytes([200 & 0x7f | 0x80, 200 >> 7, 60 << 2, 199]) + lit) == lit
    with pytest.raises(ValueError):
        B._snappy_decompress(bytes([4, 0x09, 0x05]))                                            # copy before any output
    with pytest.raises(ValueError):
        B._snappy_decompress(bytes([9, 0x10]) + b"hello")                                       # declared length 9, got 5


# ---- table / entry layout, spelled out from the specification ----------------------------------------------------------
def _trailer(block: bytes) -> bytes:
    """block trailer: compression type 0 + masked CRC-32C of (block contents + type byte)"""
    c = B.crc32c(block + b"\x00")
    return b"\x00" + struct.pack("<I", (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff)


def test_single_entry_table_bytes(tmp_path):
    key, value = b"conv/weights", b"\x08\x01payload"
    # data block: one entry (shared 0, non-shared len, value len, key, value), restart array [0], 1 restart
    data = bytes([0, len(key), len(value)]) + key + value + struct.pack("<II", 0, 1)
    meta = struct.pack("<II", 0, 1)                                                             # empty metaindex block
    off_meta = len(data) + 5
    off_index = off_meta + len(meta) + 5
    handle_data = bytes([0, len(data)])                                                         # BlockHandle: varint offset, varint size
    index = bytes([0, len(key), len(handle_data)]) + key + handle_data + struct.pack("<II", 0, 1)
    footer = bytes([off_meta, len(meta), off_index, len(index)])
    footer += bytes(40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    want = data + _trailer(data) + meta + _trailer(meta) + index + _trailer(index) + footer
    path = str(tmp_path / "t.index")
    B._write_table(path, [(key, value)])
    got = open(path, "rb").read()
    assert got == want
    assert B._read_table(path) == [(key, value)]


def test_bundle_entry_and_header_bytes():
    # BundleEntryProto: dtype = 1 (varint), shape = 2 { dim = 2 { size = 1 } }, offset = 4, size = 5, crc32c = 6 (fixed32); zero
    # fields (shard_id, offset 0) are omitted as proto3 does
    e = B._encode_entry(B.DT_FLOAT, (3, 1, 16, 24), 300, 4608, 0x01020304)
    want = bytes([0x08, 0x01,
                  0x12, 0x10, 0x12, 0x02, 0x08, 0x03, 0x12, 0x02, 0x08, 0x01, 0x12, 0x02, 0x08, 0x10, 0x12, 0x02, 0x08, 0x18,
                  0x20, 0xac, 0x02,                       # offset 300
                  0x28, 0x80, 0x24,                       # size 4608
                  0x35, 0x04, 0x03, 0x02, 0x01])
    assert e == want
    d = B._decode_entry(e)
    assert (d["dtype"], d["shape"], d["offset"], d["size"], d["crc32c"], d["shard_id"]) == (1, (3, 1, 16, 24), 300, 4608, 0x01020304, 0)
    scalar = B._encode_entry(B.DT_INT64, (), 0, 8, 7)
    assert scalar == bytes([0x08, 0x09, 0x12, 0x00, 0x28, 0x08, 0x35, 0x07, 0, 0, 0])
    assert B._decode_entry(scalar)["shape"] == ()
    # BundleHeaderProto: num_shards = 1, (endianness LITTLE = 0 omitted), version { producer = 1 }
    assert B._HEADER_PROTO == bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])


# ---- round trips --------------------------------------------------------------------------------------------------------
def _tensors():
    rng = np.random.default_rng(3)
    t = {f"TCResNet8/block{i}/conv{j}/{leaf}": rng.standard_normal((9, 1, 8 + i, 8 + j)).astype(np.float32)
         for i in range(6) for j in range(3) for leaf in ("weights", "weights/Momentum", "BatchNorm/moving_variance")}
    t["global_step"] = np.asarray(12345, dtype=np.int64)
    t["beta1_power"] = np.asarray(0.9 ** 7, dtype=np.float32)
    t["d"] = rng.standard_normal((2, 3)).astype(np.float64)
    t["i"] = np.arange(5, dtype=np.int32)
    return t


@pytest.mark.parametrize("block_size", [64, 256, 262144])
def test_multi_block_round_trip(tmp_path, block_size):
    t = _tensors()
    prefix = str(tmp_path / "m" / "Model-7")
    B.write_checkpoint(prefix, t, block_size=block_size)
    if block_size <= 256:
        # the index really spans several data blocks (index block has one entry per data block)
        buf = open(prefix + ".index", "rb").read()
        foot = buf[-48:]
        _, p = B._get_varint(foot, 0); _, p = B._get_varint(foot, p)
        io, p = B._get_varint(foot, p); isz, p = B._get_varint(foot, p)
        assert len(list(B._block_entries(B._read_block(buf, io, isz, True)))) > 4
    r = B.CheckpointReader(prefix)
    assert set(r.entries) == set(t)
    assert r.get_variable_to_shape_map()["global_step"] == []
    for k, v in t.items():
        got = r.get_tensor(k)
        assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
    assert os.path.getsize(B.data_path(prefix)) == sum(v.nbytes for v in t.values())
    # the data file is the tensors back to back in bytewise key order
    off = 0
    for k in sorted(t, key=lambda s: s.encode()):
        assert r.entries[k]["offset"] == off
        off += t[k].nbytes


def test_corruption_is_detected(tmp_path):
    t = _tensors()
    prefix = str(tmp_path / "Model-1")
    B.write_checkpoint(prefix, t, block_size=256)
    idx = open(prefix + ".index", "rb").read()
    # (1) a flipped byte inside a data block -> block checksum
    bad = bytearray(idx); bad[10] ^= 0x40
    open(prefix + ".index", "wb").write(bad)
    with pytest.raises(ValueError, match="checksum"):
        B.CheckpointReader(prefix)
    # (2) truncated index -> bad magic / short file
    open(prefix + ".index", "wb").write(idx[:-9])
    with pytest.raises(ValueError):
        B.CheckpointReader(prefix)
    open(prefix + ".index", "wb").write(idx[:20])
    with pytest.raises(ValueError):
        B.CheckpointReader(prefix)
    open(prefix + ".index", "wb").write(idx)
    # (3) a flipped bit in the data file -> tensor checksum; other tensors still readable
    data = bytearray(open(B.data_path(prefix), "rb").read())
    r0 = B.CheckpointReader(prefix)
    victim = "TCResNet8/block0/conv0/weights"
    data[r0.entries[victim]["offset"] + 5] ^= 0x01
    open(B.data_path(prefix), "wb").write(data)
    r = B.CheckpointReader(prefix)
    with pytest.raises(ValueError, match="checksum"):
        r.get_tensor(victim)
    assert np.array_equal(r.get_tensor("i"), t["i"])
    # (4) truncated data file
    open(B.data_path(prefix), "wb").write(bytes(data[:100]))
    with pytest.raises(ValueError):
        B.CheckpointReader(prefix).get_tensor("i")


def test_snappy_compressed_block_is_accepted(tmp_path):
    """TF's table builder may emit Snappy blocks (type 1): a table whose data block is a literal-only Snappy stream."""
    key, value = b"k", b"v" * 10
    raw = bytes([0, len(key), len(value)]) + key + value + struct.pack("<II", 0, 1)
    comp = bytes([len(raw), (len(raw) - 1) << 2]) + raw                                         # one literal (len < 60)
    c = B.crc32c(comp + b"\x01")
    blk = comp + b"\x01" + struct.pack("<I", B.mask_crc(c))
    meta = struct.pack("<II", 0, 1)
    handle = bytes([0, len(comp)])
    index = bytes([0, len(key), len(handle)]) + key + handle + struct.pack("<II", 0, 1)
    off_meta = len(blk)
    off_index = off_meta + len(meta) + 5
    footer = bytes([off_meta, len(meta), off_index, len(index)])
    footer += bytes(40 - len(footer)) + struct.pack("<Q", B.TABLE_MAGIC)
    path = str(tmp_path / "s.index")
    open(path, "wb").write(blk + meta + _trailer(meta) + index + _trailer(index) + footer)
    assert B._read_table(path) == [(key, value)]
