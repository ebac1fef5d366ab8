"""CPU tests of the TensorFlow-free checkpoint reader (matryodshka_amd/tf_checkpoint.py): known answers
of the format's fixed constants, and a write -> read round trip of a full network's variables."""
import os
import struct

import numpy as np
import pytest

from matryodshka_amd import tf_checkpoint as T
from oracle import nets as onets


def test_crc32c_known_answers():
    assert T.crc32c(b"123456789") == 0xE3069283          # the CRC-32C (Castagnoli) check value
    assert T.crc32c(b"") == 0
    # leveldb/TensorFlow mask: rotate right by 15, add 0xa282ead8
    assert T.masked_crc32c(b"123456789") == (((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xa282ead8) & 0xffffffff


def test_varint_and_block_prefix_compression():
    for v in (0, 1, 127, 128, 300, 2 ** 31, 2 ** 40 + 5):
        enc = T._put_varint(v)
        assert T._varint(enc, 0) == (v, len(enc))
    pairs = [(b"net/conv1_1/LayerNorm/beta", b"a"), (b"net/conv1_1/LayerNorm/gamma", b"bb"), (b"net/conv1_1/weights", b"")]
    blk = T._block(pairs, restart_interval=16) + b"\x00"          # + compression-type byte
    assert T._read_block(blk, 0, len(blk) - 1) == pairs
    assert len(blk) < sum(len(k) + len(v) for k, v in pairs) + 20  # shared prefixes are not stored twice


def test_round_trip_network_variables(tmp_path):
    w = onets.init_weights(24, 8, ngf=16, coord_net=True, seed=1, randomize_affine=True)
    tensors = {"net/" + k: v for k, v in w.items()}
    tensors["global_step"] = np.array(400000, dtype=np.int64)
    tensors["net/conv1_1/weights/Adam"] = np.zeros((3, 3, 25, 16), np.float32)      # optimizer slot: must be skipped
    prefix = T.write_checkpoint(str(tmp_path / "model.ckpt-400000"), tensors, block_entries=5)
    header, entries = T.read_index(prefix + ".index")
    assert header["num_shards"] == 1 and set(entries) == set(tensors)
    assert entries["net/conv1_1/weights"]["shape"] == (3, 3, 25, 16) and entries["global_step"]["shape"] == ()
    got = T.load_checkpoint(prefix, verify_crc=True)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and np.array_equal(got[k], v), k
    weights, step = T.network_weights(str(tmp_path))               # directory -> `checkpoint` state file
    assert step == 400000 and set(weights) == {"net/" + k for k in w}
    assert T.latest_checkpoint(str(tmp_path)) == prefix
    # the dict is what the host-side weight plumbing takes (names with the `net/` scope)
    from matryodshka_amd import nets
    blob = nets.flatten_params(weights, 24, 8, 16, True)
    assert np.array_equal(blob, nets.flatten_params(w, 24, 8, 16, True))


def test_rejects_non_checkpoints(tmp_path):
    p = tmp_path / "x.index"
    p.write_bytes(b"\x00" * 64)
    try:
        T.read_index(str(p))
    except ValueError as e:
        assert "magic" in str(e)
    else:
        raise AssertionError("bad magic accepted")
    # footer layout: the magic is the last 8 bytes, little endian
    assert struct.pack("<Q", T.TABLE_MAGIC) == bytes.fromhex("57fb808b247547db")


# ---------------------------------------------------------------------------------------------------------
# A checkpoint assembled byte by byte from the format description -- none of tf_checkpoint's writer helpers:
# two data shards, a partitioned variable (2 slices along axis 0), Adam slot variables with TensorFlow's real
# names, data blocks of 20 entries (restart interval 16 -> two restart points) and one-entry blocks.
def _vi(v):
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7f) | 0x80])
        v >>= 7
    return out + bytes([v])


def _crc32c_bitwise(data):
    c = 0xffffffff
    for byte in data:
        c ^= byte
        for _ in range(8):
            c = (c >> 1) ^ (0x82f63b78 & -(c & 1))
    return c ^ 0xffffffff


def _mask(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff


def _entry_bytes(dtype_id, shape, shard, offset, size, crc, slices=()):
    shape_pb = b"".join(b"\x12" + _vi(len(b"\x08" + _vi(d))) + b"\x08" + _vi(d) for d in shape)   # Dim{size=1} in field 2
    out = b"\x08" + _vi(dtype_id) + b"\x12" + _vi(len(shape_pb)) + shape_pb
    if shard:
        out += b"\x18" + _vi(shard)
    if offset:
        out += b"\x20" + _vi(offset)
    if size:
        out += b"\x28" + _vi(size)
    if crc is not None:
        out += b"\x35" + struct.pack("<I", crc)                # field 6, fixed32
    for ext in slices:                                         # field 7: TensorSliceProto{ repeated Extent extent = 1 }
        exts = b""
        for st, ln in ext:
            e = (b"\x08" + _vi(st) if st else b"") + (b"\x10" + _vi(ln) if ln >= 0 else b"")
            exts += b"\x0a" + _vi(len(e)) + e
        out += b"\x3a" + _vi(len(exts)) + exts
    return out


def _table_block(pairs, interval):
    body, restarts, prev = b"", [], b""
    for i, (k, v) in enumerate(pairs):
        if i % interval == 0:
            restarts.append(len(body))
            shared = 0
        else:
            shared = len(os.path.commonprefix([prev, k]))
        body += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    restarts = restarts or [0]
    return body + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))


def _slice_key(name, extents):
    """OrderedCode: num(0), escaped string, num(rank), then signed (start, length) per dimension -- the few value
    ranges used here written out literally: 0 -> 80, -1 -> 7f, 1..63 -> 80|v."""
    def signed(v):
        assert -64 <= v < 64
        return bytes([0x80 ^ (v & 0xff)])
    key = b"\x00" + name.encode() + b"\x00\x01" + b"\x01" + bytes([len(extents)])
    for st, ln in extents:
        key += signed(st) + signed(ln)
    return key


def _assemble_fixture(tmp_path, interval=16, per_block=20, tensors=None, part=None, part_name="net/part/weights", split=4):
    """Writes a TF V2 bundle byte by byte (NOT with tf_checkpoint.write_checkpoint): two data shards, multi-restart table
    blocks, Adam slots, and one variable partitioned along axis 0 at `split`.  tensors / part: the plain variables and
    the partitioned one (default: a synthetic set; tests/test_gpu_harness.py passes the variables of a real small net)."""
    rng = np.random.RandomState(42)
    if tensors is None:
        tensors = {}
        for i in range(23):                                   # enough plain variables for multi-restart blocks
            tensors["net/conv%02d/weights" % i] = rng.normal(size=(3, 3, 2, 4)).astype(np.float32)
            tensors["net/conv%02d/weights/Adam" % i] = np.zeros((3, 3, 2, 4), np.float32)
            tensors["net/conv%02d/weights/Adam_1" % i] = np.ones((3, 3, 2, 4), np.float32)
        tensors["beta1_power"] = np.array(0.9 ** 7, np.float32)
        tensors["beta2_power"] = np.array(0.999 ** 7, np.float32)
        tensors["global_step"] = np.array(7, np.int64)
    else:
        tensors = dict(tensors)
    if part is None:
        part = rng.normal(size=(6, 5)).astype(np.float32)     # partitioned along axis 0: rows [0,4) and [4,6)
    rest = [(0, -1)] * (part.ndim - 1)
    ext0, ext1 = [(0, split)] + rest, [(split, part.shape[0] - split)] + rest
    shards = [bytearray(), bytearray()]
    pairs = [(b"", b"\x08\x02")]                              # BundleHeaderProto: num_shards = 2
    k0, k1 = _slice_key(part_name, ext0), _slice_key(part_name, ext1)
    for key, arr, sid in ((k0, part[:split], 0), (k1, part[split:], 1)):
        raw = np.ascontiguousarray(arr).tobytes()
        pairs.append((key, _entry_bytes(1, arr.shape, sid, len(shards[sid]), len(raw), _mask(_crc32c_bitwise(raw)))))
        shards[sid] += raw
    named = []
    for n, (name, arr) in enumerate(sorted(tensors.items())):
        sid = n % 2                                           # alternate the shards
        raw = arr.tobytes()
        dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 9}[arr.dtype]
        named.append((name.encode(), _entry_bytes(dt, arr.shape, sid, len(shards[sid]), len(raw), _mask(_crc32c_bitwise(raw)))))
        shards[sid] += raw
    named.append((part_name.encode(), _entry_bytes(1, part.shape, 0, 0, 0, None, slices=[ext0, ext1])))
    pairs += sorted(named)
    assert [p[0] for p in pairs] == sorted(p[0] for p in pairs)          # a table's keys are sorted
    out, index = b"", []
    for i in range(0, len(pairs), per_block):
        chunk = pairs[i:i + per_block]
        blk = _table_block(chunk, interval)
        index.append((chunk[-1][0], _vi(len(out)) + _vi(len(blk))))
        out += blk + b"\x00" + b"\x00\x00\x00\x00"
    meta = _table_block([], interval)
    mh = _vi(len(out)) + _vi(len(meta))
    out += meta + b"\x00" + b"\x00\x00\x00\x00"
    ib = _table_block(index, 1)
    ih = _vi(len(out)) + _vi(len(ib))
    out += ib + b"\x00" + b"\x00\x00\x00\x00"
    footer = mh + ih
    out += footer + b"\x00" * (40 - len(footer)) + bytes.fromhex("57fb808b247547db")
    prefix = str(tmp_path / "model.latest-7")
    open(prefix + ".index", "wb").write(out)
    for sid in (0, 1):
        open("%s.data-%05d-of-00002" % (prefix, sid), "wb").write(bytes(shards[sid]))
    tensors[part_name] = part
    return prefix, tensors


def test_hand_assembled_multi_shard_partitioned_checkpoint(tmp_path):
    prefix, tensors = _assemble_fixture(tmp_path)
    header, entries = T.read_index(prefix + ".index")
    assert header["num_shards"] == 2
    assert entries["net/part/weights"]["sliced"] and entries["net/part/weights"]["slices"] == [[(0, 4), (0, -1)], [(4, 2), (0, -1)]]
    assert T.encode_tensor_name_slice("net/part/weights", [(0, 4), (0, -1)]) == _slice_key("net/part/weights", [(0, 4), (0, -1)])
    got = T.load_checkpoint(prefix)                                        # CRC verification is on by default
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    # what test.py's Saver(trainable_variables + [global_step]) restores: no Adam slots, no beta powers (msi.py:727-733, :985)
    w, step = T.network_weights(prefix)
    assert step == 7 and "net/part/weights" in w and len(w) == 24
    assert not any("Adam" in k or "power" in k for k in w)
    assert T.latest_checkpoint(str(tmp_path)) == prefix                    # no `checkpoint` state file: highest step wins


def test_crc_mismatch_and_truncation_are_detected(tmp_path):
    prefix, _ = _assemble_fixture(tmp_path)
    path = prefix + ".data-00001-of-00002"
    raw = bytearray(open(path, "rb").read())
    raw[100] ^= 0x01                                                       # one flipped bit in shard 1
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="crc32c"):
        T.load_checkpoint(prefix)
    assert len(T.load_checkpoint(prefix, verify_crc=False)) > 0            # explicitly opted out
    open(path, "wb").write(bytes(raw[:64]))
    with pytest.raises(ValueError, match="truncated|crc32c"):
        T.load_checkpoint(prefix)


def test_ordered_code_signed_numbers():
    """WriteSignedNumIncreasing known answers [TF-knowledge: ordered_code.cc]: order-preserving, length-prefixed."""
    enc = T._oc_signed_increasing
    assert enc(0) == b"\x80" and enc(-1) == b"\x7f" and enc(63) == b"\xbf" and enc(-64) == b"\x40"
    assert enc(64) == b"\xc0\x40" and enc(-65) == b"\x3f\xbf" and enc(8191) == b"\xdf\xff" and enc(8192) == b"\xe0\x20\x00"
    vals = [-70000, -8193, -8192, -65, -64, -1, 0, 1, 63, 64, 300, 8191, 8192, 70000, 2 ** 40]
    assert sorted(vals, key=enc) == vals                                   # byte order == numeric order
    assert T._oc_string(b"a\x00b\xffc") == b"a\x00\xffb\xff\x00c\x00\x01"
