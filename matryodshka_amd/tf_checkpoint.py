"""Reader for TensorFlow "V2" checkpoints (tensor bundles) without TensorFlow: what
`saver.restore(sess, tf.train.latest_checkpoint(dir))` (reference test.py:192-202) reads.

A checkpoint `<prefix>` is `<prefix>.index` + `<prefix>.data-SSSSS-of-NNNNN`:
  * `.index` is a LevelDB-format sorted string table (tensorflow/core/lib/io/table): data blocks of
    prefix-compressed (key, value) entries, an index block of block handles, and a 48-byte footer
    ending in the magic 0xdb4775248b80fb57.  Key "" holds the BundleHeaderProto, every other key is a
    variable name whose value is a BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5,
    crc32c=6};
  * the data shards hold the raw little-endian tensor bytes at [offset, offset+size).
  * a PARTITIONED variable has one entry under its name (full shape, `slices` = field 7, no data) and one entry
    per slice under the key EncodeTensorNameSlice(name, slice) (tensorflow/core/util/saved_tensor_slice_util:
    OrderedCode 0, the name as an escaped string, the rank, and (start, length) per dimension), which sorts
    before every plain name.
Format knowledge, not code, is taken from TensorFlow's public sources [TF-knowledge]; there is no
TensorFlow (and no checkpoint of the reference) in the build container, so the reader is tested against
`write_checkpoint` below AND against a byte-level fixture assembled independently from the format
description (tests/test_tf_checkpoint.py: two shards, a partitioned variable, Adam slots, multi-restart blocks).
Every tensor restored is checked against its stored crc32c (verify_crc=True by default).

    weights = load_checkpoint("checkpoints/ods-wotemp-elpips-coord/model.ckpt-400000")
    model = MSI(weights=weights)        # names 'net/<layer>/weights', ... (nets.variable_shapes)
"""
import glob
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---------------------------------------------------------------------------- varints / protobuf
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf):
    """Yields (field number, wire type, value) of one protobuf message (value: int or bytes)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _parse_shape(buf):
    dims = []
    for field, _, v in _proto_fields(buf):
        if field == 2:                                  # repeated Dim dim = 2 { int64 size = 1; string name = 2 }
            size = 0
            for f2, _, v2 in _proto_fields(v):
                if f2 == 1:
                    size = v2
            dims.append(size)
    return tuple(dims)


def _parse_entry(buf):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False, "slices": []}
    for field, _, v in _proto_fields(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            e["shape"] = _parse_shape(v)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif field == 7:                                # repeated TensorSliceProto slices
            e["sliced"] = True
            e["slices"].append(_parse_slice(v))
    return e


def _parse_slice(buf):
    """TensorSliceProto { repeated Extent extent = 1 { int64 start = 1; int64 length = 2 (absent = full) } }
    -> [(start, length or -1)] per dimension."""
    dims = []
    for field, _, v in _proto_fields(buf):
        if field == 1:
            start, length = 0, -1
            for f2, _, v2 in _proto_fields(v):
                if f2 == 1:
                    start = v2
                elif f2 == 2:
                    length = v2
            dims.append((start, length))
    return dims


# ---- OrderedCode (tensorflow/core/lib/strings/ordered_code) as far as EncodeTensorNameSlice needs it
def _oc_num_increasing(v):
    """WriteNumIncreasing: one length byte, then the value big-endian without leading zero bytes."""
    body = b"" if v == 0 else v.to_bytes((v.bit_length() + 7) // 8, "big")
    return bytes([len(body)]) + body


def _oc_string(b):
    """WriteString: byte 0x00 -> 0x00 0xff, byte 0xff -> 0xff 0x00, terminated by 0x00 0x01."""
    esc = {0: bytes([0, 255]), 255: bytes([255, 0])}
    return b"".join(esc.get(c, bytes([c])) for c in b) + bytes([0, 1])


def _oc_signed_increasing(val):
    """WriteSignedNumIncreasing: sign-extended big-endian, the leading bits carry the length (len 1: 0x80 ^ val for
    -64 <= val < 64; len n: n one-bits then a zero bit precede the payload, complemented for negatives)."""
    x = ~val if val < 0 else val
    if x < 64:
        return bytes([(0x80 ^ val) & 0xff])
    bits = x.bit_length() + 1                            # magnitude bits + sign
    n = 2
    while 7 * n < bits:                                  # n bytes carry 7 n payload bits (n <= 8), kBitsToLength
        n += 1
    if n > 8:
        n = 9 if bits <= 63 + 1 else 10
    raw = (val & ((1 << 80) - 1)).to_bytes(10, "big")    # two's complement, sign-extended to 10 bytes
    out = bytearray(raw[10 - n:])
    header = [(0x80, 0), (0xc0, 0), (0xe0, 0), (0xf0, 0), (0xf8, 0), (0xfc, 0), (0xfe, 0), (0xff, 0), (0xff, 0x80), (0xff, 0xc0)][n - 1]
    out[0] ^= header[0]
    if n >= 2:
        out[1] ^= header[1]
    return bytes(out)


def encode_tensor_name_slice(name, extents):
    """saved_tensor_slice_util EncodeTensorNameSlice: the table key of one slice of a partitioned variable.
    extents: [(start, length)] per dimension; a full dimension is (0, -1) (TensorSlice::kFullExtent)."""
    out = _oc_num_increasing(0) + _oc_string(name.encode()) + _oc_num_increasing(len(extents))
    for start, length in extents:
        out += _oc_signed_increasing(start) + _oc_signed_increasing(length)
    return out


# ---------------------------------------------------------------------------- table (.index)
def _read_block(data, offset, size):
    """Decodes one table block -> list of (key, value).  The byte after the block is its compression
    type (0 = none, the only one TensorFlow's bundle writer uses; 1 = snappy is rejected)."""
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError("compressed table block (type %d) is not supported" % ctype)
    blk = data[offset:offset + size]
    nrestarts = struct.unpack("<I", blk[-4:])[0]
    limit = size - 4 - 4 * nrestarts
    pos, key, out = 0, b"", []
    while pos < limit:
        shared, pos = _varint(blk, pos)
        nonshared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + bytes(blk[pos:pos + nonshared])
        pos += nonshared
        out.append((key, bytes(blk[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(path):
    """`<prefix>.index` -> (header dict, {variable name: entry dict})."""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos)            # metaindex handle (unused)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)         # index block handle
    isize, pos = _varint(footer, pos)
    entries, header = {}, {}
    for _, handle in _read_block(data, ioff, isize):
        boff, p2 = _varint(handle, 0)
        bsize, _ = _varint(handle, p2)
        for key, value in _read_block(data, boff, bsize):
            if key == b"":
                for field, _, v in _proto_fields(value):      # BundleHeaderProto: num_shards = 1, endianness = 2
                    if field == 1:
                        header["num_shards"] = v
                    elif field == 2:
                        header["endianness"] = v
            elif key[:1] == b"\x00":                  # slice of a partitioned variable (binary OrderedCode key)
                entries[key] = _parse_entry(value)
            else:
                entries[key.decode()] = _parse_entry(value)
    header.setdefault("num_shards", 1)
    if header.get("endianness", 0) != 0:
        raise ValueError("big-endian checkpoints are not supported")
    return header, entries


def latest_checkpoint(ckpt_dir):
    """tf.train.latest_checkpoint: the `checkpoint` state file's model_checkpoint_path, else the index
    with the highest trailing step number."""
    state = os.path.join(ckpt_dir, "checkpoint")
    if os.path.exists(state):
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(state).read())
        if m:
            p = m.group(1)
            return p if os.path.isabs(p) else os.path.join(ckpt_dir, p)
    best, best_step = None, -1
    for idx in glob.glob(os.path.join(ckpt_dir, "*.index")):
        m = re.search(r"-(\d+)\.index$", idx)
        step = int(m.group(1)) if m else 0
        if step > best_step:
            best, best_step = idx[:-len(".index")], step
    return best


def load_checkpoint(prefix, names=None, verify_crc=True):
    """{variable name: np.ndarray} of a checkpoint prefix (or of a directory's latest checkpoint).
    `names`: optional predicate / collection restricting what is read (optimizer slots are large).
    Every tensor (and every slice of a partitioned one) is checked against its stored masked crc32c."""
    if os.path.isdir(prefix):
        prefix = latest_checkpoint(prefix)
        if prefix is None:
            raise FileNotFoundError("no checkpoint found")
    header, entries = read_index(prefix + ".index")
    nshards = header["num_shards"]
    shards = {}
    out = {}

    def read(name, e):
        sid = e["shard_id"]
        if sid >= nshards:
            raise ValueError("%s: shard %d of a %d-shard checkpoint" % (name, sid, nshards))
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, nshards), dtype=np.uint8, mode="r")
        raw = np.ascontiguousarray(shards[sid][e["offset"]:e["offset"] + e["size"]])
        if raw.size != e["size"]:
            raise ValueError("%s: data shard %d is truncated" % (name, sid))
        if verify_crc and e["crc32c"] is not None and masked_crc32c(raw) != e["crc32c"]:
            raise ValueError("%s: crc32c mismatch" % name)
        return raw.view(_DTYPES[e["dtype"]]).reshape(e["shape"])

    for name, e in entries.items():
        if isinstance(name, bytes):
            continue                                           # slice entries are read through their variable
        if names is not None and not (names(name) if callable(names) else name in names):
            continue
        if e["dtype"] not in _DTYPES:
            continue                                           # strings etc.: not network weights
        if not e["sliced"]:
            out[name] = read(name, e).copy()
            continue
        full = np.empty(e["shape"], dtype=_DTYPES[e["dtype"]])
        covered = 0
        for ext in e["slices"]:
            key = encode_tensor_name_slice(name, ext)
            if key not in entries:
                raise ValueError("%s: slice %r of the partitioned variable is missing from the index" % (name, ext))
            idx = tuple(slice(None) if ln < 0 else slice(st, st + ln) for st, ln in ext)
            part = read(name, entries[key])
            full[idx] = part
            covered += part.size
        if covered != full.size:
            raise ValueError("%s: slices cover %d of %d elements" % (name, covered, full.size))
        out[name] = full
    return out


def network_weights(prefix):
    """The variables test.py restores for inference (trainable `net/...` + global_step), as the dict
    MSI(weights=...) takes, plus the global step."""
    w = load_checkpoint(prefix, names=lambda n: (n.startswith("net/") and "/Adam" not in n) or n == "global_step")
    step = int(w.pop("global_step")) if "global_step" in w else 0
    return w, step


# ---------------------------------------------------------------------------- crc32c (Castagnoli), masked as TF stores it
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data):
    t, c = _crc_table(), 0xffffffff
    for b in data:
        c = t[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def _crc32c_fast(data):
    """The native table-driven implementation when libmsi_hip.so is built (68 MB of weights in ~0.1 s; the pure
    Python loop above takes minutes), else the Python one."""
    try:
        from . import _native
    except Exception:       # library not built: the checkpoint reader stays usable
        return crc32c(bytes(data))
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data)
    return int(_native.lib.msi_crc32c_host(a.ctypes.data, a.nbytes, 0))


def masked_crc32c(data):
    c = _crc32c_fast(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff


# ---------------------------------------------------------------------------- writer (tests, export)
def _block(pairs, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(pairs):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _pb_varint_field(field, v):
    return _put_varint(field << 3) + _put_varint(v)


def _pb_bytes_field(field, b):
    return _put_varint((field << 3) | 2) + _put_varint(len(b)) + b


def write_checkpoint(prefix, tensors, block_entries=8):
    """Writes {name: array} as a single-shard V2 checkpoint (the format read_index / load_checkpoint
    parse): used by the tests and to export weights for the reference."""
    names = sorted(tensors)
    data, pairs = bytearray(), [(b"", _pb_varint_field(1, 1))]            # header: num_shards = 1 (little endian = 0)
    for n in names:
        a = np.asarray(tensors[n], order="C")      # (ascontiguousarray would turn a scalar into shape (1,))
        raw = a.tobytes()
        shape = b"".join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in a.shape)
        entry = _pb_varint_field(1, _DTYPE_IDS[a.dtype]) + _pb_bytes_field(2, shape)
        if len(data):
            entry += _pb_varint_field(4, len(data))
        entry += _pb_varint_field(5, len(raw)) + _put_varint((6 << 3) | 5) + struct.pack("<I", masked_crc32c(raw))
        pairs.append((n.encode(), entry))
        data += raw
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    out, index_pairs = bytearray(), []
    for i in range(0, len(pairs), block_entries):
        chunk = pairs[i:i + block_entries]
        blk = _block(chunk)
        index_pairs.append((chunk[-1][0] + b"\x00", _put_varint(len(out)) + _put_varint(len(blk))))
        out += blk + b"\x00" + struct.pack("<I", 0)                         # block trailer: type 0 + crc (not checked)
    meta = _block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out += meta + b"\x00" + struct.pack("<I", 0)
    idx = _block(index_pairs, restart_interval=1)
    idx_handle = _put_varint(len(out)) + _put_varint(len(idx))
    out += idx + b"\x00" + struct.pack("<I", 0)
    footer = meta_handle + idx_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out += footer
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % os.path.basename(prefix))
    return prefix
