"""CPU tests of the TensorFlow-free checkpoint reader (matryodshka_amd/tf_checkpoint.py): known answers
of the format's fixed constants, and a write -> read round trip of a full network's variables."""
import os
import struct

import numpy as np

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
