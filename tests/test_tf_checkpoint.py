"""TF checkpoint V2 (tensor bundle) reader/writer: rnnwavefunctions_amd/tf_checkpoint.py, the stand-in for
tf.train.Saver at 1DTFIM/TrainingRNN_1DTFIM.py:166,219.  CPU only.

Parity: no checkpoint written by real TensorFlow exists in the build container (TF cannot be installed), so the
byte-level agreement with TF's writer is PARITY UNPINNED.  Pinned here: the published constants of the format
(CRC-32C check values, the LevelDB footer magic, the documented header bytes), bit-exact round trips, and the table
features a TF-written index may use (prefix compression, several data blocks, snappy blocks).  A real TF1 file dropped
into tests/golden/tf1_checkpoint/ is picked up by test_real_tf_checkpoint_if_present."""
import glob
import os
import struct

import numpy as np
import pytest

from rnnwavefunctions_amd import params as P
from rnnwavefunctions_amd import tf_checkpoint as T


def test_crc32c_known_answers():
    # RFC 3720 B.4 / the vectors of tensorflow/core/lib/hash/crc32c_test.cc
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    # Extend: crc of a concatenation
    assert T.crc32c(b"world", T.crc32c(b"hello ")) == T.crc32c(b"hello world")
    # the mask is a bijection with the documented constant
    c = T.crc32c(b"foo")
    m = T.mask_crc(c)
    assert m != c and (((m - 0xA282EAD8) & 0xFFFFFFFF) >> 17 | (((m - 0xA282EAD8) & 0xFFFFFFFF) << 15) & 0xFFFFFFFF) == c


def test_varints():
    for n, enc in [(0, b"\x00"), (1, b"\x01"), (127, b"\x7f"), (128, b"\x80\x01"), (300, b"\xac\x02"),
                   (2 ** 32, b"\x80\x80\x80\x80\x10")]:
        assert T.put_varint(n) == enc
        assert T.get_varint(enc, 0) == (n, len(enc))
    with pytest.raises(T.CheckpointError):
        T.get_varint(b"\x80", 0)


def test_header_and_entry_bytes():
    # BundleHeaderProto{num_shards: 1, version{producer: 1}}; endianness LITTLE is the proto3 default and is omitted
    assert T.encode_header(1) == bytes.fromhex("08011a020801")
    assert T.decode_header(T.encode_header(1)) == {"num_shards": 1, "endianness": 0, "producer": 1, "min_consumer": 0}
    e = T.encode_entry(T.DT_FLOAT, (52, 100), 0, 800, 20800, 0xDEADBEEF)
    assert e == bytes.fromhex("0801" "1208" "12020834" "12020864" "20a006" "28c0a201" "35efbeadde")
    d = T.decode_entry(e)
    assert (d["dtype"], d["shape"], d["shard_id"], d["offset"], d["size"], d["crc32c"]) == (1, [52, 100], 0, 800, 20800, 0xDEADBEEF)
    # scalar: empty shape message
    assert T.decode_entry(T.encode_entry(T.DT_INT32, (), 0, 0, 4, 1))["shape"] == []


def test_round_trip_of_every_model_family(tmp_path):
    for name, prm in [("gru", P.init_gru_params([50], seed=3)),
                      ("gru2", P.init_gru_params([10, 10], seed=4, heads=("wf_dense_ampl", "wf_dense_phase"))),
                      ("gru64", P.init_gru_params([20], seed=5, dtype=np.float64)),
                      ("mdrnn", P.init_mdrnn_params(50, seed=6))]:
        prm = dict(prm)
        prm["Variable"] = np.array(12, dtype=np.int32)                           # the reference's un-named global step
        prm["RNNwavefunction/beta1_power"] = np.array(0.9 ** 12, dtype=np.float32)
        prefix = str(tmp_path / name / "model.ckpt")
        T.write_checkpoint(prefix, prm)
        assert sorted(os.listdir(tmp_path / name)) == ["checkpoint", "model.ckpt.data-00000-of-00001", "model.ckpt.index"]
        got = T.read_checkpoint(prefix)
        assert list(got) == sorted(prm, key=lambda k: k.encode())
        for k, v in prm.items():
            assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
        assert T.latest_checkpoint(str(tmp_path / name)) == prefix
        listed = {n: (s, d) for n, s, d in T.list_variables(prefix)}
        assert listed["Variable"] == ((), np.int32)
        sub = T.read_checkpoint(prefix, names=["Variable"])
        assert list(sub) == ["Variable"] and int(sub["Variable"]) == 12
        with pytest.raises(T.CheckpointError, match="not in checkpoint"):
            T.read_checkpoint(prefix, names=["nope"])


def test_index_file_layout(tmp_path):
    """Footer, block trailer and record layout as leveldb's table_format.md describes them."""
    prefix = str(tmp_path / "m.ckpt")
    T.write_checkpoint(prefix, {"a/kernel": np.arange(6, dtype=np.float32).reshape(2, 3), "a/bias": np.zeros(3, np.float32)})
    buf = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", buf[-8:])[0] == 0xDB4775248B80FB57
    # first record: shared 0, key length 0 (the header key ""), value length 6, then the header proto
    assert buf[:9] == b"\x00\x00\x06" + bytes.fromhex("08011a020801")
    # second record: key "a/bias" in full (restart point); third shares the prefix "a/" with it
    assert buf[9:12] == bytes([0, 6, len(T.encode_entry(T.DT_FLOAT, (3,), 0, 0, 12, 0))]) and buf[12:18] == b"a/bias"
    recs = T.read_table(prefix + ".index")
    assert [k for k, _ in recs] == [b"", b"a/bias", b"a/kernel"]
    third = 18 + recs[1][1].__len__()
    assert buf[third] == 2 and buf[third + 1] == 6 and buf[third + 3:third + 9] == b"kernel"
    # data file: tensors back to back in key order
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    assert data == np.zeros(3, np.float32).tobytes() + np.arange(6, dtype=np.float32).tobytes()
    e = T.decode_entry(recs[2][1])
    assert e["offset"] == 12 and e["size"] == 24 and e["crc32c"] == T.mask_crc(T.crc32c(data[12:]))


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "m.ckpt")
    T.write_checkpoint(prefix, {"w": np.arange(100, dtype=np.float64)})
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[17] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(T.CheckpointError, match="tensor checksum"):
        T.read_checkpoint(prefix)
    assert T.read_checkpoint(prefix, verify=False)["w"].shape == (100,)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[4] ^= 1
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(T.CheckpointError, match="block checksum"):
        T.read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(b"not a table at all, but long enough to hold a forty-eight byte footer....")
    with pytest.raises(T.CheckpointError, match="magic"):
        T.read_checkpoint(prefix)
    with pytest.raises(T.CheckpointError, match="no checkpoint index"):
        T.read_checkpoint(str(tmp_path / "absent.ckpt"))


def test_tables_with_many_blocks_and_prefix_compression(tmp_path):
    items = [(("scope/var_%05d/kernel" % i).encode(), bytes([i % 251]) * (i % 37)) for i in range(3000)]
    path = str(tmp_path / "t.sst")
    T.write_table(path, items, block_size=700)
    assert T.read_table(path) == items
    assert os.path.getsize(path) < sum(len(k) + len(v) + 3 for k, v in items)        # prefixes are shared
    with pytest.raises(T.CheckpointError, match="strictly increasing"):
        T.write_table(path, [(b"b", b""), (b"a", b"")])


def test_snappy_blocks_are_read():
    # literals of all three length classes, and copies with 1-, 2- and 4-byte offsets (overlapping run-length form)
    raw = b"abcd" * 5 + bytes(range(70)) + b"x" * 300
    comp = (T.put_varint(len(raw))
            + bytes([3 << 2]) + b"abcd"                       # literal, 4 bytes
            + bytes([((8 - 4) << 2) | 1 | (0 << 5), 4]) * 2   # copy-1: length 8, offset 4 (overlapping), twice
            + bytes([60 << 2, 69]) + bytes(range(70))         # literal with a 1-byte length (70)
            + bytes([0 << 2]) + b"x"                          # literal 'x'
            + bytes([(63 << 2) | 2, 1, 0]) * 4                # copy-2: 64 bytes from offset 1, four times
            + bytes([(42 << 2) | 3, 1, 0, 0, 0]))             # copy-4: 43 bytes from offset 1
    assert T.snappy_decompress(comp) == raw


def test_saver_variables_are_separated_from_the_model():
    prm = P.init_gru_params([10], seed=1)
    dump = dict(prm)
    for k, v in prm.items():
        dump[k + "/Adam"] = np.full_like(v, 0.5)
        dump[k + "/Adam_1"] = np.full_like(v, 0.25)
    dump["RNNwavefunction/beta1_power"] = np.float32(0.9 ** 3)
    dump["RNNwavefunction/beta2_power"] = np.float32(0.999 ** 3)
    dump["Variable"] = np.int32(3)
    model, opt = T.split_saver_variables(dump)
    assert set(model) == set(prm) and set(opt["m"]) == set(prm) and set(opt["v"]) == set(prm)
    assert opt["global_step"] == 3 and abs(opt["beta1_power"] - 0.729) < 1e-6


def test_real_tf_checkpoint_if_present():
    """Pins the reader against TensorFlow's own writer once such a file exists (none can be made in this container)."""
    found = glob.glob(os.path.join(os.path.dirname(__file__), "golden", "tf1_checkpoint", "*.index"))
    if not found:
        pytest.skip("no TF-written checkpoint under tests/golden/tf1_checkpoint/ (parity unpinned, see module docstring)")
    for idx in found:
        got = T.read_checkpoint(idx[:-len(".index")])
        assert got and all(np.all(np.isfinite(v)) for v in got.values() if v.dtype.kind == "f")


def test_converter_tool_round_trips_names_and_values(tmp_path):
    """tools/ckpt_convert.py: npz -> ckpt -> npz keeps every TF variable name (with its '/') and every bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prm = P.init_gru_params([6], seed=2)
    P.save_npz(tmp_path / "a.npz", prm)
    tool = os.path.join(root, "tools", "ckpt_convert.py")
    subprocess.run([sys.executable, tool, "ckpt", str(tmp_path / "a.npz"), str(tmp_path / "m.ckpt")], check=True, capture_output=True)
    out = subprocess.run([sys.executable, tool, "list", str(tmp_path / "m.ckpt")], check=True, capture_output=True, text=True).stdout
    assert "RNNwavefunction/wf_dense/kernel" in out and "(6, 2)" in out
    subprocess.run([sys.executable, tool, "npz", str(tmp_path / "m.ckpt"), str(tmp_path / "b.npz")], check=True, capture_output=True)
    back = P.load_npz(tmp_path / "b.npz")
    assert set(back) == set(prm) and all(np.array_equal(back[k], prm[k]) for k in prm)
    import re
    assert all(re.fullmatch(r"[A-Za-z0-9.][A-Za-z0-9_.\\/>-]*", k) for k in back)        # TF's variable-name grammar
