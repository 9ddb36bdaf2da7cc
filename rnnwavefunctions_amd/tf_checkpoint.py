"""TensorFlow checkpoint (V2 "tensor bundle") reader and writer in plain Python/NumPy - no TensorFlow needed.

The reference keeps its weights with ``tf.train.Saver`` (1DTFIM/TrainingRNN_1DTFIM.py:166 ``saver=tf.train.Saver()``,
``:219 saver.save(sess, path+'/'+filename)``, ``:172-183`` the commented restore branch; the same in
J1J2/TrainingRNN_J1J2.py:207 and both 2DTFIM folders).  TF >= 0.12 Savers write format V2:

    <prefix>.index                   an SSTable (LevelDB table format, tensorflow/core/lib/io/table*.cc): sorted
                                     key -> value records; key "" holds a BundleHeaderProto, every other key is a
                                     variable name whose value is a BundleEntryProto (dtype, shape, shard, offset,
                                     size, masked crc32c of the tensor bytes)
    <prefix>.data-0000i-of-0000n     the tensors' raw little-endian bytes, back to back
    checkpoint                       (optional) text proto naming the latest prefix in the directory

This module restates that published format (tensor_bundle.proto, tensor_shape.proto, types.proto, versions.proto,
leveldb's table_format.md) - it is the drop-in for the Saver on this path: `write_checkpoint` produces files a TF1
``saver.restore`` accepts for the variables it holds, `read_checkpoint` loads what the reference's ``saver.save`` wrote.
No file produced by real TensorFlow is available in the build container (TensorFlow is not installable here), so the
byte-level agreement with TF's own writer is PARITY UNPINNED; what is tested: write -> read round trips bit-exact,
CRC-32C and varint known-answer vectors, the table footer's magic number, prefix-compressed and multi-block tables,
snappy-compressed blocks (TF's BundleWriter does not compress, LevelDB tables in general may).  A checkpoint written
by TF 1.13 elsewhere can be dropped into tests/golden/ later to pin it (tests/test_tf_checkpoint.py picks it up).
"""
import os
import struct
from collections import OrderedDict

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57          # leveldb table_format.md: footer magic (little-endian fixed64)
FOOTER_BYTES = 48                         # 2 BlockHandles padded to 40 bytes + the magic
BLOCK_TRAILER_BYTES = 5                   # compression type (1) + masked crc32c (4)
BLOCK_RESTART_INTERVAL = 16               # table::Options default (table_options.h)
BLOCK_SIZE = 262144                       # table::Options default block_size
TENSOR_BUNDLE_VERSION = 1                 # kTensorBundleVersion (tensor_bundle.cc)

# types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT16, DT_INT8, DT_INT64, DT_BOOL = 1, 2, 3, 4, 5, 6, 9, 10
_NP_OF_DT = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_UINT8: np.uint8, DT_INT16: np.int16,
             DT_INT8: np.int8, DT_INT64: np.int64, DT_BOOL: np.bool_}
_DT_OF_NP = {np.dtype(v): k for k, v in _NP_OF_DT.items()}


class CheckpointError(ValueError):
    pass


# ---- CRC-32C (Castagnoli), as tensorflow/core/lib/hash/crc32c.h, with its mask ------------------------------------
def _make_crc_table():
    poly = 0x82F63B78
    t = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t[i] = c
    return [int(x) for x in t]


_CRC_TABLE = _make_crc_table()


def crc32c(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in bytes(data):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc):
    """crc32c::Mask - rotate right by 15 bits and add a constant (stored CRCs are masked)."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- varints and the four protobuf messages the format uses ---------------------------------------------------------
def put_varint(n):
    if n < 0:
        n += 1 << 64
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def get_varint(buf, pos):
    shift = result = 0
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
            raise CheckpointError("varint longer than 10 bytes")


def _fields(buf):
    """Iterate (field number, wire type, value) over a serialized protobuf message."""
    pos = 0
    while pos < len(buf):
        key, pos = get_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise CheckpointError("truncated length-delimited field")
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def encode_header(num_shards=1):
    """BundleHeaderProto{num_shards=1; endianness=2 (LITTLE=0, omitted); version=3 {producer=1}}"""
    version = b"\x08" + put_varint(TENSOR_BUNDLE_VERSION)
    return b"\x08" + put_varint(num_shards) + b"\x1a" + put_varint(len(version)) + version


def decode_header(buf):
    h = {"num_shards": 0, "endianness": 0, "producer": 0, "min_consumer": 0}
    for num, _, v in _fields(buf):
        if num == 1:
            h["num_shards"] = v
        elif num == 2:
            h["endianness"] = v
        elif num == 3:
            for n2, _, v2 in _fields(v):
                if n2 == 1:
                    h["producer"] = v2
                elif n2 == 2:
                    h["min_consumer"] = v2
    return h


def encode_entry(dtype, shape, shard_id, offset, size, crc):
    """BundleEntryProto{dtype=1; shape=2 {dim=2 {size=1}}; shard_id=3; offset=4; size=5; crc32c=6 (fixed32)}"""
    dims = b"".join(b"\x12" + put_varint(len(d)) + d for d in (b"\x08" + put_varint(int(s)) for s in shape))
    out = b"\x08" + put_varint(dtype) + b"\x12" + put_varint(len(dims)) + dims
    if shard_id:
        out += b"\x18" + put_varint(shard_id)
    if offset:
        out += b"\x20" + put_varint(offset)
    if size:
        out += b"\x28" + put_varint(size)
    return out + b"\x35" + struct.pack("<I", crc)


def decode_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0, "slices": 0}
    for num, _, v in _fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:
            for n2, _, v2 in _fields(v):
                if n2 == 2:
                    size = 0
                    for n3, _, v3 in _fields(v2):
                        if n3 == 1:
                            size = _signed64(v3)
                    e["shape"].append(size)
                elif n2 == 3 and v2:
                    raise CheckpointError("tensor of unknown rank in a checkpoint")
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["slices"] += 1
    return e


# ---- snappy (raw format) decompression: LevelDB tables may hold compressed blocks -----------------------------------
def snappy_decompress(buf):
    n, pos = get_varint(buf, 0)
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
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError("corrupt snappy block")
        for _ in range(ln):                      # byte-wise: the copy may overlap its own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError("snappy block decompressed to %d bytes, header says %d" % (len(out), n))
    return bytes(out)


# ---- SSTable (leveldb table format) ----------------------------------------------------------------------------------
def _block_handle(offset, size):
    return put_varint(offset) + put_varint(size)


def _short_successor(key):
    """BytewiseComparator::FindShortSuccessor: first byte that can be incremented, incremented, rest dropped."""
    k = bytearray(key)
    for i, b in enumerate(k):
        if b != 0xFF:
            k[i] = b + 1
            return bytes(k[:i + 1])
    return bytes(k)


def _shortest_separator(start, limit):
    """BytewiseComparator::FindShortestSeparator(start, limit): a short key in [start, limit)."""
    n = min(len(start), len(limit))
    d = 0
    while d < n and start[d] == limit[d]:
        d += 1
    if d < n and start[d] < 0xFF and start[d] + 1 < limit[d]:
        return start[:d] + bytes([start[d] + 1])
    return start


class _BlockBuilder:
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b""

    def add(self, key, value):
        shared = 0
        if self.counter < self.interval:
            n = min(len(self.last_key), len(key))
            while shared < n and self.last_key[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += put_varint(shared) + put_varint(len(key) - shared) + put_varint(len(value))
        self.buf += key[shared:] + value
        self.last_key = key
        self.counter += 1

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_table(path, items, block_size=BLOCK_SIZE, restart_interval=BLOCK_RESTART_INTERVAL):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order -> an uncompressed LevelDB table
    (TableBuilder with Options{compression=kNoCompression}, as BundleWriter::Finish builds the .index file)."""
    out = bytearray()

    def emit(contents):
        handle = (len(out), len(contents))
        out.extend(contents)
        out.extend(b"\x00" + struct.pack("<I", mask_crc(crc32c(contents + b"\x00"))))
        return handle

    data, index = _BlockBuilder(restart_interval), _BlockBuilder(1)
    pending = None                        # (last key of the finished data block, its handle)
    last = None
    for key, value in items:
        key, value = bytes(key), bytes(value)
        if last is not None and key <= last:
            raise CheckpointError("table keys must be strictly increasing (%r after %r)" % (key, last))
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), _block_handle(*pending[1]))
            pending = None
        data.add(key, value)
        last = key
        if data.size_estimate() >= block_size:
            pending = (last, emit(data.finish()))
            data = _BlockBuilder(restart_interval)
    if not data.empty():
        pending = (last, emit(data.finish()))
    if pending is not None:
        index.add(_short_successor(pending[0]), _block_handle(*pending[1]))
    metaindex = emit(_BlockBuilder(restart_interval).finish())       # no filter policy: an empty block
    index_handle = emit(index.finish())
    footer = _block_handle(*metaindex) + _block_handle(*index_handle)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(bytes(out))


def _read_block(buf, offset, size, verify):
    end = offset + size
    if end + BLOCK_TRAILER_BYTES > len(buf):
        raise CheckpointError("table block beyond the end of the file")
    contents, ctype = buf[offset:end], buf[end]
    if verify:
        stored = struct.unpack_from("<I", buf, end + 1)[0]
        if stored != mask_crc(crc32c(buf[offset:end + 1])):
            raise CheckpointError("table block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        contents = snappy_decompress(contents)
    elif ctype != 0:
        raise CheckpointError("unknown table block compression type %d" % ctype)
    return contents


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointError("table block too short")
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * nrestart
    if limit < 0:
        raise CheckpointError("corrupt restart array")
    pos, key = 0, b""
    while pos < limit:
        shared, pos = get_varint(block, pos)
        unshared, pos = get_varint(block, pos)
        vlen, pos = get_varint(block, pos)
        if shared > len(key) or pos + unshared + vlen > limit:
            raise CheckpointError("corrupt table entry")
        key = key[:shared] + bytes(block[pos:pos + unshared])
        pos += unshared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) records of a LevelDB table, in key order."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < FOOTER_BYTES:
        raise CheckpointError("%s: too short for a table footer" % path)
    footer = buf[-FOOTER_BYTES:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise CheckpointError("%s: not an SSTable (bad magic number); a V1 checkpoint or another file?" % path)
    pos = 0
    _, pos = get_varint(footer, pos)          # metaindex handle (filter blocks: not used here)
    _, pos = get_varint(footer, pos)
    ioff, pos = get_varint(footer, pos)
    isize, pos = get_varint(footer, pos)
    records = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, p2 = get_varint(handle, 0)
        bsize, _ = get_varint(handle, p2)
        records.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return records


# ---- the bundle -----------------------------------------------------------------------------------------------------
def _data_path(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def write_checkpoint(prefix, tensors, write_state_file=True):
    """Writes `tensors` ({variable name: ndarray}) as <prefix>.index + <prefix>.data-00000-of-00001.

    Tensors go into the data file in key order (the order SaveV2 receives them from tf.train.Saver, which sorts by
    name).  Also writes the directory's `checkpoint` state file, as tf.train.Saver.save does."""
    names = sorted(tensors, key=lambda k: k.encode())
    if "" in tensors:
        raise CheckpointError("the empty name is reserved for the bundle header")
    entries, offset = [], 0
    d = os.path.dirname(os.path.abspath(prefix))
    os.makedirs(d, exist_ok=True)
    with open(_data_path(prefix, 0, 1), "wb") as f:
        for name in names:
            a = np.asarray(tensors[name])
            if a.dtype not in _DT_OF_NP:
                raise CheckpointError("%s: dtype %s has no checkpoint encoding here" % (name, a.dtype))
            raw = np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            f.write(raw)
            entries.append((name.encode(), encode_entry(_DT_OF_NP[a.dtype], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + ".index", [(b"", encode_header(1))] + entries)
    if write_state_file:
        base = os.path.basename(prefix)
        with open(os.path.join(d, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return prefix


def list_variables(prefix):
    """[(name, shape, numpy dtype)] as tf.train.list_variables."""
    out = []
    for key, value in read_table(prefix + ".index"):
        if key == b"":
            continue
        e = decode_entry(value)
        out.append((key.decode(), tuple(e["shape"]), _NP_OF_DT.get(e["dtype"])))
    return out


def read_checkpoint(prefix, names=None, verify=True):
    """{variable name: ndarray} of a V2 checkpoint (every variable, or the listed ones)."""
    if not os.path.exists(prefix + ".index"):
        hint = " (a file by that exact name exists: V1 checkpoints are not supported)" if os.path.exists(prefix) else ""
        raise CheckpointError("no checkpoint index %s.index%s" % (prefix, hint))
    records = read_table(prefix + ".index", verify)
    if not records or records[0][0] != b"":
        raise CheckpointError("%s.index has no bundle header" % prefix)
    header = decode_header(records[0][1])
    if header["endianness"] != 0:
        raise CheckpointError("big-endian checkpoints are not supported")
    if header["min_consumer"] > TENSOR_BUNDLE_VERSION:
        raise CheckpointError("checkpoint needs tensor-bundle version %d" % header["min_consumer"])
    shards = {}
    out = OrderedDict()
    want = None if names is None else set(names)
    for key, value in records[1:]:
        name = key.decode()
        if want is not None and name not in want:
            continue
        e = decode_entry(value)
        if e["slices"]:
            raise CheckpointError("%s: partitioned (sliced) variables are not supported" % name)
        if e["dtype"] not in _NP_OF_DT:
            raise CheckpointError("%s: checkpoint dtype %d is not supported" % (name, e["dtype"]))
        if e["shard_id"] not in shards:
            with open(_data_path(prefix, e["shard_id"], max(header["num_shards"], 1)), "rb") as f:
                shards[e["shard_id"]] = f.read()
        raw = shards[e["shard_id"]][e["offset"]:e["offset"] + e["size"]]
        dt = np.dtype(_NP_OF_DT[e["dtype"]]).newbyteorder("<")
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if len(raw) != e["size"] or count * dt.itemsize != e["size"]:
            raise CheckpointError("%s: %d bytes in the data file for shape %s" % (name, len(raw), e["shape"]))
        if verify and mask_crc(crc32c(raw)) != e["crc32c"]:
            raise CheckpointError("%s: tensor checksum mismatch" % name)
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).astype(dt.newbyteorder("="))
    if want is not None and want - set(out):
        raise CheckpointError("not in checkpoint %s: %s" % (prefix, sorted(want - set(out))))
    return out


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by the directory's `checkpoint` state file, or None."""
    try:
        with open(os.path.join(directory, "checkpoint")) as f:
            for line in f:
                if line.startswith("model_checkpoint_path:"):
                    p = line.split(":", 1)[1].strip().strip('"')
                    return p if os.path.isabs(p) else os.path.join(directory, p)
    except OSError:
        pass
    return None


# ---- what tf.train.Saver() of the reference's training scripts holds besides the model ------------------------------
ADAM_SLOTS = ("Adam", "Adam_1")           # first / second moment slot names of tf.train.AdamOptimizer


def split_saver_variables(tensors):
    """Separates a Saver dump into (model variables, optimizer state).

    The reference's Saver is built after `optimizer.apply_gradients` (TrainingRNN_1DTFIM.py:163-166), so it also
    holds `<var>/Adam`, `<var>/Adam_1`, `beta1_power`, `beta2_power` (possibly under a scope prefix) and the
    un-named global step `Variable`.  Returns ({name: array}, {"m": {...}, "v": {...}, "beta1_power", "beta2_power",
    "global_step"}) with the slot dicts keyed by the model variable's name."""
    model, opt = OrderedDict(), {"m": OrderedDict(), "v": OrderedDict(), "beta1_power": None, "beta2_power": None,
                                 "global_step": None}
    for name, a in tensors.items():
        leaf = name.rsplit("/", 1)[-1]
        if leaf == "Adam":
            opt["m"][name[:-len("/Adam")]] = a
        elif leaf == "Adam_1":
            opt["v"][name[:-len("/Adam_1")]] = a
        elif leaf in ("beta1_power", "beta2_power"):
            opt[leaf] = float(a)
        elif leaf.split("_")[0] == "Variable" and a.ndim == 0 and np.issubdtype(a.dtype, np.integer):
            opt["global_step"] = int(a)
        elif leaf == "global_step" and a.ndim == 0:
            opt["global_step"] = int(a)
        else:
            model[name] = a
    return model, opt
