"""TFRecord persistence of dataset elements (reference: chambers/data/tf_record.py), without TensorFlow.

The reference stores every tensor of an element as `tf.io.serialize_tensor(t)` (a serialized TensorProto) plus its dtype enum and
shape inside a `tf.train.Example`, one Example per TFRecord record.  The same bytes are produced here by hand: the protobuf wire
format of Example / Features / Feature / BytesList / Int64List, TensorProto / TensorShapeProto (the field numbers are part of
TensorFlow's public .proto files [UPSTREAM-RECALLED]), and the TFRecord framing
    uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)        (little endian)
with CRC-32C (Castagnoli) and TensorFlow's mask `rotr(crc, 15) + 0xa282ead8`.  Files written here are meant to be readable by
`tf.data.TFRecordDataset` + the reference's `tfrecord_to_dataset`, and vice versa for numeric tensors; there is no TensorFlow in
this image to cross-check against, so the format is pinned by the CRC-32C check vectors of RFC 3720 and by round trips only.
"""
import struct

import numpy as np

from .dataset import Dataset

# tensorflow/core/framework/types.proto
_DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.uint8): 4, np.dtype(np.int16): 5,
       np.dtype(np.int8): 6, np.dtype(np.int64): 9, np.dtype(np.bool_): 10, np.dtype(np.uint16): 17, np.dtype(np.float16): 19,
       np.dtype(np.uint32): 22, np.dtype(np.uint64): 23}
_NP = {v: k for k, v in _DT.items()}
DT_STRING = 7


# ---- CRC-32C, slicing-by-8 ---------------------------------------------------------------------
def _make_tables():
    poly = 0x82F63B78
    t0 = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t0.append(c)
    tables = [t0]
    for k in range(1, 8):
        prev = tables[k - 1]
        tables.append([(prev[n] >> 8) ^ t0[prev[n] & 0xFF] for n in range(256)])
    return tables


_T = _make_tables()


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli, reflected, init / xorout 0xffffffff) of a bytes-like object."""
    t0, t1, t2, t3, t4, t5, t6, t7 = _T
    mv = memoryview(data).cast("B")
    n = len(mv)
    crc ^= 0xFFFFFFFF
    n8 = n & ~7
    if n8:
        for lo, hi in struct.iter_unpack("<II", mv[:n8]):
            lo ^= crc
            crc = (t7[lo & 0xFF] ^ t6[(lo >> 8) & 0xFF] ^ t5[(lo >> 16) & 0xFF] ^ t4[lo >> 24] ^
                   t3[hi & 0xFF] ^ t2[(hi >> 8) & 0xFF] ^ t1[(hi >> 16) & 0xFF] ^ t0[hi >> 24])
    for b in mv[n8:]:
        crc = (crc >> 8) ^ t0[(crc ^ b) & 0xFF]
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire format ------------------------------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1                       # negative int64 -> 10-byte two's complement varint
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _field_bytes(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _field_varint(num, value):
    return _varint(num << 3) + _varint(value)


def _parse(buf):
    """[(field number, wire type, value)] of one message; length-delimited values stay bytes."""
    buf = bytes(buf)
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((num, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


# ---- TensorProto (tensorflow/core/framework/tensor.proto: dtype = 1, tensor_shape = 2, tensor_content = 4, string_val = 8;
#      tensor_shape.proto: dim = 2 {size = 1}) ---------------------------------------------------------------------------------
def serialize_tensor(t):
    """tf.io.serialize_tensor of a numpy array / scalar / bytes."""
    if isinstance(t, (bytes, str)):
        s = t.encode() if isinstance(t, str) else t
        return _field_varint(1, DT_STRING) + _field_bytes(2, b"") + _field_bytes(8, s)
    a = np.asarray(t)
    if a.dtype not in _DT:
        raise ValueError("unsupported dtype %s" % a.dtype)
    shape = b"".join(_field_bytes(2, _field_varint(1, d) if d else b"") for d in a.shape)     # proto3 omits zero scalars
    return _field_varint(1, _DT[a.dtype]) + _field_bytes(2, shape) + _field_bytes(4, np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<")).tobytes())


def parse_tensor(raw, out_type=None):
    """tf.io.parse_tensor: numpy array (or bytes for DT_STRING scalars); `out_type` is checked like TF does."""
    dtype, dims, content, strings = None, [], b"", []
    for num, wt, v in _parse(raw):
        if num == 1:
            dtype = v
        elif num == 2:
            dims = [_signed64(next((x for n2, _w, x in _parse(d) if n2 == 1), 0)) for n1, _wt, d in _parse(v) if n1 == 2]
        elif num == 4:
            content = v
        elif num == 8:
            strings.append(v)
    if dtype == DT_STRING:
        if out_type is not None and out_type not in (bytes, DT_STRING):
            raise ValueError("type mismatch: tensor is DT_STRING")
        return strings[0] if not dims else np.array(strings, dtype=object).reshape(dims)
    if dtype not in _NP:
        raise ValueError("unsupported DataType enum %r" % (dtype,))
    np_dtype = _NP[dtype]
    if out_type is not None and np.dtype(out_type) != np_dtype:
        raise ValueError("type mismatch: tensor is %s, requested %s" % (np_dtype, np.dtype(out_type)))
    return np.frombuffer(content, dtype=np_dtype.newbyteorder("<")).astype(np_dtype).reshape(dims)


# ---- tf.train.Example (example.proto: features = 1; feature.proto: Features.feature = 1 map<string, Feature>;
#      Feature: bytes_list = 1, float_list = 2, int64_list = 3; *List.value = 1, numeric lists packed) -----------------------
def _bytes_feature(value):
    vals = value if isinstance(value, (list, tuple)) else [value]
    return _field_bytes(1, b"".join(_field_bytes(1, bytes(v)) for v in vals))


def _int_feature(value):
    vals = np.atleast_1d(np.asarray(value)).astype(np.int64).tolist()
    return _field_bytes(3, _field_bytes(1, b"".join(_varint(v) for v in vals)) if vals else b"")


def _float_feature(value):
    vals = np.atleast_1d(np.asarray(value)).astype("<f4")
    return _field_bytes(2, _field_bytes(1, vals.tobytes()) if vals.size else b"")


def _feature_to_example(feature):
    """Serialized tf.train.Example of {name: encoded Feature}; map entries in sorted key order (deterministic output)."""
    entries = b"".join(_field_bytes(1, _field_bytes(1, k.encode()) + _field_bytes(2, feature[k])) for k in sorted(feature))
    return _field_bytes(1, entries)


def _make_feature(tensors):
    """tf_record.py:37-52: tensor i -> t{i}_raw (serialized TensorProto), t{i}_dtype (DataType enum), t{i}_shape."""
    if not isinstance(tensors, (list, tuple)):
        tensors = (tensors,)
    feature = {}
    for i, t in enumerate(tensors):
        name = "t%d" % i
        if isinstance(t, (bytes, str)):
            dt, shape = DT_STRING, []
        else:
            t = np.asarray(t)
            dt, shape = _DT.get(t.dtype), list(t.shape)
            if dt is None:
                raise ValueError("unsupported dtype %s" % t.dtype)
        feature[name + "_raw"] = _bytes_feature(serialize_tensor(t))
        feature[name + "_dtype"] = _int_feature(dt)
        feature[name + "_shape"] = _int_feature(shape)
    return feature


def serialize_to_example(*args):
    """Element (one or more tensors) -> serialized tf.train.Example bytes (tf_record.py:74-82)."""
    return _feature_to_example(_make_feature(args))


def parse_example(raw):
    """{name: ('bytes' | 'int64' | 'float', [values])} of a serialized Example."""
    out = {}
    for num, _wt, feats in _parse(raw):
        if num != 1:
            continue
        for n1, _w1, entry in _parse(feats):
            if n1 != 1:
                continue
            key, feat = None, b""
            for n2, _w2, v in _parse(entry):
                if n2 == 1:
                    key = v.decode()
                elif n2 == 2:
                    feat = v
            for n3, _w3, lst in _parse(feat):
                if n3 == 1:
                    out[key] = ("bytes", [v for n4, _w4, v in _parse(lst) if n4 == 1])
                elif n3 == 3:
                    vals = []
                    for n4, w4, v in _parse(lst):
                        if n4 != 1:
                            continue
                        if w4 == 2:                      # packed
                            pos = 0
                            while pos < len(v):
                                x, pos = _read_varint(v, pos)
                                vals.append(_signed64(x))
                        else:
                            vals.append(_signed64(v))
                    out[key] = ("int64", vals)
                elif n3 == 2:
                    vals = []
                    for n4, w4, v in _parse(lst):
                        if n4 == 1:
                            vals.extend(np.frombuffer(v, dtype="<f4").tolist())
                    out[key] = ("float", vals)
    return out


def _get_tensor_ids(dictionary):
    """tf_record.py:30-34 (sorted as strings, like the reference: 't10' sorts before 't2')."""
    return sorted({k.split("_")[0] for k in dictionary})


def _make_feature_deserialize_fn(feature, set_shape=False, set_dimension=False):
    """tf_record.py:85-121.  `feature` = parsed first example; the returned function maps serialized Example bytes to the tensor
    (one) or tuple of tensors (several).  With set_shape the stored shape of every later element must equal the first
    element's (TensorFlow raises at run time when set_shape's static shape is violated); set_dimension pins the rank only."""
    ids = _get_tensor_ids(feature)
    dtypes = [feature[t + "_dtype"][1][0] for t in ids]
    shapes = [list(feature[t + "_shape"][1]) for t in ids]

    def deserialize_fn(x):
        ex = parse_example(x)
        tensors = []
        for t, dt, shp in zip(ids, dtypes, shapes):
            if t + "_raw" not in ex:
                raise ValueError("feature %s_raw is missing from the example" % t)
            arr = parse_tensor(ex[t + "_raw"][1][0], out_type=DT_STRING if dt == DT_STRING else _NP[dt])
            got = list(np.shape(arr)) if not isinstance(arr, bytes) else []
            if set_shape and got != shp:
                raise ValueError("element shape %s differs from the shape %s the dataset was opened with" % (got, shp))
            if set_dimension and not set_shape and len(got) != len(shp):
                raise ValueError("element rank %d differs from rank %d" % (len(got), len(shp)))
            tensors.append(arr)
        return tensors[0] if len(tensors) == 1 else tuple(tensors)

    return deserialize_fn


def make_dataset_deserialize_fn(dataset, set_shape=False, set_dimension=False):
    """tf_record.py:124-133: derive names / dtypes / shapes from the first serialized element of `dataset`."""
    sample = next(iter(dataset))
    sample = sample[0] if isinstance(sample, tuple) else sample
    return _make_feature_deserialize_fn(parse_example(sample), set_shape=set_shape, set_dimension=set_dimension)


# ---- TFRecord files -------------------------------------------------------------------------------
class TFRecordWriter:
    def __init__(self, path):
        self._f = open(path, "wb")

    def write(self, record):
        record = bytes(record)
        head = struct.pack("<Q", len(record))
        self._f.write(head + struct.pack("<I", masked_crc32c(head)) + record + struct.pack("<I", masked_crc32c(record)))

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def tfrecord_iterator(path, check_crc=True):
    """Records of one TFRecord file; a truncated or corrupted file raises ValueError (TensorFlow: DataLossError)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError("truncated record header in %s" % path)
            (length,), (hcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if check_crc and masked_crc32c(head[:8]) != hcrc:
                raise ValueError("corrupted record length in %s" % path)
            body = f.read(length + 4)
            if len(body) < length + 4:
                raise ValueError("truncated record in %s" % path)
            data = body[:length]
            if check_crc and masked_crc32c(data) != struct.unpack("<I", body[length:])[0]:
                raise ValueError("corrupted record data in %s" % path)
            yield data


def TFRecordDataset(paths):
    """Dataset of the raw records of one file or of several files read one after the other."""
    paths = [paths] if isinstance(paths, (str, bytes)) or hasattr(paths, "__fspath__") else list(paths)

    def gen():
        for p in paths:
            for rec in tfrecord_iterator(p):
                yield (rec,)

    return Dataset(gen)


def dataset_to_tfrecord(dataset, path):
    """Write every element of `dataset` as one Example record (tf_record.py:136-140)."""
    with TFRecordWriter(path) as w:
        for e in dataset:
            w.write(serialize_to_example(*(e if isinstance(e, tuple) else (e,))))


def tfrecord_to_dataset(paths, set_shape=True, set_dimension=False):
    """tf_record.py:143-150: elements come back as the tensor (one) or the tuple of tensors (several) that were written."""
    td = TFRecordDataset(paths)
    fn = make_dataset_deserialize_fn(td, set_shape=set_shape, set_dimension=set_dimension)

    def gen():
        for (rec,) in td:
            out = fn(rec)
            yield out if isinstance(out, tuple) else (out,)

    return Dataset(gen)
