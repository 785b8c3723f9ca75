"""CPU tier: TFRecord persistence without TensorFlow (chambers_amd.data.tf_record / persist; reference: chambers/data/tf_record.py,
persist.py, tests test_units/data/test_tf_record.py).  The CRC-32C implementation is pinned by the RFC 3720 (iSCSI) check
vectors, the protobuf encoders by hand-assembled byte strings of the public wire format, the rest by the round trips the
reference's own tests perform (serialize -> deserialize over the class-interleaved MNIST sample)."""
import os
import struct

import numpy as np
import pytest

from chambers_amd.data import InterleaveImageClassDataset, match_nested_set
from chambers_amd.data import persist, tf_record as T
from chambers_amd.data.dataset import Dataset

MNIST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_data", "mnist", "train")


def _td():
    dirs = sorted(match_nested_set(MNIST))
    return InterleaveImageClassDataset(class_dirs=dirs, labels=list(range(10)), class_cycle_length=5, images_per_block=2, image_channels=3,
                                       block_bound=False, sample_block_random=False, shuffle=False, reshuffle_iteration=False,
                                       buffer_size=1024, seed=None, repeats=None)


def test_crc32c_check_vectors():
    assert T.crc32c(b"123456789") == 0xE3069283                      # the CRC catalogue's check value for CRC-32C
    assert T.crc32c(bytes(32)) == 0x8A9136AA                         # RFC 3720 B.4: 32 bytes of zeros
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43                      # 32 bytes of ones
    assert T.crc32c(bytes(range(32))) == 0x46DD794E                  # incrementing
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C          # decrementing
    data = np.random.default_rng(0).integers(0, 256, 1000, dtype=np.uint8).tobytes()
    bytewise = 0xFFFFFFFF
    for b in data:                                                   # the slicing-by-8 path against the plain byte loop
        bytewise = (bytewise >> 8) ^ T._T[0][(bytewise ^ b) & 0xFF]
    assert T.crc32c(data) == bytewise ^ 0xFFFFFFFF
    assert T.crc32c(b"") == 0


def test_wire_format_of_small_messages():
    # TensorProto of int32 [[1, 2]]: dtype DT_INT32 (3); tensor_shape {dim {size: 1} dim {size: 2}}; tensor_content 8 bytes
    raw = T.serialize_tensor(np.array([[1, 2]], dtype=np.int32))
    assert raw == bytes([0x08, 0x03, 0x12, 0x08, 0x12, 0x02, 0x08, 0x01, 0x12, 0x02, 0x08, 0x02, 0x22, 0x08]) + struct.pack("<ii", 1, 2)
    # scalar int64 7: empty shape message, 8 content bytes
    assert T.serialize_tensor(np.int64(7)) == bytes([0x08, 0x09, 0x12, 0x00, 0x22, 0x08]) + struct.pack("<q", 7)
    # Example {features {feature {key: "a" value {int64_list {value: [3]}}}}}
    ex = T._feature_to_example({"a": T._int_feature(3)})
    assert ex == bytes([0x0A, 0x0C, 0x0A, 0x0A, 0x0A, 0x01, 0x61, 0x12, 0x05, 0x1A, 0x03, 0x0A, 0x01, 0x03])
    assert T.parse_example(ex) == {"a": ("int64", [3])}
    neg = T.parse_example(T._feature_to_example({"n": T._int_feature([-1, 300])}))
    assert neg == {"n": ("int64", [-1, 300])}                        # negative int64: 10-byte varint
    fl = T.parse_example(T._feature_to_example({"f": T._float_feature([0.5, -2.0])}))
    assert fl == {"f": ("float", [0.5, -2.0])}
    assert T._varint(300) == b"\xac\x02"


@pytest.mark.parametrize("arr", [np.arange(24, dtype=np.uint8).reshape(2, 3, 4), np.float32(1.5), np.zeros((0, 3), np.float64),
                                 np.array([True, False]), np.arange(5, dtype=np.int64) - 2, np.ones((2, 2), np.float16)])
def test_tensor_round_trip(arr):
    back = T.parse_tensor(T.serialize_tensor(arr), out_type=np.asarray(arr).dtype)
    assert back.dtype == np.asarray(arr).dtype and back.shape == np.asarray(arr).shape
    np.testing.assert_array_equal(back, arr)
    with pytest.raises(ValueError):
        T.parse_tensor(T.serialize_tensor(arr), out_type=np.complex64 if np.asarray(arr).dtype != np.complex64 else np.uint8)


def test_string_tensor_and_unsupported_dtype():
    assert T.parse_tensor(T.serialize_tensor(b"abc")) == b"abc"
    with pytest.raises(ValueError):
        T.serialize_tensor(np.zeros(2, np.complex64))


# ---- the reference's round trips (test_units/data/test_tf_record.py: test_serialize_deserialize0..2)
def test_serialize_deserialize_pairs():
    td = _td()
    x, y = next(iter(td))
    ser = td.map(T.serialize_to_example)
    de = ser.map(T.make_dataset_deserialize_fn(ser))
    xd, yd = next(iter(de))
    np.testing.assert_array_equal(x, xd)
    assert y == yd and xd.dtype == np.uint8 and yd.dtype == np.int64


def test_serialize_deserialize_three_tensors_and_single():
    td = _td().map(lambda x, y: (x, x.astype(np.float32), y))
    a = next(iter(td))
    ser = td.map(T.serialize_to_example)
    b = next(iter(ser.map(T.make_dataset_deserialize_fn(ser))))
    assert len(b) == 3 and b[1].dtype == np.float32
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    single = _td().map(lambda x, y: x)
    ser = single.map(T.serialize_to_example)
    (xd,) = next(iter(ser.map(T.make_dataset_deserialize_fn(ser))))
    np.testing.assert_array_equal(next(iter(single))[0], xd)


def test_shape_pinning():
    rng = np.random.default_rng(0)
    elems = [(rng.integers(0, 255, size=(s, s, 3), dtype=np.uint8), np.int64(k)) for k, s in enumerate((16, 16, 24))]
    td = Dataset(lambda: iter(elems))
    ser = td.map(T.serialize_to_example)
    free = list(ser.map(T.make_dataset_deserialize_fn(ser)))                          # no static shape: any size passes
    assert [e[0].shape for e in free] == [(16, 16, 3), (16, 16, 3), (24, 24, 3)]
    assert len(list(ser.map(T.make_dataset_deserialize_fn(ser, set_dimension=True)))) == 3
    with pytest.raises(ValueError):                                                   # static shape of the first element
        list(ser.map(T.make_dataset_deserialize_fn(ser, set_shape=True)))


def test_tfrecord_file_round_trip_and_corruption(tmp_path):
    td = _td()
    path = str(tmp_path / "mnist.tfrecord")
    T.dataset_to_tfrecord(td, path)
    back = list(T.tfrecord_to_dataset(path))
    orig = list(td)
    assert len(back) == len(orig) == 30
    for (x, y), (xd, yd) in zip(orig, back):
        np.testing.assert_array_equal(x, xd)
        assert y == yd
    # framing: uint64 length | masked crc | data | masked crc
    blob = open(path, "rb").read()
    (ln,) = struct.unpack("<Q", blob[:8])
    assert struct.unpack("<I", blob[8:12])[0] == T.masked_crc32c(blob[:8])
    assert struct.unpack("<I", blob[12 + ln:16 + ln])[0] == T.masked_crc32c(blob[12:12 + ln])
    two = list(T.tfrecord_to_dataset([path, path]))
    assert len(two) == 60
    bad = bytearray(blob)
    bad[40] ^= 0x01
    open(path, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        list(T.tfrecord_to_dataset(path))
    open(path, "wb").write(blob[:-3])
    with pytest.raises(ValueError):
        list(T.tfrecord_to_dataset(path))


@pytest.mark.parametrize("n_files", [1, 3])
def test_save_and_load_dataset(tmp_path, n_files):
    td = _td()
    path = str(tmp_path / "saved")
    persist.save_dataset(td, path, n_files=n_files)
    meta = persist._load_dataset_metadata(os.path.join(path, "dataset.metadata"))
    assert meta["enumerated"] == (n_files > 1) and meta["n_elements"] == 30
    assert meta["element_spec"] == [{"shape": [28, 28, 3], "dtype": 4, "name": None}, {"shape": [], "dtype": 9, "name": None}]
    assert sorted(f for f in os.listdir(path) if f.endswith(".tfrecord")) == ["shard-%05d-of-%05d.tfrecord" % (k, n_files) for k in range(n_files)]
    back = list(persist.load_dataset(path))
    assert [int(y) for _x, y in back] == [int(y) for _x, y in td]                     # original order across the shards
    np.testing.assert_array_equal(back[7][0], list(td)[7][0])
