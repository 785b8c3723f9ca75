"""Fixture generator for tests/test_hdf5_lite.py - run with an interpreter that HAS h5py (this image: /opt/conda/bin/python3.9,
h5py 3.3.0 / libhdf5 1.10.6); the test-suite interpreter has none, which is why chambers_amd/utils/hdf5_lite.py exists.

    /opt/conda/bin/python3.9 tests/golden/make_h5_golden.py

Writes real HDF5 files in the layout keras `Model.save_weights(path.h5)` produces (keras/saving/hdf5_format.py,
save_weights_to_hdf5_group + save_attributes_to_hdf5_group, restated): root attributes `layer_names` (numpy array of byte strings),
`backend`, `keras_version` (python bytes -> variable-length strings); one group per layer with attribute `weight_names`; one dataset
per weight, named by the variable name (its '/' make nested groups).  Values come from numpy's PCG64 with a fixed seed, so the
test regenerates the expected arrays itself.
  keras_weights_vit_tiny.h5   a 1-block ViT (patch 8, D = 64, 1 head, ff 128, 16x16 images, 10 classes) in the reference's
                              variable names (encoder/encoder_layer/multi_head_attention/w_query:0, ...)
  keras_weights_fixedlen.h5   attributes as fixed-length byte-string arrays (h5py 2.x style)
  keras_weights_misc.h5       the same mechanism on edge cases: a layer without weights, a scalar weight, float64 and int32 datasets,
                              a weight list long enough for chunked attributes (weight_names0, weight_names1, ...), a full-model file
                              layout (`model_weights` group)
"""
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HDF5_OBJECT_HEADER_LIMIT = 64512


def save_attributes(group, name, data):
    """keras save_attributes_to_hdf5_group: one attribute, or name0, name1, ... when it would not fit an object header."""
    bad = [x for x in data if len(x) > HDF5_OBJECT_HEADER_LIMIT]
    assert not bad
    data_npy = np.asarray(data)
    num_chunks = 1
    chunked = np.array_split(data_npy, num_chunks)
    while any(x.nbytes > HDF5_OBJECT_HEADER_LIMIT for x in chunked):
        num_chunks += 1
        chunked = np.array_split(data_npy, num_chunks)
    if num_chunks > 1:
        for i, chunk in enumerate(chunked):
            group.attrs["%s%d" % (name, i)] = chunk
    else:
        group.attrs[name] = data


def save_weights(f, layers, fixed_length=False):
    """layers: [(layer name, [(weight name, array)])].  fixed_length: attributes as numpy 'S' arrays (what h5py 2.x made of a list of
    bytes; h5py 3 writes variable-length strings)."""
    if fixed_length:
        f.attrs["layer_names"] = np.array([n.encode("utf8") for n, _ in layers])
        f.attrs["backend"] = np.bytes_("tensorflow")
        for lname, weights in layers:
            g = f.create_group(lname)
            g.attrs["weight_names"] = np.array([n.encode("utf8") for n, _ in weights], dtype="S64") if weights else np.zeros((0,), dtype="S1")
            for wname, val in weights:
                g.create_dataset(wname, data=val)
        return
    save_attributes(f, "layer_names", [n.encode("utf8") for n, _ in layers])
    f.attrs["backend"] = "tensorflow".encode("utf8")
    f.attrs["keras_version"] = "2.6.0".encode("utf8")
    for lname, weights in layers:
        g = f.create_group(lname)
        save_attributes(g, "weight_names", [n.encode("utf8") for n, _ in weights])
        for wname, val in weights:
            d = g.create_dataset(wname, val.shape, dtype=val.dtype)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val


def vit_tiny_layers(rng):
    d, ff, p, n_tok, classes = 64, 128, 8, 5, 10
    r = lambda *s: rng.standard_normal(s).astype(np.float32)      # noqa: E731
    enc = "encoder/encoder_layer/"
    mha = enc + "multi_head_attention/"
    return [
        ("input_1", []),
        ("patch_embeddings", [("embedding/kernel:0", r(p, p, 3, d)), ("embedding/bias:0", r(d))]),
        ("add_cls_token", [("add_cls_token/embeddings:0", r(1, d))]),
        ("pos_embedding", [("pos_embedding/embeddings:0", r(n_tok, d))]),
        ("dropout", []),
        ("encoder", [(mha + "w_query:0", r(d, 1, 64)), (mha + "b_query:0", r(1, 1, 64)), (mha + "w_value:0", r(d, 1, 64)),
                     (mha + "b_value:0", r(1, 1, 64)), (mha + "w_key:0", r(d, 1, 64)), (mha + "b_key:0", r(1, 1, 64)),
                     (mha + "w_projection:0", r(1, d, 64)), (mha + "b_projection:0", r(1, d)),
                     (enc + "layer_normalization/gamma:0", r(d)), (enc + "layer_normalization/beta:0", r(d)),
                     (enc + "dense/kernel:0", r(d, ff)), (enc + "dense/bias:0", r(ff)),
                     (enc + "dense_1/kernel:0", r(ff, d)), (enc + "dense_1/bias:0", r(d)),
                     (enc + "layer_normalization_1/gamma:0", r(d)), (enc + "layer_normalization_1/beta:0", r(d)),
                     ("encoder/layer_normalization_2/gamma:0", r(d)), ("encoder/layer_normalization_2/beta:0", r(d))]),
        ("cls_embedding", []),
        ("predictions", [("predictions/kernel:0", r(d, classes)), ("predictions/bias:0", r(classes))]),
        ("cast_float32", []),
    ]


def misc_layers(rng):
    pad = "x" * 96          # 620 names of 110 bytes: 68 KB as a numpy 'S' array, more than one object header holds -> weight_names0, weight_names1
    many = [("block/w_%04d_%s:0" % (i, pad), rng.standard_normal((2, 3)).astype(np.float32)) for i in range(620)]
    return [
        ("no_weights", []),
        ("scalars", [("scalars/step:0", np.asarray(7, dtype=np.int32)), ("scalars/scale:0", np.asarray(0.125, dtype=np.float64))]),
        ("mixed", [("mixed/f64:0", rng.standard_normal((3, 4))), ("mixed/i32:0", rng.integers(-5, 5, size=(6,)).astype(np.int32)),
                   ("mixed/f16:0", rng.standard_normal((2, 2)).astype(np.float16))]),
        ("many", many),
    ]


def main():
    with h5py.File(os.path.join(HERE, "keras_weights_vit_tiny.h5"), "w") as f:
        save_weights(f, vit_tiny_layers(np.random.Generator(np.random.PCG64(2021))))
    with h5py.File(os.path.join(HERE, "keras_weights_misc.h5"), "w") as f:
        f.attrs["keras_version"] = "2.6.0".encode("utf8")
        f.attrs["model_config"] = '{"class_name": "Functional"}'
        save_weights(f.create_group("model_weights"), misc_layers(np.random.Generator(np.random.PCG64(2022))))
    with h5py.File(os.path.join(HERE, "keras_weights_fixedlen.h5"), "w") as f:
        rng = np.random.Generator(np.random.PCG64(2023))
        save_weights(f, [("dense", [("dense/kernel:0", rng.standard_normal((4, 5)).astype(np.float32)), ("dense/bias:0", rng.standard_normal(5).astype(np.float32))]),
                         ("empty", [])], fixed_length=True)
    print("h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version, "->", [n for n in os.listdir(HERE) if n.endswith(".h5")])


if __name__ == "__main__":
    sys.exit(main())
