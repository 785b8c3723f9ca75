"""CPU tier: chambers_amd.utils.hdf5_lite (the pure-Python reader of the HDF5 subset Keras weight files use) against REAL HDF5
files written by h5py 3.3 / libhdf5 1.10.6 in the layout of keras `Model.save_weights` (tests/golden/make_h5_golden.py: the
generator needs h5py, this interpreter has none).  The expected arrays are regenerated here from the generator's seeds."""
import os

import numpy as np
import pytest

from chambers_amd.utils import hdf5_lite as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _vit_tiny_expected():
    rng = np.random.Generator(np.random.PCG64(2021))
    d, ff, p, n_tok, classes = 64, 128, 8, 5, 10
    r = lambda *s: rng.standard_normal(s).astype(np.float32)      # noqa: E731
    enc = "encoder/encoder_layer/"
    mha = enc + "multi_head_attention/"
    order = [("embedding/kernel:0", (p, p, 3, d)), ("embedding/bias:0", (d,)), ("add_cls_token/embeddings:0", (1, d)),
             ("pos_embedding/embeddings:0", (n_tok, d)),
             (mha + "w_query:0", (d, 1, 64)), (mha + "b_query:0", (1, 1, 64)), (mha + "w_value:0", (d, 1, 64)), (mha + "b_value:0", (1, 1, 64)),
             (mha + "w_key:0", (d, 1, 64)), (mha + "b_key:0", (1, 1, 64)), (mha + "w_projection:0", (1, d, 64)), (mha + "b_projection:0", (1, d)),
             (enc + "layer_normalization/gamma:0", (d,)), (enc + "layer_normalization/beta:0", (d,)), (enc + "dense/kernel:0", (d, ff)),
             (enc + "dense/bias:0", (ff,)), (enc + "dense_1/kernel:0", (ff, d)), (enc + "dense_1/bias:0", (d,)),
             (enc + "layer_normalization_1/gamma:0", (d,)), (enc + "layer_normalization_1/beta:0", (d,)),
             ("encoder/layer_normalization_2/gamma:0", (d,)), ("encoder/layer_normalization_2/beta:0", (d,)),
             ("predictions/kernel:0", (d, classes)), ("predictions/bias:0", (classes,))]
    return [(n, r(*s)) for n, s in order]


def test_reads_a_keras_weight_file_written_by_libhdf5():
    weights, layout = H.load_keras_weights(os.path.join(GOLD, "keras_weights_vit_tiny.h5"))
    want = _vit_tiny_expected()
    assert [l for l, _ in layout] == ["input_1", "patch_embeddings", "add_cls_token", "pos_embedding", "dropout", "encoder", "cls_embedding",
                                      "predictions", "cast_float32"]
    assert [n for _l, names in layout for n in names] == [n for n, _ in want]
    for name, arr in want:
        assert weights[name].dtype == np.float32 and weights[name].shape == arr.shape
        np.testing.assert_array_equal(weights[name], arr)
    f = H.File(os.path.join(GOLD, "keras_weights_vit_tiny.h5"))
    assert f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == b"2.6.0"          # variable-length string scalars
    assert "encoder" in f and "nope" not in f
    assert f["encoder"]["encoder"]["encoder_layer"]["dense"].keys() == ["bias:0", "kernel:0"]    # '/' in a variable name = nested groups
    assert f["encoder/encoder/encoder_layer/dense/kernel:0"].shape == (64, 128)


def test_edge_cases_scalars_dtypes_chunked_attributes_model_weights_group():
    weights, layout = H.load_keras_weights(os.path.join(GOLD, "keras_weights_misc.h5"))       # a full-model file: weights under /model_weights
    rng = np.random.Generator(np.random.PCG64(2022))
    pad = "x" * 96
    many = [("block/w_%04d_%s:0" % (i, pad), rng.standard_normal((2, 3)).astype(np.float32)) for i in range(620)]
    f64 = rng.standard_normal((3, 4))
    i32 = rng.integers(-5, 5, size=(6,)).astype(np.int32)
    f16 = rng.standard_normal((2, 2)).astype(np.float16)
    assert dict(layout)["no_weights"] == [] and len(dict(layout)["many"]) == 620             # weight_names0 + weight_names1 re-joined
    assert float(weights["scalars/step:0"]) == 7.0 and float(weights["scalars/scale:0"]) == 0.125
    np.testing.assert_array_equal(weights["mixed/f64:0"], f64.astype(np.float32))
    np.testing.assert_array_equal(weights["mixed/i32:0"], i32.astype(np.float32))
    np.testing.assert_array_equal(weights["mixed/f16:0"], f16.astype(np.float32))
    for name, arr in many[::37]:
        np.testing.assert_array_equal(weights[name], arr)
    f = H.File(os.path.join(GOLD, "keras_weights_misc.h5"))
    g = f["model_weights"]["many"]
    assert "weight_names" not in g.attrs and {"weight_names0", "weight_names1"} <= set(g.attrs)
    raw = f["model_weights/mixed/mixed/f64:0"][()]
    assert raw.dtype == np.float64 and np.array_equal(raw, f64)


def test_fixed_length_string_attributes_and_errors(tmp_path):
    weights, layout = H.load_keras_weights(os.path.join(GOLD, "keras_weights_fixedlen.h5"))     # h5py 2.x style 'S' arrays
    rng = np.random.Generator(np.random.PCG64(2023))
    np.testing.assert_array_equal(weights["dense/kernel:0"], rng.standard_normal((4, 5)).astype(np.float32))
    np.testing.assert_array_equal(weights["dense/bias:0"], rng.standard_normal(5).astype(np.float32))
    assert layout == [("dense", ["dense/kernel:0", "dense/bias:0"]), ("empty", [])]
    bad = tmp_path / "x.h5"
    bad.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(H.Hdf5Error):
        H.File(str(bad))
    with pytest.raises(KeyError):
        H.File(os.path.join(GOLD, "keras_weights_fixedlen.h5"))["dense/missing"]


def test_vision_transformer_loads_a_keras_h5_weight_file():
    """`VisionTransformer(..., weights="file.h5")` / `model.load_weights("file.h5")` (vision_transformer.py:149-169): topological
    loading - the file's layers with weights against the model's, in order, each layer's arrays in `weight_names` order."""
    from chambers_amd.models.backbones.vision_transformer import VisionTransformer
    path = os.path.join(GOLD, "keras_weights_vit_tiny.h5")
    m = VisionTransformer(8, 64, 1, 1, 128, input_shape=(16, 16, 3), weights=path, classes=10)
    kw = m.keras_weights()
    want = dict(_vit_tiny_expected())
    enc, mha = "encoder/encoder_layer/", "encoder/encoder_layer/multi_head_attention/"
    pairs = {"patch_embeddings/embedding/kernel": "embedding/kernel:0", "patch_embeddings/embedding/bias": "embedding/bias:0",
             "add_cls_token/embeddings": "add_cls_token/embeddings:0", "pos_embedding/embeddings": "pos_embedding/embeddings:0",
             "encoder/layer_0/norm1/gamma": enc + "layer_normalization/gamma:0", "encoder/layer_0/norm1/beta": enc + "layer_normalization/beta:0",
             "encoder/layer_0/dense1/kernel": enc + "dense/kernel:0", "encoder/layer_0/dense1/bias": enc + "dense/bias:0",
             "encoder/layer_0/dense2/kernel": enc + "dense_1/kernel:0", "encoder/layer_0/dense2/bias": enc + "dense_1/bias:0",
             "encoder/layer_0/norm2/gamma": enc + "layer_normalization_1/gamma:0", "encoder/layer_0/norm2/beta": enc + "layer_normalization_1/beta:0",
             "encoder/norm/gamma": "encoder/layer_normalization_2/gamma:0", "encoder/norm/beta": "encoder/layer_normalization_2/beta:0",
             "predictions/kernel": "predictions/kernel:0", "predictions/bias": "predictions/bias:0"}
    for nm in ("w_query", "b_query", "w_value", "b_value", "w_key", "b_key", "w_projection", "b_projection"):
        pairs["encoder/layer_0/multi_head_attention/" + nm] = mha + nm + ":0"
    assert set(pairs) == set(kw)
    for mine, theirs in pairs.items():
        np.testing.assert_array_equal(kw[mine], want[theirs], err_msg=mine)
    m2 = VisionTransformer(8, 64, 2, 1, 128, input_shape=(16, 16, 3), weights=None, classes=10)      # another depth: counts differ
    with pytest.raises(ValueError, match="expects 34 weights|weights, but the saved weights have"):
        m2.load_weights(path)
    m3 = VisionTransformer(8, 64, 1, 1, 128, input_shape=(16, 16, 3), weights=None, include_top=False)
    with pytest.raises(ValueError, match="containing 5 layers into a model with 4 layers"):
        m3.load_weights(path)


# ---- the writer (Model.save_weights("x.h5")) ---------------------------------------------------------------------------------------
def _writer_case():
    rng = np.random.Generator(np.random.PCG64(77))
    return [("dense", [("dense/kernel:0", rng.standard_normal((5, 7)).astype(np.float32)), ("dense/bias:0", rng.standard_normal(7).astype(np.float32))]),
            ("empty", []),
            ("encoder", [("encoder/l%d/w:0" % i, rng.standard_normal((3,)).astype(np.float32)) for i in range(300)]     # > 256 links: two B-tree levels
             + [("encoder/s:0", np.float32(3.0)), ("encoder/d:0", rng.standard_normal((2, 2))), ("encoder/i:0", np.arange(6, dtype=np.int32).reshape(2, 3)),
                ("encoder/z:0", np.zeros((0, 4), np.float32))]),
            ("many", [("many/w%04d_%s:0" % (i, "x" * 60), np.float32(i)) for i in range(1200)])]                       # weight_names0, weight_names1


def test_writer_round_trip_through_the_reader(tmp_path):
    layers = _writer_case()
    path = str(tmp_path / "w.h5")
    H.save_keras_weights(path, layers)
    f = H.File(path)
    assert f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == b"2.6.0"
    assert [n.decode() for n in f.attrs["layer_names"]] == [l for l, _ in layers]
    assert sorted(f["many"].attrs) == ["weight_names0", "weight_names1"] and len(f["encoder/encoder"].keys()) == 304
    values, layout = H.load_keras_weights(path)
    assert layout == [(l, [n for n, _ in ws]) for l, ws in layers]
    for _l, ws in layers:
        for n, a in ws:
            np.testing.assert_array_equal(values[n], np.asarray(a, dtype=np.float32), err_msg=n)
    assert f["encoder/encoder/d:0"][()].dtype == np.float64 and f["encoder/encoder/i:0"][()].dtype == np.int32
    assert f["encoder/encoder/s:0"].shape == () and f["encoder/encoder/z:0"].shape == (0, 4)
    with pytest.raises(ValueError, match="duplicate"):
        H.save_keras_weights(path, [("a", [("a/w:0", np.zeros(2)), ("a/w:0", np.zeros(2))])])
    with pytest.raises(ValueError, match="runs through the dataset"):
        H.save_keras_weights(path, [("a", [("a/w", np.zeros(2)), ("a/w/x", np.zeros(2))])])


def _h5py_python():
    import shutil
    import subprocess
    for exe in ("/opt/conda/bin/python3.9", "/opt/conda/bin/python", shutil.which("python3.9") or ""):
        if exe and os.path.exists(exe):
            try:
                if subprocess.run([exe, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0:
                    return exe
            except (OSError, subprocess.TimeoutExpired):
                pass
    return None


def test_writer_output_is_read_by_libhdf5(tmp_path):
    """The written file through REAL h5py / libhdf5 (tests/golden/check_h5_with_h5py.py in the image's other interpreter): groups,
    chunked name attributes, two-level group B-trees, scalar / empty / float64 / int32 datasets all come back."""
    import hashlib
    import json
    import subprocess
    exe = _h5py_python()
    if exe is None:
        pytest.skip("no interpreter with h5py on this box")
    layers = _writer_case()
    path = str(tmp_path / "w.h5")
    H.save_keras_weights(path, layers)
    r = subprocess.run([exe, os.path.join(GOLD, "check_h5_with_h5py.py"), path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["attrs"]["/"] == {"layer_names": [l for l, _ in layers], "backend": "tensorflow", "keras_version": "2.6.0"}
    assert got["attrs"]["dense"] == {"weight_names": ["dense/kernel:0", "dense/bias:0"]} and got["attrs"]["empty"] == {"weight_names": []}
    names = got["attrs"]["many"]["weight_names0"] + got["attrs"]["many"]["weight_names1"]
    assert names == [n for n, _ in layers[3][1]]
    n_sets = 0
    for lname, ws in layers:
        for n, a in ws:
            d = got["datasets"][lname + "/" + n]
            a = np.asarray(a)
            assert d["shape"] == list(a.shape) and d["dtype"] == str(a.dtype) and d["sha1"] == hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest(), n
            n_sets += 1
    assert n_sets == len(got["datasets"]) == 1506


def test_vision_transformer_saves_and_reloads_a_keras_h5_weight_file(tmp_path):
    """`model.save_weights("x.h5")` (the reference's checkpoints, callbacks.py:31-38,99,103) writes Keras' layout with Keras-scoped
    variable names; another model loads it back bit for bit."""
    from chambers_amd.models.backbones.vision_transformer import VisionTransformer
    m = VisionTransformer(8, 64, 2, 1, 128, input_shape=(16, 16, 3), weights=None, classes=10)
    path = str(tmp_path / "vit.h5")
    m.save_weights(path)
    values, layout = H.load_keras_weights(path)
    names = [n for _l, ws in layout for n in ws]
    assert [l for l, _ in layout] == [l.name for l in m._layers]
    enc = [l.name for l in m._layers if l.name.startswith("encoder")][0]
    block0 = m.get_layer(enc).layers[0]
    assert "%s/%s/%s/w_query:0" % (enc, block0.name, block0.multi_head_attention.name) in names and "predictions/kernel:0" in names
    assert len(names) == len(m.weights) == len(set(names))
    m2 = VisionTransformer(8, 64, 2, 1, 128, input_shape=(16, 16, 3), weights=None, classes=10)
    assert any(not np.array_equal(a, b) for a, b in zip(m.get_weights(), m2.get_weights()))
    m2.load_weights(path)
    for a, b in zip(m.get_weights(), m2.get_weights()):
        np.testing.assert_array_equal(a, b)
