"""CPU tier: host-side mirror of the reference interface - serialisation round trips, weight shapes / order, the parameter
table, weight-layout conversion, error behaviour.  Everything the reference TEXT can supply (signatures, defaults, get_config
key sets, the policy table, magnitude maps, op order, zoo constants, the decay filter) is asserted against the extracted
fixture tests/golden/reference_api.json in tests/test_reference_api.py; the expectations here that name reference constants
read them from that fixture too (REF)."""
import json
import os

import numpy as np
import pytest

from oracle import augment_ref as A
from oracle import rng_ref

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_api.json")) as _f:
    REF = json.load(_f)


def test_randaugment_layers_carry_the_reference_kwargs():
    """RandAugment(2, 9) builds the 16 transforms of augmentation_schemes.py:181-198 with the kwargs the reference's magnitude
    functions give at magnitude 9 (fixture) - and the oracle restates the same table."""
    from chambers_amd.augmentations import augmentation_schemes as S
    ra = S.RandAugment(2, 9)
    names = [type(t).__name__ for t in ra.transforms]
    assert names == REF["randaugment_ops"] == A.RANDAUGMENT_OPS
    for t in ra.transforms:
        for k, v in REF["magnitude_kwargs"][type(t).__name__][9].items():
            assert getattr(t, k) == v, (type(t).__name__, k)
    for name in names:
        assert S._get_transform(name, 4) is not None


def test_random_layers_serialisation_round_trip():
    """RandomChance / RandomChoice serialise their inner layers through the Keras registry (image_augmentations.py:534-545,588-604)."""
    from chambers_amd import augmentations as aug
    base = {"name", "trainable", "dtype"}
    keys = lambda c: set(REF["classes"]["augmentations/image_augmentations.py"][c]["get_config_keys"])  # noqa: E731
    rc = aug.RandomChance(aug.Invert(), 0.3)
    assert rc.name == "random_chance_invert" or rc.name.startswith("random_chance_invert")
    cfg = rc.get_config()
    assert set(cfg) == base | keys("RandomChance") and cfg["transform"]["class_name"] == "Chambers>Invert"
    rc2 = aug.RandomChance.from_config(cfg)
    assert isinstance(rc2.transform, aug.Invert) and rc2.probability == 0.3
    ch = aug.RandomChoice([aug.Invert(), aug.Posterize(2)], n_transforms=2)
    assert set(ch.get_config()) == base | keys("RandomChoice")
    ch2 = aug.RandomChoice.from_config(ch.get_config())
    assert [type(t).__name__ for t in ch2.transforms] == ["Invert", "Posterize"] and ch2.n_transforms == 2
    assert ch.compute_output_shape((4, 8, 8, 3)) == [4, 8, 8, 3]


def test_layer_configs_and_weight_order():
    from chambers_amd.layers.attention import MultiHeadAttention
    from chambers_amd.layers.transformer import Encoder, EncoderLayer
    base = {"name", "trainable", "dtype"}
    mha = MultiHeadAttention(head_dim=64, num_heads=2)
    mref = REF["classes"]["layers/attention.py"]["MultiHeadAttention"]
    assert set(mha.get_config()) == base | set(mref["get_config_keys"])
    mha.build([(None, 5, 128)] * 3)
    assert [w.name.split("/")[-1] for w in mha.weights] == [n + ":0" for n in mref["add_weight_names"]]      # layers/attention.py:54-96
    assert [w.shape for w in mha.weights] == [(128, 2, 64), (2, 1, 64)] * 3 + [(2, 128, 64), (1, 128)]
    el = EncoderLayer(embed_dim=128, num_heads=2, ff_dim=256, pre_norm=True)
    assert set(el.get_config()) == base | set(REF["classes"]["layers/transformer.py"]["EncoderLayer"]["get_config_keys"])
    enc = Encoder(128, 2, 256, 3, pre_norm=True, norm_output=True)
    assert set(enc.get_config()) == base | set(REF["classes"]["layers/transformer.py"]["Encoder"]["get_config_keys"])
    enc.build((None, 5, 128))
    assert len(enc.layers) == 3 and len(enc.weights) == 3 * 16 + 2
    with pytest.raises(ValueError):
        mha.set_weights([np.zeros((1,))])


def test_model_builder_signature_names_and_errors():
    from chambers_amd.models.backbones import vision_transformer as V
    from chambers_amd.models import vit
    assert vit.ViTB16 is V.ViTB16 and V.preprocess_input.mode == REF["preprocess_input"]["kwargs"]["mode"] \
        and V.preprocess_input.name == REF["preprocess_input"]["kwargs"]["name"]
    m = V.VisionTransformer(16, 128, 2, 2, 256, input_shape=(64, 48, 3), weights=None, classes=10, model_name="tiny")
    assert m.name == "tiny"
    names = [l.name for l in m.layers]
    assert names[:3] == ["patch_embeddings", "add_cls_token", "pos_embedding"] and names[3].startswith("dropout") and names[4:] == ["encoder", "predictions"]
    assert m.get_layer("patch_embeddings").get_layer("embedding").kernel.shape == (16, 16, 3, 128)
    assert m.get_layer("add_cls_token").embedding.shape == (1, 128) and m.get_layer("pos_embedding").embedding.shape == (13, 128)
    assert m.get_layer("encoder").norm_layer is not None and len(m.get_layer("encoder").layers) == 2
    n_params = 16 * 16 * 3 * 128 + 128 + 128 + 13 * 128 + 2 * (4 * (128 * 128 + 128) + 2 * 128 * 256 + 256 + 128 + 4 * 128) + 2 * 128 + 128 * 10 + 10
    assert m.count_params() == n_params
    w = m.get_weights()
    m.set_weights(w)
    with pytest.raises(ValueError):
        V.VisionTransformer(16, 128, 2, 2, 256, input_shape=(8, 8, 3), weights=None)            # smaller than a patch
    with pytest.raises(ValueError):
        V.VisionTransformer(16, 128, 2, 2, 256, input_shape=(64, None, 3), weights=None)         # not fully specified
    with pytest.raises(ValueError, match="mutually exclusive"):
        V.ViTB16(weights="imagenet21k+_224", feature_dim=8)          # a release weight name of THIS model (vision_transformer.py:213-214)
    with pytest.raises(ValueError, match="require `input_shape`"):
        V.ViTB16(weights="imagenet21k+_384", input_shape=(224, 224, 3))                          # :120-128
    with pytest.raises(RuntimeError, match="no network"):
        V.ViTB16(weights="imagenet21k+_224")                         # resolves to a cache file that is not there
    # ViT-B/16 parameter count of the reference zoo config (86,567,656 - SURVEY §5)
    from chambers_amd.engine import ViTConfig, build_param_table
    zb = REF["zoo"]["ViTB16"]["constants"]
    specs, _tot, buckets = build_param_table(ViTConfig(zb["patch_size"], zb["patch_dim"], zb["n_encoder_layers"], zb["n_heads"], zb["ff_dim"]))
    real = sum(s.size for s in specs) - (1024 - 1000) * (768 + 1)
    assert real == 86567656
    assert len(buckets) == 2 + 2 * 12 and buckets[0][0] == 0 and all(a[1] == b[0] for a, b in zip(buckets, buckets[1:]))


def test_adamw_facade_regex_semantics():
    from chambers_amd.optimizers import AdamW
    with pytest.raises(ValueError):
        AdamW(0.1, decay_include=["a"], decay_exclude=["b"])
    for row in REF["adamw_is_decay_allowed"]:          # outcomes of the reference's own _is_decay_allowed (optimizers.py:169-181)
        o = AdamW(0.05, decay_include=row["filter"]["decay_include"], decay_exclude=row["filter"]["decay_exclude"])
        assert {n: o._is_decay_allowed(n) for n in row["allowed"]} == row["allowed"]
    o = AdamW(0.05, decay_exclude=["bias", "norm", "embeddings"])
    assert set(o.get_config()) >= set(REF["classes"]["optimizers.py"]["WeightDecayExtension"]["get_config_keys"]) | \
        set(REF["classes"]["optimizers.py"]["AdamW"]["init"]["args"]) - {"name"}
    from chambers_amd.engine import ViTConfig, build_param_table
    specs, _, _ = build_param_table(ViTConfig(16, 128, 1, 2, 256, image_size=(32, 32), classes=10), o.decay_fn())
    assert {s.name: s.decay for s in specs}["encoder/layer_0/qkv/kernel"] and not {s.name: s.decay for s in specs}["encoder/layer_0/qkv/bias"]


def test_adamw_regex_on_keras_variable_names():
    """decay_include / decay_exclude are matched on the reference's `var.name` strings (optimizers.py:169-181)."""
    from chambers_amd.engine import ViTConfig, build_param_table, keras_variable_names
    from chambers_amd.optimizers import AdamW
    cfg = ViTConfig(16, 128, 2, 2, 256, image_size=(32, 32), classes=10, feature_dim=32)
    names = keras_variable_names(cfg)
    specs, _, _ = build_param_table(cfg)
    assert set(names) == {s.name for s in specs}
    assert names["encoder/layer_0/norm2/gamma"] == ["encoder/encoder_layer/layer_normalization_1/gamma:0"]
    assert names["encoder/layer_1/dense1/kernel"] == ["encoder/encoder_layer_1/dense_2/kernel:0"]
    assert names["encoder/norm/beta"] == ["encoder/layer_normalization_4/beta:0"]
    assert names["encoder/layer_1/qkv/kernel"][0] == "encoder/encoder_layer_1/multi_head_attention_1/w_query:0"
    o = AdamW(0.05, decay_exclude=["bias", "/b_", "layer_normalization", "embeddings"])
    dec = {s.name: s.decay for s in build_param_table(cfg, o.decay_fn(cfg))[0]}
    assert dec["encoder/layer_0/qkv/kernel"] and dec["encoder/layer_1/dense2/kernel"] and dec["feature/kernel"]
    assert not dec["encoder/layer_0/qkv/bias"] and not dec["encoder/layer_0/proj/bias"] and not dec["encoder/norm/gamma"]
    assert not dec["pos_embedding/embeddings"] and not dec["add_cls_token/embeddings"] and not dec["predictions/bias"]
    # a filter that would split the fused QKV tensor is refused, not silently approximated
    with pytest.raises(ValueError):
        build_param_table(cfg, AdamW(0.05, decay_include=["w_query"]).decay_fn(cfg))


def test_linear_warmup_schedule():
    """schedules.py:5-48: ramp (step * lr0/warmup, then the inner schedule shifted by warmup) and multiplier modes."""
    from chambers_amd.schedules import LearningRateSchedule, LinearWarmup
    s = LinearWarmup(1e-3, 10)
    assert float(s(0)) == 0.0 and np.isclose(float(s(5)), 5e-4, rtol=1e-6) and float(s(10)) == np.float32(1e-3) == float(s(1000))
    m = LinearWarmup(2e-3, 4, ramp=False)
    assert np.isclose(float(m(1)), 5e-4, rtol=1e-6) and float(m(4)) == np.float32(2e-3) == float(m(9))

    class Halving(LearningRateSchedule):
        def __call__(self, step):
            return np.float32(1e-2) * np.float32(0.5) ** np.float32(step)
    r = LinearWarmup(Halving(), 2)
    assert np.isclose(float(r(1)), 5e-3) and np.isclose(float(r(2)), 1e-2) and np.isclose(float(r(4)), 2.5e-3)
    c = LinearWarmup(lambda: 4e-3, 2, ramp=False)
    assert np.isclose(float(c(1)), 2e-3) and np.isclose(float(c(3)), 4e-3)
    assert set(s.get_config()) == {"learning_rate", "warmup_steps", "ramp"}
    # drives the optimizer facade: hyper-parameters that are schedules are evaluated at the optimizer step count
    from chambers_amd.optimizers import AdamW
    assert np.isclose(AdamW(0.0, learning_rate=s)._value(s, 5), 5e-4, rtol=1e-6)


def test_rng_contract_matches_oracle_definition():
    from chambers_amd import rng
    for seed, step, site in ((0, 0, 0), (7, 3, 11), (2 ** 40 + 5, 1000, 36)):
        assert rng.site_key(seed, step, site) == rng_ref.site_key(seed, step, site)
    keys = {rng.site_key(1, s, k) for s in range(50) for k in range(40)}
    assert len(keys) == 2000
    m = rng_ref.keep_mask(1 << 20, 12345, 0.1)
    assert abs(m.mean() - 0.9) < 2e-3
    assert rng.site_attn(2) == 7 and rng.site_proj(2) == 8 and rng.site_mlp(2) == 9 and rng.SITE_EMBED == 0


def test_contrast_constant_host_side():
    from chambers_amd.augmentations import Contrast
    for n in (16, 224 * 224, 2 * 224 * 224, 384 * 384, 65279, 65280, 65535, 65536):
        assert Contrast.degenerate_constant(n) == A.contrast_constant(n)


def test_rotate_transform_host_side_matches_oracle():
    from chambers_amd.augmentations import Rotate
    import math
    for deg, neg, h, w in ((27.0, False, 224, 224), (27.0, True, 37, 53), (0.0, False, 5, 9), (90.0, True, 64, 48)):
        rad = deg * math.pi / 180.0
        np.testing.assert_array_equal(Rotate.transform_for(-rad if neg else rad, h, w), A.rotate_transform(deg, neg, h, w))


def test_distilled_parameter_table_and_weight_roundtrip():
    """DistilledVisionTransformer weights (vision_transformer.py:340-357,383-390): add_dist_token / predictions_dist exist, the two
    special tokens share one internal [2, D] tensor, and the Keras <-> internal conversion round-trips exactly."""
    from chambers_amd.engine import (ViTConfig, build_param_table, init_keras_weights, internal_to_keras, keras_to_internal,
                                     keras_variable_names)
    cfg = ViTConfig(16, 128, 2, 2, 256, image_size=(32, 32), classes=10, distilled=True)
    assert cfg.n_special == 2 and cfg.n_tokens == 4 + 2
    kw = init_keras_weights(cfg)
    assert kw["add_dist_token/embeddings"].shape == (1, 128) and kw["predictions_dist/kernel"].shape == (128, 10)
    iw = keras_to_internal(kw, cfg)
    assert iw["add_cls_token/embeddings"].shape == (2, 128)
    back = internal_to_keras(iw, cfg)
    assert set(back) == set(kw)
    for k in kw:
        np.testing.assert_array_equal(back[k], kw[k])
    specs, _, buckets = build_param_table(cfg)
    assert set(keras_variable_names(cfg)) == {s.name for s in specs} and len(buckets) == 2 + 2 * cfg.n_encoder_layers
    assert keras_variable_names(cfg)["add_cls_token/embeddings"] == ["add_cls_token/embeddings:0", "add_dist_token/embeddings:0"]
    with pytest.raises(ValueError):
        ViTConfig(16, 128, 2, 2, 256, distilled=True, feature_dim=64)


def test_items_sort_groups_counts_and_row_shift_flags():
    """chb_aug_items_sort (host only, no GPU): images of an elementwise batch sorted by what their chain needs - chains without a table op
    by kind (pixel-local | warps that keep rows | the rest, chains with a Sharpness first), then chains with one by kind - for the whole
    chain (row 0) and, per table level, by the kind of the levels under it; `pad` marks pure row shifts."""
    from chambers_amd import _lib
    from chambers_amd import kernels as K
    n = 2
    G = K.ITEMS_GROUPS

    def affine(f):
        r = np.zeros((), dtype=K.FUSED_OP_DTYPE)
        r["op"] = _lib.AUG_AFFINE
        r["f"] = np.asarray(f, dtype=np.float32)
        return r

    shift = affine([1, 0, 10, 0, 1, 0])            # TranslateX: a pure row shift
    rot = affine([0.9, 0.4, 0, -0.4, 0.9, 0])      # rows do not stay rows
    #            level 0                level 1
    chains = [(_lib.AUG_INVERT,        _lib.AUG_EQUALIZE),      # 0  local, table at level 1 under a local level
              (shift,                  _lib.AUG_AUTOCONTRAST),  # 1  rows, table at level 1 under a row warp
              (rot,                    _lib.AUG_EQUALIZE),      # 2  general, table at level 1 under a general warp
              (_lib.AUG_EQUALIZE,      shift),                  # 3  rows, table at level 0 (nothing under it)
              (_lib.AUG_SHARPNESS,     _lib.AUG_INVERT),        # 4  general, in front of 6
              (_lib.AUG_CUTOUT,        _lib.AUG_POSTERIZE),     # 5  local
              (_lib.AUG_IDENTITY,      rot),                    # 6  general
              (shift,                  _lib.AUG_SHARPNESS),     # 7  general, in front of 6
              (_lib.AUG_SHARPNESS,     _lib.AUG_EQUALIZE),      # 8  general with a table, in front of 2
              (_lib.AUG_BRIGHTNESS,    _lib.AUG_INVERT),        # 9  local
              (_lib.AUG_INVERT,        shift)]                  # 10 rows
    b = len(chains)
    recs = np.zeros((n, b), dtype=K.FUSED_OP_DTYPE)
    for i, (a, c) in enumerate(chains):
        for l, v in enumerate((a, c)):
            if isinstance(v, np.ndarray):
                recs[l, i] = v
            else:
                recs[l, i]["op"] = v
    order = np.full((1 + n, b), -1, dtype=np.int32)
    counts = np.zeros((1 + n) * G, dtype=np.int32)
    _lib.call("chb_aug_items_sort", recs.ctypes.data, b, 32, 48, n, order.ctypes.data, counts.ctypes.data)
    counts = counts.reshape(1 + n, G)
    assert counts.tolist() == [[2, 1, 3, 1, 2, 2], [1, 0, 0, 0, 0, 0], [1, 1, 2, 0, 0, 0]]
    assert order[0].tolist() == [5, 9, 10, 4, 7, 6, 0, 1, 3, 8, 2]
    assert order[1, :1].tolist() == [3]
    assert order[2, :4].tolist() == [0, 1, 8, 2]              # under the table of chain 8 sits a Sharpness: general, and first
    assert recs["pad"][0].tolist() == [0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0] and recs["pad"][1].tolist() == [0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1]
    bad = recs.copy()
    bad[0, 0]["op"] = 99
    with pytest.raises(ValueError):
        _lib.call("chb_aug_items_sort", bad.ctypes.data, b, 32, 48, n, order.ctypes.data, counts.ctypes.data)
