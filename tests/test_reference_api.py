"""CPU tier: the host logic of BOTH the product (`chambers_amd`) and the checker (`oracle/`) against
tests/golden/reference_api.json - data extracted from the reference TEXT by tools/extract_reference_api.py (ast over
/root/reference/chambers in the build container; no TensorFlow, nothing imported): constructor signatures and defaults,
get_config key sets, weight names, the AutoAugment policy, the magnitude -> kwargs maps evaluated by the reference's own
pure-Python functions, the RandAugment op order, the zoo constants, the pretrained-weight table logic and AdamW's
decay filter evaluated by the reference's own `_is_decay_allowed`.  No expectation below is typed by hand when the fixture
can supply it."""
import inspect
import json
import os

import pytest

from oracle import augment_ref as A

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_api.json")) as f:
    REF = json.load(f)

AUG = "augmentations/image_augmentations.py"
SCH = "augmentations/augmentation_schemes.py"


def _jsonable(v):
    if isinstance(v, tuple):
        return [_jsonable(x) for x in v]
    if isinstance(v, list):
        return [_jsonable(x) for x in v]
    if isinstance(v, dict):
        return {k: _jsonable(x) for k, x in v.items()}
    return v


def _sig(obj):
    """Argument names (without self) and literal defaults of a constructor / function, in the fixture's form."""
    fn = obj.__init__ if inspect.isclass(obj) else obj
    sig = inspect.signature(fn)
    args, defaults, varkw, varargs = [], {}, None, None
    for name, p in sig.parameters.items():
        if name in ("self", "cls"):
            continue
        if p.kind == p.VAR_KEYWORD:
            varkw = name
        elif p.kind == p.VAR_POSITIONAL:
            varargs = name
        else:
            args.append(name)
            if p.default is not p.empty:
                defaults[name] = _jsonable(p.default)
    return args, defaults, varargs, varkw


def _check_signature(obj, ref_init, where, allow_extra=()):
    """Same positional names in the same order; every literal default of the reference equal; **kwargs accepted where the
    reference accepts them.  `allow_extra`: trailing arguments this build adds (none of them positional before a reference one)."""
    args, defaults, _va, varkw = _sig(obj)
    ref_args = ref_init["args"] + ref_init.get("kwonly", [])
    assert args[:len(ref_args)] == ref_args, "%s: arguments %s, reference %s" % (where, args, ref_args)
    assert set(args[len(ref_args):]) <= set(allow_extra), "%s: extra arguments %s" % (where, args[len(ref_args):])
    for name, val in ref_init["defaults"].items():
        if isinstance(val, dict) and "expr" in val:
            continue                       # a non-literal default (an initializer object, ...)
        assert name in defaults, "%s: %s has no default (reference: %r)" % (where, name, val)
        assert defaults[name] == val and type(defaults[name]) is type(val), "%s: default %s = %r, reference %r" % (where, name, defaults[name], val)
    for name in ref_args:
        if name not in ref_init["defaults"]:
            assert name not in defaults, "%s: %s is required in the reference" % (where, name)
    if ref_init.get("varkw"):
        assert varkw is not None, "%s: the reference accepts **%s" % (where, ref_init["varkw"])


# ---------------------------------------------------------------------------------------------------------------------
def test_policy_constants_and_op_order_product_and_oracle():
    from chambers_amd.augmentations import augmentation_schemes as S
    policy = [[tuple(t) for t in pair] for pair in REF["auto_augment_policy_v0"]]
    assert len(policy) == 25
    assert [list(map(tuple, p)) for p in S._AUTO_AUGMENT_POLICY_V0] == policy
    assert [list(map(tuple, p)) for p in A.AUTO_AUGMENT_POLICY_V0] == policy
    c = REF["augmentation_constants"]
    assert (S._INTERPOLATION_MODE, S._FILL_MODE, S._FILL_VALUE, S._MAX_MAGNITUDE) == \
           (c["_INTERPOLATION_MODE"], c["_FILL_MODE"], c["_FILL_VALUE"], c["_MAX_MAGNITUDE"])
    assert (A.FILL_VALUE, A.MAX_MAGNITUDE) == (c["_FILL_VALUE"], c["_MAX_MAGNITUDE"])
    assert A.RANDAUGMENT_OPS == REF["randaugment_ops"]
    ra = S.RandAugment(2, 9)
    assert [type(t).__name__ for t in ra.transforms] == REF["randaugment_ops"]
    assert REF["autoaugment_choice"] == {"n_transforms": 1, "elementwise": {"expr": "elementwise"}}
    aa = S.AutoAugment()
    assert len(aa.transforms) == len(policy)


def test_magnitude_maps_equal_the_reference_functions_output():
    """[transform][magnitude 0..10] -> kwargs, computed by the reference's own _magnitude_to_*_kwargs (pure Python)."""
    from chambers_amd.augmentations import augmentation_schemes as S
    assert set(REF["magnitude_kwargs"]) == set(REF["randaugment_ops"]) == set(REF["magnitude_fn_map"])
    for name, rows in REF["magnitude_kwargs"].items():
        for m, want in enumerate(rows):
            layer = S._get_transform(name, m)
            assert type(layer).__name__ == name
            for k, v in want.items():
                got = getattr(layer, k)
                assert got == v and type(got) is type(v), "%s(m=%d).%s = %r, reference %r" % (name, m, k, got, v)
            ok = A.magnitude_to_kwargs(name, m)
            for k, v in ok.items():                 # the oracle keeps only the arguments that change the arithmetic
                assert want[k] == v and type(want[k]) is type(v), "oracle %s(m=%d).%s = %r, reference %r" % (name, m, k, v, want[k])
            arithmetic = set(want) - {"interpolation", "fill_mode"}
            assert set(ok) == arithmetic, (name, set(ok), arithmetic)
    # a fractional magnitude goes through the same expressions (spot value from the fixture's integer rows: linear maps)
    assert S._get_transform("Brightness", 9).factor == REF["magnitude_kwargs"]["Brightness"][9]["factor"]


def test_augmentation_layer_signatures_config_keys_and_registration():
    from chambers_amd import augmentations as aug
    from chambers_amd import _keras_like as KL
    base = {"name", "trainable", "dtype"}
    built = {"AutoContrast": (), "Equalize": (), "Invert": (), "Rotate": (3.0,), "Posterize": (3,), "Solarize": (), "SolarizeAdd": (),
             "Color": (0.5,), "Contrast": (0.5,), "Brightness": (0.5,), "Sharpness": (0.5,), "ShearX": (0.1,), "ShearY": (0.1,),
             "TranslateX": (3,), "TranslateY": (3,), "CutOut": (8,), "ImageNetNormalization": ()}
    for cname, info in REF["classes"][AUG].items():
        if cname == "ResizingMinMax":
            cls = aug.ResizingMinMax
        else:
            cls = getattr(aug, cname)
        _check_signature(cls, info["init"], cname)
        assert info["registered_package"] == "Chambers"
        assert KL._REGISTRY["Chambers>" + cname] is cls
        if cname in built:
            layer = cls(*built[cname])
            keys = set(layer.get_config())
            want = set(info.get("get_config_keys") or [])
            assert keys == base | want, "%s.get_config keys %s, reference %s + base" % (cname, keys, want)
            assert cls.from_config(layer.get_config()).get_config() == layer.get_config()
    rc = aug.RandomChance(aug.Invert(), 0.3)
    assert set(rc.get_config()) == base | set(REF["classes"][AUG]["RandomChance"]["get_config_keys"])
    ch = aug.RandomChoice([aug.Invert()], n_transforms=1)
    assert set(ch.get_config()) == base | set(REF["classes"][AUG]["RandomChoice"]["get_config_keys"])
    for cname in ("AutoAugment", "RandAugment"):
        info = REF["classes"][SCH][cname]
        cls = getattr(aug, cname)
        _check_signature(cls, info["init"], cname)
        layer = cls() if cname == "AutoAugment" else cls(2, 9)
        assert set(layer.get_config()) == base | set(info["get_config_keys"])
        assert KL._REGISTRY["Chambers>" + cname] is cls
    p = REF["preprocess_input"]
    from chambers_amd.models.backbones import vision_transformer as V
    assert type(V.preprocess_input).__name__ == p["class"] and V.preprocess_input.mode == p["kwargs"]["mode"] \
        and V.preprocess_input.name == p["kwargs"]["name"]


def test_layer_signatures_config_keys_and_weight_names():
    from chambers_amd.layers import attention, embedding, normalization, transformer
    base = {"name", "trainable", "dtype"}
    cases = [("layers/attention.py", "MultiHeadAttention", attention.MultiHeadAttention, dict(head_dim=64, num_heads=2), [(None, 5, 128)] * 3),
             ("layers/transformer.py", "EncoderLayer", transformer.EncoderLayer, dict(embed_dim=128, num_heads=2, ff_dim=256), (None, 5, 128)),
             ("layers/transformer.py", "Encoder", transformer.Encoder, dict(embed_dim=128, num_heads=2, ff_dim=256, num_layers=2), (None, 5, 128)),
             ("layers/embedding.py", "LearnedEmbedding1D", embedding.LearnedEmbedding1D, {}, (None, 5, 128)),
             ("layers/embedding.py", "ConcatEmbedding", embedding.ConcatEmbedding, dict(n_embeddings=1, embedding_dim=128, axis=1), (None, 5, 128)),
             ("layers/normalization.py", "L2Normalization", normalization.L2Normalization, dict(axis=1), None)]
    for rel, cname, cls, kw, shape in cases:
        info = REF["classes"][rel][cname]
        _check_signature(cls, info["init"], cname, allow_extra=("name",))
        layer = cls(**kw)
        assert set(layer.get_config()) == base | set(info["get_config_keys"]), cname
        if shape is not None and info.get("add_weight_names"):
            layer.build(shape)
            own = [w.name.split("/")[-1].split(":")[0] for w in layer._weights]
            assert own == info["add_weight_names"], "%s weights %s, reference %s" % (cname, own, info["add_weight_names"])
    # the ScaledAttention subclass adds `key_dim` in front of keras Attention's arguments
    assert _sig(attention.ScaledAttention)[0][0] == REF["classes"]["layers/attention.py"]["ScaledAttention"]["init"]["args"][0] == "key_dim"
    # what EncoderLayer / Encoder construct inside (layers/transformer.py:8-77, 256-314): sub-layer kinds in source order
    el = transformer.EncoderLayer(embed_dim=128, num_heads=2, ff_dim=256)
    kinds = [k for k, _ in REF["encoder_sublayers"]["EncoderLayer"]]
    assert kinds == ["MultiHeadAttention", "Dropout", "LayerNormalization", "Dense", "Dense", "Dropout", "LayerNormalization"]
    got = [type(x).__name__ for x in (el.multi_head_attention, el.dropout1, el.norm1, el.dense1, el.dense2, el.dropout2, el.norm2)]
    assert got == kinds
    assert REF["encoder_sublayers"]["EncoderLayer"][0][1]["head_dim"] == {"expr": "embed_dim // num_heads"} and el.multi_head_attention.head_dim == 64
    assert REF["encoder_sublayers"]["EncoderLayer"][0][1]["causal"] is False and el.multi_head_attention.causal is False


def test_model_builders_zoo_constants_and_weight_table():
    from chambers_amd.models.backbones import vision_transformer as V
    rel = "models/backbones/vision_transformer.py"
    for fname in ("VisionTransformer", "DistilledVisionTransformer"):
        _check_signature(getattr(V, fname), REF["functions"][rel][fname], fname)
    for zname, z in REF["zoo"].items():
        fn = getattr(V, zname)
        args, defaults, _, _ = _sig(fn)
        assert defaults == {k: v for k, v in z["defaults"].items()}, zname
        assert args == list(REF["functions"][rel][zname]["args"]), zname
        consts = z["constants"]
        captured = {}
        builder = z["builder"]
        orig = getattr(V, builder)
        try:
            setattr(V, builder, lambda **kw: captured.update(kw))
            fn(weights=None)
        finally:
            setattr(V, builder, orig)
        for k, v in z["call_literals"].items():
            assert captured[k] == v, "%s passes %s=%r, reference %r" % (zname, k, captured[k], v)
        for k in ("patch_size", "patch_dim", "n_encoder_layers", "n_heads", "ff_dim"):
            assert captured[k] == consts[k]
    # pretrained-weight names and the reference's own _get_model_info / _are_weights_pretrained over them
    for model, rows in REF["model_info"].items():
        for wname, info in rows.items():
            w = None if wname == "None" else wname
            assert V._are_weights_pretrained(w, model) == info["pretrained"], (model, wname)
            assert tuple(V._get_model_info(w, model)) == (info["default_size"], info["has_feature"]), (model, wname)
    # names the builders give their layers, in source order (literal ones)
    names = [n for _f, n in REF["builder_layer_names"]["VisionTransformer"] if isinstance(n, str)]
    m = V.VisionTransformer(16, 128, 2, 2, 256, input_shape=(64, 48, 3), weights=None, classes=10, feature_dim=32)
    have = [l.name for l in m.layers] + [m.get_layer("patch_embeddings").get_layer("embedding").name]
    assert sorted(names) == sorted(["embedding", "patch_embeddings", "add_cls_token", "pos_embedding", "feature", "predictions"])
    for n in names:
        assert n in have, n
    assert "encoder" in have          # the reference leaves the Encoder unnamed: Keras derives "encoder" from the class name
    enc_kw = dict(next(kw for kind, kw in REF["builder_calls"]["VisionTransformer"] if kind == "Encoder"))
    enc = m.get_layer("encoder")
    assert enc.pre_norm is enc_kw["pre_norm"] is True and (enc.norm_layer is not None) is enc_kw["norm_output"]
    conv_kw = dict(next(kw for kind, kw in REF["builder_calls"]["VisionTransformer"] if kind == "Conv2D"))
    assert conv_kw["padding"] == "valid" and conv_kw["strides"] == {"expr": "patch_size"} and conv_kw["kernel_size"] == {"expr": "patch_size"}
    feat_kw = next(kw for kind, kw in REF["builder_calls"]["VisionTransformer"] if kind == "Dense" and kw.get("name") == "feature")
    assert feat_kw["activation"] == "tanh"
    d = V.DistilledVisionTransformer(16, 128, 2, 2, 256, input_shape=(32, 32, 3), weights=None, classes=10, pooling="cls")
    dnames = [n for _f, n in REF["builder_layer_names"]["DistilledVisionTransformer"] if isinstance(n, str)]
    for n in ("add_dist_token", "predictions_dist"):
        assert n in dnames and n in [l.name for l in d.layers]
    # the oracle takes its geometry from the weights' shapes; what it is TOLD is what the builder passes on (oracle cfg keys)
    from chambers_amd.engine import ViTConfig
    z = REF["zoo"]["ViTB16"]["constants"]
    oc = ViTConfig(z["patch_size"], z["patch_dim"], z["n_encoder_layers"], z["n_heads"], z["ff_dim"]).as_oracle_cfg()
    assert (oc["patch_size"], oc["n_encoder_layers"], oc["n_heads"]) == (z["patch_size"], z["n_encoder_layers"], z["n_heads"])
    assert oc["dropout_rate"] == REF["zoo"]["ViTB16"]["call_literals"]["dropout_rate"]
    assert oc["norm_epsilon"] == REF["classes"]["layers/transformer.py"]["Encoder"]["init"]["defaults"]["norm_epsilon"]


def test_adamw_defaults_and_decay_filter_equal_the_reference_function():
    from chambers_amd.optimizers import AdamW
    info = REF["classes"]["optimizers.py"]["AdamW"]["init"]
    _check_signature(AdamW, info, "AdamW")
    assert REF["adamw_init_raises"] == ["ValueError"]
    with pytest.raises(ValueError):
        AdamW(0.1, decay_include=["a"], decay_exclude=["b"])
    for row in REF["adamw_is_decay_allowed"]:
        flt = row["filter"]
        opt = AdamW(0.05, decay_include=flt["decay_include"], decay_exclude=flt["decay_exclude"])
        for name, want in row["allowed"].items():
            assert opt._is_decay_allowed(name) == want, "filter %s on %s: %s, reference %s" % (flt, name, not want, want)
    keys = set(AdamW(0.05).get_config())
    assert keys >= set(REF["classes"]["optimizers.py"]["WeightDecayExtension"]["get_config_keys"]) | {"weight_decay"}


def test_decay_filter_names_are_the_engines_keras_names():
    """The variable names the fixture evaluates the reference filter on are names this build derives for its own tensors."""
    from chambers_amd.engine import ViTConfig, keras_variable_names
    cfg = ViTConfig(16, 128, 2, 2, 256, image_size=(32, 32), classes=10, feature_dim=32)
    mine = {n for names in keras_variable_names(cfg).values() for n in names}
    cfg_d = ViTConfig(16, 128, 2, 2, 256, image_size=(32, 32), classes=10, distilled=True)
    mine |= {n for names in keras_variable_names(cfg_d).values() for n in names}
    fixture_names = set(REF["adamw_is_decay_allowed"][0]["allowed"])
    assert fixture_names <= mine, sorted(fixture_names - mine)


def test_schedule_activation_miner_loss_signatures():
    from chambers_amd import activations, miners, schedules
    from chambers_amd.losses import metric_learning
    lw = REF["classes"]["schedules.py"]["LinearWarmup"]
    _check_signature(schedules.LinearWarmup, lw["init"], "LinearWarmup")
    assert set(schedules.LinearWarmup(1e-3, 10).get_config()) == set(lw["get_config_keys"])
    g = REF["gelu"]
    _check_signature(activations.gelu, g["signature"], "gelu")
    # the constants of the exact and the tanh form (activations.py:30-56)
    assert g["float_literals"] == [0.044715, 0.5, 0.7978845608028654, 1.0, 1.4142135623730951]
    _check_signature(miners.MultiSimilarityMiner, REF["classes"]["miners.py"]["MultiSimilarityMiner"]["init"], "MultiSimilarityMiner")
    for cls in ("MultiSimilarityLoss", "MultiSimilarityLossMatrix", "ContrastiveLoss", "NTXentLoss"):
        _check_signature(getattr(metric_learning, cls), REF["classes"]["losses/metric_learning.py"][cls]["init"], cls)
    # the default miner of the two multi-similarity losses is the reference's expression
    assert REF["classes"]["losses/metric_learning.py"]["MultiSimilarityLossMatrix"]["init"]["defaults"]["miner"] == {"expr": "_MSMiner(margin=0.1)"}
    assert metric_learning.MultiSimilarityLossMatrix().miner.margin == 0.1 and metric_learning.MultiSimilarityLoss().miner.margin == 0.1
    assert metric_learning.ContrastiveLoss().miner is None


def test_oracle_layer_constants_follow_the_fixture():
    """oracle/vit_ref.py restates the block with the fixture's literals: pre-norm + final norm in the ViT builder, epsilon passed
    through `norm_epsilon` (default 1e-6), head_dim = embed_dim // num_heads, tanh feature head."""
    from oracle import vit_ref
    el = REF["classes"]["layers/transformer.py"]["EncoderLayer"]["init"]["defaults"]
    assert el["norm_epsilon"] == 1e-06 and el["pre_norm"] is False
    src = inspect.getsource(vit_ref)
    assert "1e-6" in src or "1e-06" in src
    enc_kw = dict(next(kw for kind, kw in REF["builder_calls"]["VisionTransformer"] if kind == "Encoder"))
    assert enc_kw["pre_norm"] is True and enc_kw["norm_output"] is True
    assert hasattr(vit_ref, "encoder_layer") and hasattr(vit_ref, "encoder_layer_post_norm")
