"""GPU parity through the chambers API surface (layer classes and model builders), i.e. the drop-in boundary:
chambers.layers.attention.MultiHeadAttention, chambers.layers.transformer.EncoderLayer/Encoder,
chambers.models.backbones.vision_transformer.VisionTransformer — against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import augment_ref as A
from oracle import vit_ref

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _randomize(layer, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    layer.set_weights([(w + g.normal(0, 0.05, size=w.shape)).astype(np.float32) for w in layer.get_weights()])


def test_multi_head_attention_layer_matches_oracle():
    from chambers_amd.layers.attention import MultiHeadAttention
    b, t, d, heads = 2, 197, 192, 3
    mha = MultiHeadAttention(head_dim=64, num_heads=heads, dropout_rate=0.1)
    x = torch.randn(b, t, d, generator=torch.Generator().manual_seed(0))
    xd = x.cuda()
    out = mha([xd, xd, xd], training=False)
    _randomize(mha, 1)
    out = mha([xd, xd, xd], training=False)
    names = ["w_query", "b_query", "w_value", "b_value", "w_key", "b_key", "w_projection", "b_projection"]
    p = {"m/" + n: torch.tensor(w) for n, w in zip(names, mha.get_weights())}
    ref = vit_ref.multi_head_attention(x, p, "m/", heads, 0.0, None, True)
    assert tuple(out.shape) == (b, t, d) and out.dtype == torch.float32
    assert rel_l2(out, ref) < 4e-3
    ref32 = vit_ref.multi_head_attention(x, p, "m/", heads, 0.0, None, False)
    assert rel_l2(out, ref32) < 1e-2
    # cross-attention inputs (distinct tensors) take the general kernel; identical values must give the self-attention result
    out_x = mha([xd, xd.clone(), xd.clone()], training=False)
    assert rel_l2(out_x, out) < 4e-3


def test_encoder_layer_and_encoder_match_oracle():
    from chambers_amd.layers.transformer import Encoder, EncoderLayer
    b, t, d, heads, ff = 2, 50, 128, 2, 256
    x = torch.randn(b, t, d, generator=torch.Generator().manual_seed(2))
    el = EncoderLayer(embed_dim=d, num_heads=heads, ff_dim=ff, pre_norm=True)
    el(x.cuda(), training=False)
    _randomize(el, 3)
    out = el(x.cuda(), training=False)
    w = el.get_weights()
    names = ["multi_head_attention/" + n for n in ("w_query", "b_query", "w_value", "b_value", "w_key", "b_key", "w_projection", "b_projection")] + \
            ["norm1/gamma", "norm1/beta", "dense1/kernel", "dense1/bias", "dense2/kernel", "dense2/bias", "norm2/gamma", "norm2/beta"]
    p = {"encoder/layer_0/" + n: torch.tensor(a) for n, a in zip(names, w)}
    cfg = {"dropout_rate": 0.0, "n_heads": heads, "norm_epsilon": 1e-6}
    ref = vit_ref.encoder_layer(x, p, "encoder/layer_0/", cfg, {}, 0, True)
    assert rel_l2(out, ref) < 4e-3
    # the reference's default block (pre_norm=False, layers/transformer.py:59-61): same weights, post-norm composition
    el_post = EncoderLayer(embed_dim=d, num_heads=heads, ff_dim=ff)
    assert el_post.pre_norm is False
    el_post(x.cuda(), training=False)
    el_post.set_weights(w)
    out_post = el_post(x.cuda(), training=False)
    ref_post = vit_ref.encoder_layer_post_norm(x, p, "encoder/layer_0/", cfg, {}, 0, True)
    assert rel_l2(out_post, ref_post) < 4e-3 and rel_l2(out_post, ref) > 0.1
    out_m = el_post(x.cuda(), mask=torch.ones(b, t, dtype=torch.bool, device="cuda"), training=False)      # an all-ones mask changes nothing
    assert rel_l2(out_m, out_post) < 6e-3
    enc = Encoder(d, heads, ff, 2, pre_norm=True, norm_output=True)
    y = enc(x.cuda(), training=False)
    assert tuple(y.shape) == (b, t, d) and len(enc.get_weights()) == 2 * 16 + 2


def test_vision_transformer_model_matches_oracle_and_trains():
    from chambers_amd.models.backbones.vision_transformer import VisionTransformer, preprocess_input
    m = VisionTransformer(16, 128, 2, 2, 256, input_shape=(64, 48, 3), weights=None, classes=10, model_name="tiny")
    _randomize(m, 5)
    g = np.random.Generator(np.random.PCG64(0))
    images = g.integers(0, 256, size=(4, 64, 48, 3), dtype=np.uint8)
    xd = torch.as_tensor(images, device="cuda")
    xf = preprocess_input(xd)                                   # ImageNetNormalization("tf"), fp32 NHWC
    np.testing.assert_array_equal(xf.cpu().numpy(), A.imagenet_normalize(images, "tf"))
    logits_f = m(xf)                                            # the reference model's own input convention
    logits_u = m(xd)                                            # uint8 input: normalisation fused into the patch gather
    assert torch.equal(logits_f, logits_u)
    kw = {k: torch.tensor(v) for k, v in m.keras_weights().items()}
    ref = vit_ref.vit_forward(kw, torch.from_numpy(A.imagenet_normalize(images, "tf")), m.cfg.as_oracle_cfg(), bf16=True)
    assert rel_l2(logits_f, ref) < 4e-3
    # get_weights/set_weights round trip changes the prediction consistently
    w = m.get_weights()
    m.set_weights([a * 0.5 for a in w])
    assert not torch.allclose(m(xd), logits_u)
    m.set_weights(w)
    assert torch.equal(m(xd), logits_u)
    # training through the Model facade + weights synced back into the Keras-named variables
    labels = torch.as_tensor(g.integers(0, 10, size=(4,)), device="cuda")
    first = float(m.train_step(xd, labels, learning_rate=1e-3).mean())
    for _ in range(15):
        last = float(m.train_step(xd, labels, learning_rate=1e-3).mean())
    assert last < first
    m.sync_from_engine(4)
    assert not np.allclose(m.get_weights()[0], w[0])


def test_save_and_load_weights(tmp_path):
    from chambers_amd.models.backbones.vision_transformer import VisionTransformer
    m = VisionTransformer(16, 128, 1, 2, 256, input_shape=(32, 32, 3), weights=None, classes=5)
    path = str(tmp_path / "w.npz")
    m.save_weights(path)
    m2 = VisionTransformer(16, 128, 1, 2, 256, input_shape=(32, 32, 3), weights=path, classes=5)
    for a, b in zip(m.get_weights(), m2.get_weights()):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("pooling,feature_dim,include_top", [("avg", 48, False), ("max", None, True), (None, None, False)])
def test_model_builder_pooling_and_feature_head(pooling, feature_dim, include_top):
    """Builder kwargs `pooling`, `feature_dim`, `include_top` (vision_transformer.py:194-210,272-283): output shape and values."""
    from chambers_amd.models.backbones.vision_transformer import VisionTransformer
    m = VisionTransformer(16, 128, 2, 2, 256, input_shape=(64, 48, 3), weights=None, classes=10, pooling=pooling,
                          feature_dim=feature_dim, include_top=include_top)
    _randomize(m, 7)
    names = [l.name for l in m.layers]
    assert ("feature" in names) == bool(feature_dim) and ("predictions" in names) == include_top
    g = np.random.Generator(np.random.PCG64(1))
    images = g.integers(0, 256, size=(3, 64, 48, 3), dtype=np.uint8)
    out = m(torch.as_tensor(images, device="cuda")).float().cpu()
    kw = {k: torch.tensor(v) for k, v in m.keras_weights().items()}
    ref = vit_ref.vit_forward(kw, torch.from_numpy(A.imagenet_normalize(images, "tf")), m.cfg.as_oracle_cfg(), bf16=True,
                              return_tokens=pooling is None)
    expect = (3, 13, 128) if pooling is None else (3, 10 if include_top else (feature_dim or 128))
    assert tuple(out.shape) == expect == tuple(ref.shape)
    assert rel_l2(out, ref) < 4e-3


def test_classifier_activation_softmax():
    from chambers_amd.models.backbones.vision_transformer import VisionTransformer
    m = VisionTransformer(16, 128, 1, 2, 256, input_shape=(32, 32, 3), weights=None, classes=5, classifier_activation="softmax")
    m2 = VisionTransformer(16, 128, 1, 2, 256, input_shape=(32, 32, 3), weights=None, classes=5)
    m2.set_weights(m.get_weights())
    x = torch.randint(0, 256, (2, 32, 32, 3), dtype=torch.uint8, device="cuda")
    pr, lg = m(x), m2(x)
    assert torch.allclose(pr, torch.softmax(lg, dim=-1), atol=1e-6)
    with pytest.raises(ValueError):
        VisionTransformer(16, 128, 1, 2, 256, input_shape=(32, 32, 3), weights=None, classes=5, classifier_activation="relu6")


def test_compile_with_adamw_regex_and_warmup_schedule():
    """Model.compile(optimizer=AdamW(decay_exclude=..., learning_rate=LinearWarmup(...))): excluded variables see no decay,
    the step size follows the schedule (step 0 of a ramp has lr 0: only the decay moves the weights)."""
    from chambers_amd.models.backbones.vision_transformer import VisionTransformer
    from chambers_amd.optimizers import AdamW
    from chambers_amd.schedules import LinearWarmup
    m = VisionTransformer(16, 128, 1, 2, 256, input_shape=(32, 32, 3), weights=None, classes=5)
    _randomize(m, 3)
    before = {k: v.copy() for k, v in m.keras_weights().items()}
    m.compile(optimizer=AdamW(0.1, decay_exclude=["bias", "/b_", "layer_normalization", "embeddings"], learning_rate=LinearWarmup(1e-3, 4)))
    x = torch.randint(0, 256, (4, 32, 32, 3), dtype=torch.uint8, device="cuda")
    y = torch.randint(0, 5, (4,), device="cuda")
    m.train_step(x, y)
    m.sync_from_engine(4)
    after = m.keras_weights()
    for k in before:
        decayed = (k.endswith("kernel") or "/w_" in k) and "embeddings" not in k   # "patch_embeddings/..." matches the regex too
        if decayed:
            np.testing.assert_allclose(after[k], before[k] * np.float32(0.9), rtol=2e-6, atol=1e-9, err_msg=k)   # lr(0) = 0
        else:
            np.testing.assert_array_equal(after[k], before[k], err_msg=k)
    with pytest.raises(ValueError):
        m.train_step(x, y, learning_rate=1.0)
    l1 = m.train_step(x, y)        # lr(1) = 2.5e-4 now moves everything
    m.sync_from_engine(4)
    assert not np.array_equal(m.keras_weights()["predictions/bias"], before["predictions/bias"]) and torch.isfinite(l1).all()


def test_deit_builder_outputs_and_timm_import():
    """DeiTS16-style builder (vision_transformer.py:583-617) at a reduced width: layer names, weight count, the [cls, dist] output
    pair against the oracle, and import of a timm-layout distilled state dict (dist_token, head_dist)."""
    from chambers_amd.models.backbones.vision_transformer import DeiTS16, DistilledVisionTransformer
    with pytest.raises(RuntimeError):
        DeiTS16()                                              # default weights tag needs the network
    with pytest.raises(ValueError):
        DistilledVisionTransformer(16, 128, 1, 2, 256, input_shape=(32, 32, 3), weights=None)   # reference default pooling=None
    m = DistilledVisionTransformer(16, 128, 2, 2, 256, input_shape=(64, 48, 3), weights=None, pooling="cls", classes=10)
    names = [l.name for l in m.layers]
    assert names[:4] == ["patch_embeddings", "add_dist_token", "add_cls_token", "pos_embedding"] and names[-2:] == ["predictions", "predictions_dist"]
    _randomize(m, 11)
    g = np.random.Generator(np.random.PCG64(2))
    images = g.integers(0, 256, size=(3, 64, 48, 3), dtype=np.uint8)
    out = m(torch.as_tensor(images, device="cuda"))
    kw = {k: torch.tensor(v) for k, v in m.keras_weights().items()}
    ref = vit_ref.vit_forward(kw, torch.from_numpy(A.imagenet_normalize(images, "tf")), m.cfg.as_oracle_cfg(), bf16=True)
    assert isinstance(out, list) and len(out) == 2
    for o, r in zip(out, ref):
        assert tuple(o.shape) == (3, 10) and rel_l2(o, r) < 4e-3
    avg = DistilledVisionTransformer(16, 128, 2, 2, 256, input_shape=(64, 48, 3), weights=None, pooling="cls", classes=10,
                                     return_dist_token=False)
    avg.set_weights(m.get_weights())
    assert torch.allclose(avg(torch.as_tensor(images, device="cuda")), (out[0] + out[1]) / 2, atol=1e-6)
    # timm-layout distilled checkpoint -> same predictions as assigning the equivalent Keras arrays
    d, heads, ff, L, p_ = 128, 2, 256, 2, 16
    r = lambda *s_: (g.normal(0, 0.05, size=s_)).astype(np.float32)   # noqa: E731
    sd = {"patch_embed.proj.weight": r(d, 3, p_, p_), "patch_embed.proj.bias": r(d), "cls_token": r(1, 1, d), "dist_token": r(1, 1, d),
          "pos_embed": r(1, 14, d), "norm.weight": 1 + r(d), "norm.bias": r(d), "head.weight": r(10, d), "head.bias": r(10),
          "head_dist.weight": r(10, d), "head_dist.bias": r(10)}
    for i in range(L):
        t = "blocks.%d." % i
        sd.update({t + "norm1.weight": 1 + r(d), t + "norm1.bias": r(d), t + "attn.qkv.weight": r(3 * d, d), t + "attn.qkv.bias": r(3 * d),
                   t + "attn.proj.weight": r(d, d), t + "attn.proj.bias": r(d), t + "norm2.weight": 1 + r(d), t + "norm2.bias": r(d),
                   t + "mlp.fc1.weight": r(ff, d), t + "mlp.fc1.bias": r(ff), t + "mlp.fc2.weight": r(d, ff), t + "mlp.fc2.bias": r(d)})
    m.load_timm_state_dict(sd)
    kw2 = m.keras_weights()
    np.testing.assert_array_equal(kw2["add_dist_token/embeddings"], sd["dist_token"].reshape(1, d))
    np.testing.assert_array_equal(kw2["predictions_dist/kernel"], sd["head_dist.weight"].T)
    out2 = m(torch.as_tensor(images, device="cuda"))
    ref2 = vit_ref.vit_forward({k: torch.tensor(v) for k, v in kw2.items()}, torch.from_numpy(A.imagenet_normalize(images, "tf")),
                               m.cfg.as_oracle_cfg(), bf16=True)
    assert rel_l2(out2[1], ref2[1]) < 4e-3


def test_train_step_then_predict_uses_trained_weights_without_manual_sync():
    """Keras fit followed by predict / get_weights / save_weights: the trained weights are what every reader sees, the training
    engine keeps its Adam state across compile(), and an explicit set_weights supersedes the engine."""
    from chambers_amd.models.backbones.vision_transformer import VisionTransformer
    from chambers_amd.optimizers import AdamW
    m = VisionTransformer(patch_size=16, patch_dim=64, n_encoder_layers=2, n_heads=1, ff_dim=128, dropout_rate=0.0, input_shape=(32, 32, 3),
                          weights=None, classes=5)
    g = np.random.Generator(np.random.PCG64(3))
    x = torch.as_tensor(g.integers(0, 256, size=(4, 32, 32, 3), dtype=np.uint8), device="cuda")
    y = torch.as_tensor(g.integers(0, 5, size=(4,)), device="cuda")
    before = m(x).clone()
    w_before = [w.copy() for w in m.get_weights()]
    for _ in range(5):
        m.train_step(x, y, learning_rate=1e-2)
    after = m(x).clone()                                       # no sync_from_engine
    assert float((after - before).abs().max()) > 1e-3
    w_after = m.get_weights()
    assert any(float(np.abs(a - b).max()) > 1e-4 for a, b in zip(w_after, w_before))
    eng = m.engine(4, training=True)
    assert eng.opt_step == 5
    # the inference engine and the training engine agree on the weights
    exported = eng.export_keras_weights()
    kw = m.keras_weights()
    for k in kw:
        np.testing.assert_array_equal(kw[k], exported[k])
    # compile keeps weights, Adam moments and the step count
    mo = eng.Mo.clone()
    m.compile(optimizer=AdamW(weight_decay=0.01, learning_rate=1e-3, decay_exclude=["bias", "gamma", "beta", "embeddings"]))
    assert m.engine(4, training=True) is eng and torch.equal(eng.Mo, mo)
    np.testing.assert_array_equal(m.keras_weights()["predictions/kernel"], exported["predictions/kernel"])
    m.train_step(x, y)
    assert eng.opt_step == 6
    spec = eng.by_name["encoder/layer_0/dense1/bias"]
    assert not spec.decay and eng.by_name["encoder/layer_0/dense1/kernel"].decay
    # set_weights after training replaces what the engine holds
    m.set_weights(w_before)
    np.testing.assert_allclose(m(x).cpu().numpy(), before.cpu().numpy(), rtol=0, atol=0)
