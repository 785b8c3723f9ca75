"""CPU tier: the oracle against INDEPENDENT implementations of the same published algorithms (not the reference, which cannot run
here - SURVEY 8c - but third-party code written by other people):

  * Pillow.  The AutoAugment / RandAugment colour ops were defined through PIL (`ImageOps`, `ImageEnhance`); the TF / TFA functions
    the reference calls are ports that say so in their docstrings.  Invert, Posterize, Solarize, Equalize, AutoContrast and
    Brightness must agree with Pillow BIT FOR BIT (same integer / truncation arithmetic); Color and Sharpness agree to one grey
    level (Pillow's luma conversion is integer ITU-R 601 and its SMOOTH filter rounds, TF's are the float forms the oracle restates).
    Contrast is not compared: the reference's degenerate "mean" is its own quirk (pixels / 256), not PIL's mean.
  * torch's own operators for the ViT pieces: exact-erf GELU, LayerNorm, softmax attention, Adam with decoupled weight decay.
An oracle function that drifted from its published algorithm fails here even though the HIP path (checked against the oracle)
would still pass."""
import math

import numpy as np
import pytest
import torch

from oracle import augment_ref as A
from oracle import vit_ref

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance, ImageOps      # noqa: E402


def _images():
    g = np.random.Generator(np.random.PCG64(7))
    out = []
    for shape in [(64, 48), (33, 50), (224, 224)]:
        x = g.integers(0, 256, size=shape + (3,), dtype=np.uint8)
        out.append(x)
        lo = x.copy()
        lo[: shape[0] // 2] = lo[: shape[0] // 2] // 3 + 40          # a low-contrast half: AutoContrast / Equalize do real work
        out.append(lo)
    out.append(np.full((16, 16, 3), 93, dtype=np.uint8))             # constant image: the identity branches
    return out


def _pil(x):
    return Image.fromarray(x)


@pytest.mark.parametrize("name,ours,theirs", [
    ("Invert", lambda x: A.invert(x), lambda im: ImageOps.invert(im)),
    ("Posterize(3 bits)", lambda x: A.posterize(x, 3), lambda im: ImageOps.posterize(im, 3)),
    ("Posterize(1 bit)", lambda x: A.posterize(x, 1), lambda im: ImageOps.posterize(im, 1)),
    ("Solarize(230)", lambda x: A.solarize(x, 230), lambda im: ImageOps.solarize(im, 230)),
    ("Solarize(0)", lambda x: A.solarize(x, 0), lambda im: ImageOps.solarize(im, 0)),
    ("Equalize", lambda x: A.equalize(x), lambda im: ImageOps.equalize(im)),
    ("AutoContrast", lambda x: A.autocontrast(x), lambda im: ImageOps.autocontrast(im)),
    ("Brightness(1.72)", lambda x: A.brightness(x, 1.72), lambda im: ImageEnhance.Brightness(im).enhance(1.72)),
    ("Brightness(0.28)", lambda x: A.brightness(x, 0.28), lambda im: ImageEnhance.Brightness(im).enhance(0.28)),
])
def test_colour_ops_equal_pillow_bit_for_bit(name, ours, theirs):
    for x in _images():
        got = ours(x[None])[0]
        ref = np.asarray(theirs(_pil(x)))
        assert np.array_equal(got, ref), "%s differs from Pillow on a %s image: %d pixels" % (name, x.shape, int((got != ref).sum()))


@pytest.mark.parametrize("name,ours,theirs", [
    ("Color(1.72)", lambda x: A.color(x, 1.72), lambda im: ImageEnhance.Color(im).enhance(1.72)),
    ("Color(0.28)", lambda x: A.color(x, 0.28), lambda im: ImageEnhance.Color(im).enhance(0.28)),
    ("Sharpness(1.72)", lambda x: A.sharpness(x, 1.72), lambda im: ImageEnhance.Sharpness(im).enhance(1.72)),
    ("Sharpness(0.28)", lambda x: A.sharpness(x, 0.28), lambda im: ImageEnhance.Sharpness(im).enhance(0.28)),
])
def test_colour_ops_within_one_level_of_pillow(name, ours, theirs):
    for x in _images():
        got = ours(x[None])[0].astype(np.int32)
        ref = np.asarray(theirs(_pil(x))).astype(np.int32)
        d = np.abs(got - ref)
        assert d.max() <= 1 and (d == 0).mean() > 0.6, "%s vs Pillow on %s: max %d, exact %.3f" % (name, x.shape, d.max(), (d == 0).mean())


def test_gelu_layernorm_and_attention_equal_torch_operators():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 37, 96, generator=g) * 3
    assert torch.allclose(vit_ref.gelu(x), torch.nn.functional.gelu(x), rtol=0, atol=2e-6)            # exact-erf GELU (activations.py:46-56)
    gamma, beta = torch.randn(96, generator=g), torch.randn(96, generator=g)
    assert torch.allclose(vit_ref.layer_norm(x, gamma, beta, 1e-6), torch.nn.functional.layer_norm(x, (96,), gamma, beta, 1e-6), rtol=0, atol=3e-6)
    heads, hd = 3, 32
    p = {"w_query": torch.randn(96, heads, hd, generator=g) * 0.1, "b_query": torch.randn(heads, 1, hd, generator=g) * 0.1,
         "w_key": torch.randn(96, heads, hd, generator=g) * 0.1, "b_key": torch.randn(heads, 1, hd, generator=g) * 0.1,
         "w_value": torch.randn(96, heads, hd, generator=g) * 0.1, "b_value": torch.randn(heads, 1, hd, generator=g) * 0.1,
         "w_projection": torch.randn(heads, 96, hd, generator=g) * 0.1, "b_projection": torch.randn(1, 96, generator=g) * 0.1}
    ours = vit_ref.multi_head_attention(x, p, "", heads, 0.0, None, False)
    q = torch.einsum("btd,dnh->bnth", x, p["w_query"]) + p["b_query"]
    k = torch.einsum("btd,dnh->bnth", x, p["w_key"]) + p["b_key"]
    v = torch.einsum("btd,dnh->bnth", x, p["w_value"]) + p["b_value"]
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v)                                     # softmax(q k^T / sqrt(hd)) v
    theirs = torch.einsum("bnth,ndh->btd", o, p["w_projection"]) + p["b_projection"]
    assert torch.allclose(ours, theirs, rtol=0, atol=2e-5)


def test_adamw_equals_torch_adamw_with_the_references_decay_convention():
    """keras Adam + the reference's decoupled decay `var -= wd * var` (optimizers.py:147-155: wd is NOT scaled by lr) against
    torch.optim.AdamW, whose decay is `var *= 1 - lr * wd` (so its weight_decay = wd / lr) and whose epsilon sits beside
    sqrt(v / (1 - b2^t)) instead of sqrt(v) - keras' epsilon is in effect eps / sqrt(1 - b2^t) = 3e-6 at step 1, which shows only
    on the few elements whose gradient is that small (update differs by up to 1.5 % of lr there: measured max 1.5e-5 at lr 1e-3,
    0.05 % of elements above 2e-6)."""
    g = torch.Generator().manual_seed(5)
    w0 = torch.randn(257, 33, generator=g)
    grads = [torch.randn(257, 33, generator=g) for _ in range(5)]
    lr, wd = 1e-3, 0.05
    ours = {"w": w0.clone()}
    m, v = {"w": torch.zeros_like(w0)}, {"w": torch.zeros_like(w0)}
    ref = torch.nn.Parameter(w0.clone())
    opt = torch.optim.AdamW([ref], lr=lr, betas=(0.9, 0.999), eps=1e-7, weight_decay=wd / lr)
    for t, gr in enumerate(grads, start=1):
        vit_ref.adamw_step(ours, {"w": gr}, m, v, t, lr=lr, eps=1e-7, weight_decay=wd)
        ref.grad = gr.clone()
        opt.step()
        d = (ours["w"] - ref.detach()).abs()
        assert float(d.max()) < 0.03 * lr and float((d > 2e-6).float().mean()) < 2e-3 and float(d.mean()) < 2e-7, t
    assert float((ours["w"] - w0).abs().max()) > 1e-3 * math.sqrt(2)


@pytest.mark.parametrize("name", ["ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate", "Rotate (negated)"])
def test_warps_equal_scipy_ndimage_away_from_the_half_pixel_border(name):
    """tfa.image.transform semantics as restated by the oracle (output pixel -> input coordinate through the 8-float row, nearest
    neighbour, constant fill 128) against scipy.ndimage.affine_transform(order=0, mode='constant'): identical on every pixel whose
    source coordinate is not within half a pixel OUTSIDE the image - there scipy fills (it tests the unrounded coordinate) while
    TFA rounds first and keeps the edge pixel, which stays the oracle's upstream-recalled choice."""
    ndimage = pytest.importorskip("scipy.ndimage")
    g = np.random.Generator(np.random.PCG64(11))
    for h, w in [(64, 48), (37, 50), (224, 224)]:
        x = g.integers(0, 256, size=(1, h, w, 3), dtype=np.uint8)
        t = {"ShearX": A.shear_x_transform(0.27, False), "ShearY": A.shear_y_transform(0.27, True),
             "TranslateX": A.translate_x_transform(90.0, False), "TranslateY": A.translate_y_transform(9.0, True),
             "Rotate": A.rotate_transform(27.0, False, h, w), "Rotate (negated)": A.rotate_transform(27.0, True, h, w)}[name]
        t = np.asarray(t, dtype=np.float32).reshape(-1)
        ours = A.projective_transform(x, t, 128)[0]
        a0, a1, a2, b0, b1, b2 = (float(v) for v in t[:6])
        mat, off = np.array([[b1, b0], [a1, a0]]), np.array([b2, a2])          # arrays are indexed [y, x]
        ref = np.stack([ndimage.affine_transform(x[0, ..., c].astype(np.float64), mat, offset=off, order=0, mode="constant", cval=128.0)
                        for c in range(3)], axis=-1).astype(np.uint8)
        ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
        ix, iy = (t[0] * xs + t[1] * ys) + t[2], (t[3] * xs + t[4] * ys) + t[5]
        e = 1e-3      # scipy works in float64: a coordinate that is exactly W - 1 in fp32 may sit a hair outside there
        band = ((ix < e) & (ix > -0.5)) | ((ix > w - 1 - e) & (ix < w - 0.5)) | ((iy < e) & (iy > -0.5)) | ((iy > h - 1 - e) & (iy < h - 0.5))
        tie = (np.abs(ix - np.floor(ix) - 0.5) < 1e-4) | (np.abs(iy - np.floor(iy) - 0.5) < 1e-4)      # exact .5: rounding conventions differ
        diff = (ours != ref).any(-1)
        assert not (diff & ~band & ~tie).any(), "%s at %dx%d: %d pixels differ away from the border band" % (name, h, w, int((diff & ~band & ~tie).sum()))
        assert (~band).mean() > 0.9


def test_resize_equals_torch_interpolate():
    """tf.image.resize (TF2: half-pixel centres, no antialiasing) as restated by the oracle against torch's interpolate with the same
    convention (`align_corners=False`, `antialias=False`; `nearest-exact`): nearest identical, bilinear to fp32 rounding (the two
    evaluate the same two lerps in a different association)."""
    g = np.random.Generator(np.random.PCG64(2))
    x = g.integers(0, 256, size=(2, 37, 50, 3), dtype=np.uint8)
    t = torch.from_numpy(x).permute(0, 3, 1, 2).float()
    for oh, ow in [(64, 64), (20, 31), (74, 100), (224, 224)]:
        ref = torch.nn.functional.interpolate(t, size=(oh, ow), mode="bilinear", align_corners=False, antialias=False).permute(0, 2, 3, 1).numpy()
        assert float(np.abs(A.resize(x, oh, ow, "bilinear").astype(np.float64) - ref).max()) < 5e-4
        refn = torch.nn.functional.interpolate(t, size=(oh, ow), mode="nearest-exact").permute(0, 2, 3, 1).numpy()
        assert np.array_equal(A.resize(x, oh, ow, "nearest").astype(np.float32), refn)


def test_encoder_layer_equals_torch_transformer_encoder_layer():
    """The pre-norm block the ViT uses (layers/transformer.py:56-58: x += MHA(LN1 x); x += W2 gelu(W1 LN2 x)) against
    torch.nn.TransformerEncoderLayer(norm_first=True, activation=gelu) - somebody else's implementation of the same block - under
    the weight mapping chambers' einsum layouts imply (w_query [D, heads, hd] = W_q^T etc.)."""
    d, heads, ff, b, t = 96, 3, 192, 2, 29
    hd = d // heads
    g = torch.Generator().manual_seed(9)
    pre = "encoder/layer_0/"
    p = {pre + "multi_head_attention/w_query": torch.randn(d, heads, hd, generator=g) * 0.1, pre + "multi_head_attention/b_query": torch.randn(heads, 1, hd, generator=g) * 0.1,
         pre + "multi_head_attention/w_key": torch.randn(d, heads, hd, generator=g) * 0.1, pre + "multi_head_attention/b_key": torch.randn(heads, 1, hd, generator=g) * 0.1,
         pre + "multi_head_attention/w_value": torch.randn(d, heads, hd, generator=g) * 0.1, pre + "multi_head_attention/b_value": torch.randn(heads, 1, hd, generator=g) * 0.1,
         pre + "multi_head_attention/w_projection": torch.randn(heads, d, hd, generator=g) * 0.1, pre + "multi_head_attention/b_projection": torch.randn(1, d, generator=g) * 0.1,
         pre + "norm1/gamma": 1 + torch.randn(d, generator=g) * 0.1, pre + "norm1/beta": torch.randn(d, generator=g) * 0.1,
         pre + "norm2/gamma": 1 + torch.randn(d, generator=g) * 0.1, pre + "norm2/beta": torch.randn(d, generator=g) * 0.1,
         pre + "dense1/kernel": torch.randn(d, ff, generator=g) * 0.1, pre + "dense1/bias": torch.randn(ff, generator=g) * 0.1,
         pre + "dense2/kernel": torch.randn(ff, d, generator=g) * 0.1, pre + "dense2/bias": torch.randn(d, generator=g) * 0.1}
    x = torch.randn(b, t, d, generator=g)
    ours = vit_ref.encoder_layer(x, p, pre, {"dropout_rate": 0.0, "n_heads": heads, "norm_epsilon": 1e-6}, {}, 0, False)

    layer = torch.nn.TransformerEncoderLayer(d, heads, dim_feedforward=ff, dropout=0.0, activation="gelu", layer_norm_eps=1e-6, batch_first=True,
                                             norm_first=True)
    m = pre + "multi_head_attention/"
    with torch.no_grad():
        wq, wk, wv = (p[m + n].reshape(d, d).t() for n in ("w_query", "w_key", "w_value"))
        layer.self_attn.in_proj_weight.copy_(torch.cat([wq, wk, wv], dim=0))
        layer.self_attn.in_proj_bias.copy_(torch.cat([p[m + n].reshape(d) for n in ("b_query", "b_key", "b_value")]))
        layer.self_attn.out_proj.weight.copy_(p[m + "w_projection"].permute(1, 0, 2).reshape(d, d))
        layer.self_attn.out_proj.bias.copy_(p[m + "b_projection"].reshape(d))
        layer.linear1.weight.copy_(p[pre + "dense1/kernel"].t()); layer.linear1.bias.copy_(p[pre + "dense1/bias"])
        layer.linear2.weight.copy_(p[pre + "dense2/kernel"].t()); layer.linear2.bias.copy_(p[pre + "dense2/bias"])
        layer.norm1.weight.copy_(p[pre + "norm1/gamma"]); layer.norm1.bias.copy_(p[pre + "norm1/beta"])
        layer.norm2.weight.copy_(p[pre + "norm2/gamma"]); layer.norm2.bias.copy_(p[pre + "norm2/beta"])
    layer.train()          # the fused inference fast path is a third implementation; the plain module path is the one to compare
    theirs = layer(x)
    assert torch.allclose(ours, theirs.detach(), rtol=0, atol=3e-5), float((ours - theirs.detach()).abs().max())
    # and the reference's default post-norm composition against norm_first=False
    layer.norm_first = False
    ours_post = vit_ref.encoder_layer_post_norm(x, p, pre, {"dropout_rate": 0.0, "n_heads": heads, "norm_epsilon": 1e-6}, {}, 0, False)
    assert torch.allclose(ours_post, layer(x).detach(), rtol=0, atol=3e-5)


def _hf_to_keras_named(sd, root, d, heads, layers):
    """transformers ViT / DeiT state dict -> the Keras-named tensors the oracle takes, through the layouts of
    test_units/manual_test_vit_weights.py (conv OIHW -> HWIO, Linear weight^T -> [D, heads, hd] einsum kernels)."""
    hd = d // heads
    layer_key = root + ".layers.%d." if root + ".layers.0.layernorm_before.weight" in sd else root + ".encoder.layer.%d."

    def pick(prefix, *names):      # the attribute names of the attention / MLP sub-modules changed between transformers releases
        for n in names:
            if prefix + n + ".weight" in sd:
                return sd[prefix + n + ".weight"], sd[prefix + n + ".bias"]
        raise KeyError(prefix + "|".join(names))

    p = {"patch_embeddings/embedding/kernel": sd[root + ".embeddings.patch_embeddings.projection.weight"].permute(2, 3, 1, 0).contiguous(),
         "patch_embeddings/embedding/bias": sd[root + ".embeddings.patch_embeddings.projection.bias"],
         "add_cls_token/embeddings": sd[root + ".embeddings.cls_token"].reshape(1, d),
         "pos_embedding/embeddings": sd[root + ".embeddings.position_embeddings"].reshape(-1, d),
         "encoder/norm/gamma": sd[root + ".layernorm.weight"], "encoder/norm/beta": sd[root + ".layernorm.bias"]}
    for i in range(layers):
        s, pre = layer_key % i, "encoder/layer_%d/" % i
        for nm, cands in (("query", ("attention.q_proj", "attention.attention.query")), ("key", ("attention.k_proj", "attention.attention.key")),
                          ("value", ("attention.v_proj", "attention.attention.value"))):
            wt, bs = pick(s, *cands)
            p[pre + "multi_head_attention/w_" + nm] = wt.t().reshape(d, heads, hd)
            p[pre + "multi_head_attention/b_" + nm] = bs.reshape(heads, 1, hd)
        wt, bs = pick(s, "attention.o_proj", "attention.output.dense")
        p[pre + "multi_head_attention/w_projection"] = wt.reshape(d, heads, hd).permute(1, 0, 2)
        p[pre + "multi_head_attention/b_projection"] = bs.reshape(1, d)
        p[pre + "norm1/gamma"], p[pre + "norm1/beta"] = sd[s + "layernorm_before.weight"], sd[s + "layernorm_before.bias"]
        p[pre + "norm2/gamma"], p[pre + "norm2/beta"] = sd[s + "layernorm_after.weight"], sd[s + "layernorm_after.bias"]
        wt, bs = pick(s, "mlp.fc1", "intermediate.dense")
        p[pre + "dense1/kernel"], p[pre + "dense1/bias"] = wt.t(), bs
        wt, bs = pick(s, "mlp.fc2", "output.dense")
        p[pre + "dense2/kernel"], p[pre + "dense2/bias"] = wt.t(), bs
    return p


def _randomized(model):
    torch.manual_seed(0)
    model.eval()
    with torch.no_grad():
        for t in model.parameters():
            t.copy_(torch.randn_like(t) * 0.1)
    return model


def test_vit_forward_equals_huggingface_vit():
    """The whole ViT graph of the oracle (vision_transformer.py:235-283: patch conv, class token on the LEFT, learned positions,
    pre-norm blocks, final LayerNorm, class-token pooling, `predictions` head) against transformers' ViTForImageClassification - an
    independent, widely used implementation of the same architecture - with random weights mapped by _hf_to_keras_named."""
    tr = pytest.importorskip("transformers")
    from chambers_amd.engine import ViTConfig
    d, heads, layers, ff, patch, h, w, classes = 96, 3, 2, 192, 16, 64, 48, 10
    cfg = tr.ViTConfig(hidden_size=d, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ff, hidden_act="gelu",
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-6, image_size=(h, w), patch_size=patch,
                       num_channels=3, qkv_bias=True, num_labels=classes)
    model = _randomized(tr.ViTForImageClassification(cfg))
    sd = model.state_dict()
    p = _hf_to_keras_named(sd, "vit", d, heads, layers)
    p["predictions/kernel"], p["predictions/bias"] = sd["classifier.weight"].t(), sd["classifier.bias"]
    x = torch.randn(2, h, w, 3, generator=torch.Generator().manual_seed(1))
    c = ViTConfig(patch_size=patch, patch_dim=d, n_encoder_layers=layers, n_heads=heads, ff_dim=ff, image_size=(h, w), classes=classes, dropout_rate=0.0)
    ours = vit_ref.vit_forward(p, x, c.as_oracle_cfg(), keys=None)
    with torch.no_grad():
        theirs = model(pixel_values=x.permute(0, 3, 1, 2).contiguous()).logits
    assert ours.shape == theirs.shape == (2, classes)
    assert float((ours - theirs).abs().max()) < 2e-4 * float(theirs.abs().max()), float((ours - theirs).abs().max())


def test_distilled_vit_forward_equals_huggingface_deit():
    """DistilledVisionTransformer (vision_transformer.py:295-400: distillation token concatenated first, class token on its left,
    so the sequence is [cls, dist, patches]; heads `predictions` on row 0 and `predictions_dist` on row 1; their average when
    return_dist_token=False) against transformers' DeiTForImageClassificationWithTeacher."""
    tr = pytest.importorskip("transformers")
    from chambers_amd.engine import ViTConfig
    d, heads, layers, ff, patch, size, classes = 96, 3, 2, 192, 16, 64, 10
    cfg = tr.DeiTConfig(hidden_size=d, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ff, hidden_act="gelu",
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-6, image_size=size, patch_size=patch,
                        num_channels=3, qkv_bias=True, num_labels=classes)
    model = _randomized(tr.DeiTForImageClassificationWithTeacher(cfg))
    sd = model.state_dict()
    p = _hf_to_keras_named(sd, "deit", d, heads, layers)
    p["add_dist_token/embeddings"] = sd["deit.embeddings.distillation_token"].reshape(1, d)
    p["predictions/kernel"], p["predictions/bias"] = sd["cls_classifier.weight"].t(), sd["cls_classifier.bias"]
    p["predictions_dist/kernel"], p["predictions_dist/bias"] = sd["distillation_classifier.weight"].t(), sd["distillation_classifier.bias"]
    x = torch.randn(2, size, size, 3, generator=torch.Generator().manual_seed(2))
    c = ViTConfig(patch_size=patch, patch_dim=d, n_encoder_layers=layers, n_heads=heads, ff_dim=ff, image_size=(size, size), classes=classes,
                  dropout_rate=0.0, distilled=True)
    ours_cls, ours_dist = vit_ref.vit_forward(p, x, dict(c.as_oracle_cfg(), return_dist_token=True), keys=None)
    ours_avg = vit_ref.vit_forward(p, x, dict(c.as_oracle_cfg(), return_dist_token=False), keys=None)
    with torch.no_grad():
        out = model(pixel_values=x.permute(0, 3, 1, 2).contiguous())
    scale = float(out.logits.abs().max())
    for a, b in ((ours_cls, out.cls_logits), (ours_dist, out.distillation_logits), (ours_avg, out.logits)):
        assert float((a - b).abs().max()) < 2e-4 * scale


@pytest.mark.parametrize("negate", [False, True])
def test_rotate_direction_and_centre_equal_scipy_rotate(negate):
    """tfa.image.rotate: counter-clockwise for a positive angle, about the centre ((W-1)/2, (H-1)/2) - the oracle's matrix
    (augment_ref.rotate_transform, "upstream restated") against scipy.ndimage.rotate, which builds its own matrix from the angle:
    Rotate(27 deg) is scipy's +27 deg and the negated draw is -27 deg, identical away from the half-pixel border band."""
    ndimage = pytest.importorskip("scipy.ndimage")
    g = np.random.Generator(np.random.PCG64(3))
    for h, w in [(64, 48), (51, 51), (224, 224)]:
        x = g.integers(0, 256, size=(1, h, w, 3), dtype=np.uint8)
        t = np.asarray(A.rotate_transform(27.0, negate, h, w), dtype=np.float32).reshape(-1)
        ours = A.projective_transform(x, t, 128)[0]
        ref = ndimage.rotate(x[0].astype(np.float64), -27.0 if negate else 27.0, axes=(1, 0), reshape=False, order=0, mode="constant",
                             cval=128.0).astype(np.uint8)
        wrong = ndimage.rotate(x[0].astype(np.float64), 27.0 if negate else -27.0, axes=(1, 0), reshape=False, order=0, mode="constant",
                               cval=128.0).astype(np.uint8)
        ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
        ix, iy = (t[0] * xs + t[1] * ys) + t[2], (t[3] * xs + t[4] * ys) + t[5]
        e = 1e-3
        band = ((ix < e) & (ix > -0.5)) | ((ix > w - 1 - e) & (ix < w - 0.5)) | ((iy < e) & (iy > -0.5)) | ((iy > h - 1 - e) & (iy < h - 0.5))
        tie = (np.abs(ix - np.floor(ix) - 0.5) < 1e-4) | (np.abs(iy - np.floor(iy) - 0.5) < 1e-4)
        diff = (ours != ref).any(-1)
        assert not (diff & ~band & ~tie).any(), (h, w, int((diff & ~band & ~tie).sum()))
        assert (ours != wrong).any(-1).mean() > 0.5          # the other direction is a different image
