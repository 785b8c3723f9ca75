"""GPU: the stand-alone Keras-style layers are TRAINABLE (missing 1).  A model is composed from the LAYERS - not from
ViTEngine - differentiated with `loss.backward()` (torch's tape over torch.autograd.Function wrappers whose forward and backward are
HIP library calls, chambers_amd/layers/autograd.py) and stepped with AdamW.apply_gradients.  Every parameter gradient is held against
the oracle's autograd (oracle/vit_ref.py with the build's rounding points emulated) within the bounds the engine's own tests use, and
against the whole-model engine on the same weights, inputs and dropout keys.  Then each layer by itself against a plain PyTorch
fp64 reference of the same op on the same bf16-rounded operands."""
import numpy as np
import pytest
import torch

from oracle import augment_ref as A
from oracle import rng_ref, vit_ref

pytestmark = pytest.mark.gpu

B_GRAD_EMU = 2.15e-2       # the engine tests' bound for any single weight gradient of a 2-block toy model vs the emulating oracle
B_LOGITS_EMU = 5.8e-3


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


class LayerViT:
    """VisionTransformer graph (vision_transformer.py:235-283) assembled from chambers_amd.layers, weights in the Keras names."""

    def __init__(self, cfg):
        from chambers_amd import initializers
        from chambers_amd.layers.core import Conv2D, Dense, Dropout, LayerNormalization
        from chambers_amd.layers.embedding import ConcatEmbedding, LearnedEmbedding1D
        from chambers_amd.layers.transformer import EncoderLayer
        self.cfg = cfg
        d = cfg.patch_dim
        tn = initializers.TruncatedNormal(stddev=0.02)
        self.embed = Conv2D(filters=d, kernel_size=cfg.patch_size, strides=cfg.patch_size, padding="valid", name="embedding")
        self.add_cls = ConcatEmbedding(n_embeddings=1, embedding_dim=d, side="left", axis=1, initializer=tn, name="add_cls_token")
        self.pos = LearnedEmbedding1D(initializer=tn, name="pos_embedding")
        self.drop = Dropout(cfg.dropout_rate)
        self.blocks = [EncoderLayer(embed_dim=d, num_heads=cfg.n_heads, ff_dim=cfg.ff_dim, attention_dropout_rate=cfg.dropout_rate,
                                    dense_dropout_rate=cfg.dropout_rate, pre_norm=True) for _ in range(cfg.n_encoder_layers)]
        self.norm = LayerNormalization(epsilon=1e-6)
        self.head = Dense(cfg.classes, name="predictions")
        h, w = cfg.image_size
        self.embed.build((None, h, w, 3)); self.embed.built = True
        self.add_cls.build((None, cfg.n_patches, d)); self.add_cls.built = True
        self.pos.build((None, cfg.n_tokens, d)); self.pos.built = True
        for b in self.blocks:
            b.build((None, cfg.n_tokens, d)); b.built = True
        self.norm.build((None, cfg.n_tokens, d)); self.norm.built = True
        self.head.build((None, d)); self.head.built = True

    def named_variables(self):
        out = {"patch_embeddings/embedding/kernel": self.embed.kernel, "patch_embeddings/embedding/bias": self.embed.bias,
               "add_cls_token/embeddings": self.add_cls.embedding, "pos_embedding/embeddings": self.pos.embedding}
        for i, l in enumerate(self.blocks):
            p = "encoder/layer_%d/" % i
            m = l.multi_head_attention
            for nm in ("w_query", "b_query", "w_value", "b_value", "w_key", "b_key", "w_projection", "b_projection"):
                out[p + "multi_head_attention/" + nm] = getattr(m, nm)
            out[p + "norm1/gamma"], out[p + "norm1/beta"] = l.norm1.gamma, l.norm1.beta
            out[p + "dense1/kernel"], out[p + "dense1/bias"] = l.dense1.kernel, l.dense1.bias
            out[p + "dense2/kernel"], out[p + "dense2/bias"] = l.dense2.kernel, l.dense2.bias
            out[p + "norm2/gamma"], out[p + "norm2/beta"] = l.norm2.gamma, l.norm2.beta
        out["encoder/norm/gamma"], out["encoder/norm/beta"] = self.norm.gamma, self.norm.beta
        out["predictions/kernel"], out["predictions/bias"] = self.head.kernel, self.head.bias
        return out

    def assign(self, kw):
        for k, v in self.named_variables().items():
            v.assign(kw[k])

    def __call__(self, images_u8, training, keys):
        from chambers_amd.layers import autograd as AG
        from chambers_amd import rng
        cfg = self.cfg
        b = images_u8.shape[0]
        x = self.embed(images_u8).reshape(b, cfg.n_patches, cfg.patch_dim)       # uint8 in: ImageNetNormalization("tf") fused in front
        x = self.pos(self.add_cls(x))
        x = self.drop(x, training=training, key=keys.get(rng.SITE_EMBED))
        for l, blk in enumerate(self.blocks):
            x = blk(x, training=training, keys={"attn": keys.get(rng.site_attn(l)), "proj": keys.get(rng.site_proj(l)), "mlp": keys.get(rng.site_mlp(l))})
        x = self.norm(x)
        return self.head(AG.TakeTokenFn.apply(x, 0))


def _toy():
    from chambers_amd.engine import ViTConfig, init_keras_weights
    cfg = ViTConfig(patch_size=16, patch_dim=128, n_encoder_layers=2, n_heads=2, ff_dim=256, dropout_rate=0.1, image_size=(64, 48), classes=10)
    kw = init_keras_weights(cfg, seed=1234)
    g = np.random.Generator(np.random.PCG64(0))
    for k in kw:
        if k.endswith(("bias", "beta", "b_query", "b_key", "b_value", "b_projection")):
            kw[k] = g.normal(0, 0.05, size=kw[k].shape).astype(np.float32)
        if k.endswith("gamma"):
            kw[k] = (1.0 + g.normal(0, 0.1, size=kw[k].shape)).astype(np.float32)
    return cfg, kw, g


def test_model_composed_from_layers_trains_and_matches_oracle_and_engine():
    from chambers_amd.engine import ViTEngine
    from chambers_amd.optimizers import AdamW
    cfg, kw, g = _toy()
    bsz, seed = 4, 3
    images = g.integers(0, 256, size=(bsz,) + cfg.image_size + (3,), dtype=np.uint8)
    labels = torch.as_tensor(g.integers(0, cfg.classes, size=(bsz,)))
    keys = {s: rng_ref.site_key(seed, 0, s) for s in range(1 + 3 * cfg.n_encoder_layers)}
    model = LayerViT(cfg)
    model.assign(kw)
    logits = model(torch.as_tensor(images, device="cuda"), True, keys)
    assert logits.shape == (bsz, cfg.classes) and logits.requires_grad
    loss = torch.nn.functional.cross_entropy(logits, labels.cuda())
    loss.backward()
    # the oracle, rounding where the build rounds, differentiated by torch on the CPU
    p = {k: torch.tensor(v, requires_grad=True) for k, v in kw.items()}
    ref = vit_ref.vit_forward(p, torch.from_numpy(A.imagenet_normalize(images, "tf")), cfg.as_oracle_cfg(), keys=keys, bf16=True)
    torch.nn.functional.cross_entropy(ref, labels).backward()
    assert rel_l2(logits.detach(), ref.detach()) < B_LOGITS_EMU
    # the whole-model engine on the same weights / images / keys
    eng = ViTEngine(cfg, bsz, training=True, seed=seed)
    eng.load_keras_weights(kw)
    elog = eng.forward(torch.as_tensor(images, device="cuda"), training=True).clone()
    eng.loss(labels.cuda())
    eng.backward()
    egrads = eng.export_keras_grads()
    assert rel_l2(logits.detach(), elog) < 1e-5
    worst = {}
    for name, var in model.named_variables().items():
        assert var.grad is not None, "no gradient for " + name
        got = var.grad.detach().cpu()
        assert got.shape == p[name].shape
        if name.endswith("b_key"):
            continue           # exactly zero in exact arithmetic (softmax shift invariance): noise against noise
        worst[name] = (rel_l2(got, p[name].grad), rel_l2(got, egrads[name]))
        assert worst[name][0] < B_GRAD_EMU, "%s: rel-L2 %.3g vs the oracle" % (name, worst[name][0])
        assert worst[name][1] < 2e-3, "%s: rel-L2 %.3g vs the engine's gradient" % (name, worst[name][1])
    # a few optimizer steps on the layer model: the loss falls, AdamW.apply_gradients updates every variable
    opt = AdamW(0.01, decay_exclude=["bias", "/b_", "gamma", "beta", "embeddings"], learning_rate=2e-3)
    variables = list(model.named_variables().values())
    before = {n: v.numpy().copy() for n, v in model.named_variables().items()}
    losses = [float(loss)]
    for step in range(1, 6):
        opt.apply_gradients([(None, v) for v in variables])
        for v in variables:
            v.zero_grad()
        k2 = {s: rng_ref.site_key(seed, step, s) for s in range(1 + 3 * cfg.n_encoder_layers)}
        l2 = torch.nn.functional.cross_entropy(model(torch.as_tensor(images, device="cuda"), True, k2), labels.cuda())
        l2.backward()
        losses.append(float(l2))
    assert losses[-1] < losses[0] - 0.05, losses
    moved = [n for n, v in model.named_variables().items() if not np.array_equal(v.numpy(), before[n])]
    assert len(moved) == len(before)


def test_apply_gradients_equals_the_oracle_adamw_step():
    from chambers_amd._keras_like import Variable
    from chambers_amd.optimizers import AdamW
    g = torch.Generator().manual_seed(5)
    w0 = torch.randn(64, 48, generator=g)
    b0 = torch.randn(50, generator=g)            # odd size: padded update
    vw, vb = Variable("dense/kernel:0", w0.clone().cuda()), Variable("dense/bias:0", b0.clone().cuda())
    opt = AdamW(0.05, decay_exclude=["bias"], learning_rate=1e-2)
    params = {"w": w0.clone(), "b": b0.clone()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(v) for k, v in params.items()}
    for step in range(1, 4):
        gw, gb = torch.randn(64, 48, generator=g), torch.randn(50, generator=g)
        opt.apply_gradients([(gw.cuda(), vw), (gb.cuda(), vb)])
        vit_ref.adamw_step(params, {"w": gw, "b": gb}, m, v, step, lr=1e-2, weight_decay=0.05, decay_mask={"w": True, "b": False})
    pw, pb = params["w"].numpy(), params["b"].numpy()
    np.testing.assert_allclose(vw.numpy(), pw, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(vb.numpy(), pb, rtol=2e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ single layers vs plain PyTorch
def _grads(fn, *tensors, seed=123):
    """Run fn, back-propagate a fixed random dy (no cancellation structure in its column sums); returns (output, dy)."""
    for t in tensors:
        t.grad = None
    out = fn()
    dy = torch.randn(out.shape, generator=torch.Generator().manual_seed(seed)).cuda()
    out.backward(dy.to(out.dtype))
    return out, dy


@pytest.mark.parametrize("activation,units,m", [(None, 64, 48), ("gelu", 128, 70), ("tanh", 32, 5), (None, 10, 33)])
def test_dense_is_differentiable(activation, units, m):
    from chambers_amd.layers.core import Dense
    g = torch.Generator().manual_seed(7)
    x = torch.randn(m, 64, generator=g).cuda().requires_grad_(True)
    layer = Dense(units, activation=activation)
    y = layer(x)
    w, b = layer.kernel.value, layer.bias.value
    with torch.no_grad():
        b.copy_(torch.randn(units, generator=g) * 0.1)
    y, dy = _grads(lambda: layer(x), x, w, b)
    xr = bf(x.detach()).double().requires_grad_(True)
    wr = bf(w.detach()).double().requires_grad_(True)
    br = b.detach().double().requires_grad_(True)
    z = xr @ wr + br
    ref = torch.nn.functional.gelu(z) if activation == "gelu" else torch.tanh(z) if activation == "tanh" else z
    # dY is a bf16 GEMM operand in the layer's backward: without an activation the reference takes the same rounded dY and the
    # gradients agree to fp32 accumulation; with one, dz = bf16(dy * act') is rounded after the product (bf16-level agreement)
    ref.backward((bf(dy) if activation is None else dy).double())
    tol = 2e-5 if activation is None else 6e-3
    assert y.dtype == torch.float32 and rel_l2(y.detach(), ref.detach()) < 1e-5
    assert rel_l2(x.grad, xr.grad) < tol and rel_l2(w.grad, wr.grad) < tol and rel_l2(b.grad, br.grad) < tol


def test_dense_operand_cache_follows_every_kind_of_weight_write():
    """repeated inference calls of a stand-alone layer re-use the bf16 images of its kernel (no tape, no padded copies);
    the cache must notice a torch write (set_weights / copy_: the tensor's version counter), the optimizer's HIP update (raw pointers:
    AdamW.apply_gradients calls weights_written()) and a NEW tensor that re-uses a freed one's address."""
    from chambers_amd.layers import autograd as AG
    from chambers_amd.layers.core import Dense
    from chambers_amd.optimizers import AdamW
    g = torch.Generator().manual_seed(13)
    x = torch.randn(40, 64, generator=g).cuda()
    layer = Dense(96)
    layer(x)                                     # build
    w = layer.kernel.value

    def ref():
        return (bf(x).double() @ bf(w.detach()).double() + layer.bias.value.detach().double()).float()

    with torch.no_grad():
        y1 = layer(x)
        n_cached = len(AG._OPERANDS)
        y2 = layer(x)
        assert len(AG._OPERANDS) == n_cached and torch.equal(y1, y2) and rel_l2(y1, ref()) < 1e-5
        w.copy_(torch.randn(w.shape, generator=g) * 0.1)            # torch write
        assert rel_l2(layer(x), ref()) < 1e-5
    xg = x.clone().requires_grad_(True)
    layer(xg).square().mean().backward()
    before = w.detach().clone()
    AdamW(learning_rate=1e-2, weight_decay=0.0).apply_gradients([(None, v) for v in layer.trainable_weights])     # HIP write
    assert not torch.equal(before, w.detach())
    with torch.no_grad():
        assert rel_l2(layer(x), ref()) < 1e-5
    for _ in range(4):                           # fresh layers whose kernels may land on a freed kernel's address
        other = Dense(96)
        with torch.no_grad():
            yo = other(x)
            wo = other.kernel.value
            assert rel_l2(yo, (bf(x).double() @ bf(wo.detach()).double() + other.bias.value.detach().double()).float()) < 1e-5
        del other, wo, yo


def test_layernorm_dropout_embeddings_and_gelu_are_differentiable():
    from chambers_amd import activations
    from chambers_amd.layers.core import Dropout, LayerNormalization
    from chambers_amd.layers.embedding import ConcatEmbedding, LearnedEmbedding1D
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 5, 64, generator=g).cuda().requires_grad_(True)
    ln = LayerNormalization(epsilon=1e-6)
    ln(x)
    with torch.no_grad():
        ln.gamma.value.copy_(1.0 + 0.1 * torch.randn(64, generator=g)); ln.beta.value.copy_(0.1 * torch.randn(64, generator=g))
    y, dy = _grads(lambda: ln(x), x, ln.gamma.value, ln.beta.value)
    xr = x.detach().double().requires_grad_(True)
    gr, br = ln.gamma.value.detach().double().requires_grad_(True), ln.beta.value.detach().double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (64,), gr, br, 1e-6)
    ref.backward(bf(dy).double())            # dy reaches the kernel as bf16 (the layer's output dtype): the reference takes the same
    assert y.dtype == torch.bfloat16 and rel_l2(y.detach().float(), ref.detach()) < 4e-3
    assert rel_l2(x.grad, xr.grad) < 2e-5 and rel_l2(ln.gamma.value.grad, gr.grad) < 2e-5 and rel_l2(ln.beta.value.grad, br.grad) < 2e-5
    # Dropout: the mask is the counter-hash definition on the flat index; backward is the same map
    drop = Dropout(0.25)
    key = 0x1234
    y, dy = _grads(lambda: drop(x, training=True, key=key), x)
    keep = torch.from_numpy(rng_ref.keep_mask(x.numel(), key, 0.25)).reshape(x.shape).cuda()
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(0.25)))
    assert torch.equal(y.detach(), torch.where(keep, x.detach() * scale, torch.zeros_like(x.detach())))
    assert torch.equal(x.grad, torch.where(keep, dy * scale, torch.zeros_like(dy)))
    assert drop(x, training=False) is x
    # LearnedEmbedding1D / ConcatEmbedding
    pos = LearnedEmbedding1D(name="pos")
    cat = ConcatEmbedding(2, 64, axis=1, side="left", name="cat")
    pos(cat(x))
    e, t = cat.embedding.value, pos.embedding.value
    y, dy = _grads(lambda: pos(cat(x)), x, e, t)
    ref = torch.cat([e.detach().unsqueeze(0).expand(3, -1, -1), x.detach()], dim=1) + t.detach()
    assert y.shape == (3, 7, 64) and torch.allclose(y.detach(), ref, rtol=1e-6, atol=1e-7)
    assert torch.equal(x.grad, dy[:, 2:])
    assert torch.allclose(e.grad, dy[:, :2].sum(0), rtol=1e-5, atol=1e-6) and torch.allclose(t.grad, dy.sum(0), rtol=1e-5, atol=1e-6)
    right = ConcatEmbedding(1, 64, axis=1, side="right", name="cat_r")
    yr = right(x)
    assert torch.equal(yr[:, :5].detach(), x.detach()) and torch.equal(yr[:, 5].detach(), right.embedding.value.detach().expand(3, 64))
    # chambers.activations.gelu, both forms
    for approx in (False, True):
        xx = torch.randn(257, generator=g).cuda().requires_grad_(True)
        y, dy = _grads(lambda: activations.gelu(xx, approximate=approx), xx)
        xr = xx.detach().double().requires_grad_(True)
        ref = torch.nn.functional.gelu(xr, approximate="tanh" if approx else "none")
        ref.backward(dy.double())
        assert torch.allclose(y.detach().double(), ref.detach(), rtol=1e-5, atol=2e-6) and torch.allclose(xx.grad.double(), xr.grad, rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("pre_norm", [True, False])
def test_attention_and_encoder_layer_are_differentiable(pre_norm):
    """MultiHeadAttention and EncoderLayer (pre- and post-norm) against oracle/vit_ref.py's block on the same weights, inference mode
    (no dropout) so that no mask definition enters: outputs and every gradient."""
    from chambers_amd.layers.transformer import EncoderLayer
    g = torch.Generator().manual_seed(11)
    b, t, d, h, ff = 2, 17, 128, 2, 256
    layer = EncoderLayer(embed_dim=d, num_heads=h, ff_dim=ff, pre_norm=pre_norm)
    x = torch.randn(b, t, d, generator=g).cuda().requires_grad_(True)
    layer(x)
    names = ["w_query", "b_query", "w_value", "b_value", "w_key", "b_key", "w_projection", "b_projection"]
    m = layer.multi_head_attention
    with torch.no_grad():
        for nm in names:
            if nm.startswith("b_"):
                getattr(m, nm).value.copy_(0.05 * torch.randn(getattr(m, nm).shape, generator=g))
        for ln in (layer.norm1, layer.norm2):
            ln.gamma.value.copy_(1.0 + 0.1 * torch.randn(d, generator=g)); ln.beta.value.copy_(0.05 * torch.randn(d, generator=g))
        layer.dense1.bias.value.copy_(0.05 * torch.randn(ff, generator=g)); layer.dense2.bias.value.copy_(0.05 * torch.randn(d, generator=g))
    variables = {"multi_head_attention/" + nm: getattr(m, nm) for nm in names}
    variables.update({"norm1/gamma": layer.norm1.gamma, "norm1/beta": layer.norm1.beta, "norm2/gamma": layer.norm2.gamma, "norm2/beta": layer.norm2.beta,
                      "dense1/kernel": layer.dense1.kernel, "dense1/bias": layer.dense1.bias, "dense2/kernel": layer.dense2.kernel,
                      "dense2/bias": layer.dense2.bias})
    for v in variables.values():
        v.zero_grad()
    x.grad = None
    y = layer(x, training=False)
    lin = torch.linspace(-1.0, 1.0, y.numel(), device="cuda").reshape(y.shape)
    (y * lin).sum().backward()
    p = {"blk/" + k: v.value.detach().cpu().clone().requires_grad_(True) for k, v in variables.items()}
    xr = x.detach().cpu().clone().requires_grad_(True)
    cfg = {"n_heads": h, "dropout_rate": 0.0, "norm_epsilon": 1e-6}
    fn = vit_ref.encoder_layer if pre_norm else vit_ref.encoder_layer_post_norm
    ref = fn(xr, p, "blk/", cfg, {}, 0, True)
    (ref * lin.cpu()).sum().backward()
    assert y.dtype == torch.float32 and rel_l2(y.detach(), ref.detach()) < 4e-3
    assert rel_l2(x.grad, xr.grad) < B_GRAD_EMU
    for k, v in variables.items():
        if k.endswith("b_key"):
            continue
        assert rel_l2(v.grad, p["blk/" + k].grad) < B_GRAD_EMU, k


# ------------------------------------------------------------------------------------------------ masks / causal / cross-attention
def _keras_attention(q, k, v, vmask=None, qmask=None, causal=False, keep=None, inv_keep=1.0):
    """keras Attention (BaseDenseAttention._apply_scores) under ScaledAttention, in torch fp64: [B, H, T, hd] tensors."""
    s = torch.matmul(q, k.transpose(-1, -2)) / np.sqrt(q.shape[-1])
    b, _h, tq, tk = s.shape
    m = torch.ones(b, 1, tq, tk, dtype=torch.bool, device=s.device)
    if vmask is not None:
        m = m & vmask[:, None, None, :].bool()
    if causal:
        m = m & torch.tril(torch.ones(tq, tk, dtype=torch.bool, device=s.device))
    s = s - 1e9 * (~m).to(s.dtype)
    w = torch.softmax(s, dim=-1)
    if keep is not None:
        w = w * keep.to(w.dtype) * inv_keep
    out = torch.matmul(w, v)
    if qmask is not None:
        out = out * qmask[:, None, :, None].to(out.dtype)
    return out


@pytest.mark.parametrize("causal,use_masks,tq,tk,hd,rate", [(False, True, 9, 13, 64, 0.0), (True, False, 12, 12, 32, 0.0), (True, True, 7, 7, 64, 0.0),
                                                          (False, True, 5, 70, 16, 0.25), (False, False, 6, 11, 128, 0.0)])
def test_scaled_attention_masks_causal_cross_attention(causal, use_masks, tq, tk, hd, rate):
    """ScaledAttention with the call arguments the ViT never uses (layers/attention.py:7-23 on keras Attention): value mask, query
    mask, causal mask, key / value length != query length, head widths other than 64 - forward and every gradient against keras
    Attention's definition in fp64 (operands are bf16 on both sides)."""
    from chambers_amd.layers.attention import ScaledAttention
    g = torch.Generator().manual_seed(31)
    b, h = 2, 3
    q = bf(torch.randn(b, h, tq, hd, generator=g)).cuda().requires_grad_(True)
    k = bf(torch.randn(b, h, tk, hd, generator=g)).cuda().requires_grad_(True)
    v = bf(torch.randn(b, h, tk, hd, generator=g)).cuda().requires_grad_(True)
    vmask = qmask = None
    if use_masks:
        vmask = (torch.rand(b, tk, generator=g) > 0.3)
        vmask[:, 0] = True
        qmask = (torch.rand(b, tq, generator=g) > 0.3)
    layer = ScaledAttention(key_dim=hd, causal=causal, dropout=rate)
    key = 0x77
    out = layer([q, v, k], mask=[qmask, vmask] if use_masks else None, training=rate > 0, key=key)
    dy = torch.randn(out.shape, generator=g).cuda()
    out.backward(dy.to(out.dtype))
    keep, inv_keep = None, 1.0
    if rate:
        tk4 = (tk + 3) // 4 * 4
        keep = torch.from_numpy(rng_ref.keep_mask(b * h * tq * tk4, key, rate)).reshape(b, h, tq, tk4)[..., :tk].cuda()
        inv_keep = float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate)))
    qr, kr, vr = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    ref = _keras_attention(qr, kr, vr, None if vmask is None else vmask.cuda(), None if qmask is None else qmask.cuda(), causal, keep, inv_keep)
    ref.backward(bf(dy).double())
    assert out.shape == (b, h, tq, hd) and rel_l2(out.detach().float(), ref.detach()) < 4e-3            # one bf16 rounding of the output
    for got, want, name in ((q.grad, qr.grad, "dq"), (k.grad, kr.grad, "dk"), (v.grad, vr.grad, "dv")):
        assert rel_l2(got.float(), want) < 8e-3, name                                                  # bf16 o enters delta; gradients stored bf16
    if qmask is not None:
        assert not bool(out.detach()[~qmask.cuda()[:, None, :].expand(b, h, tq)].any())                # masked queries give zero rows


@pytest.mark.parametrize("hd,key_dim,t", [(64, 16, 12), (32, 64, 9)])
def test_scaled_attention_divides_by_sqrt_key_dim_whatever_the_tensor_width(hd, key_dim, t):
    """the reference divides the scores by sqrt(key_dim) whenever key_dim is given (layers/attention.py:8-22), also when it
    differs from the width of the tensors - a layer that builds and runs there must run here, with that scale (general kernel)."""
    from chambers_amd.layers.attention import ScaledAttention
    g = torch.Generator().manual_seed(41)
    b, h = 2, 2
    q = bf(torch.randn(b, h, t, hd, generator=g)).cuda().requires_grad_(True)
    k = bf(torch.randn(b, h, t, hd, generator=g)).cuda().requires_grad_(True)
    v = bf(torch.randn(b, h, t, hd, generator=g)).cuda().requires_grad_(True)
    out = ScaledAttention(key_dim=key_dim)([q, v, k])
    dy = torch.randn(out.shape, generator=g).cuda()
    out.backward(dy.to(out.dtype))
    qr, kr, vr = (x.detach().double().requires_grad_(True) for x in (q, k, v))
    ref = torch.matmul(torch.softmax(torch.matmul(qr, kr.transpose(-1, -2)) / np.sqrt(key_dim), dim=-1), vr)
    ref.backward(bf(dy).double())
    assert rel_l2(out.detach().float(), ref.detach()) < 4e-3
    for got, want, name in ((q.grad, qr.grad, "dq"), (k.grad, kr.grad, "dk"), (v.grad, vr.grad, "dv")):
        assert rel_l2(got.float(), want) < 8e-3, name


def test_multi_head_attention_cross_attention_with_masks_trains():
    """MultiHeadAttention.call(inputs=[q, v, k], mask=[query_mask, value_mask]) with different query / memory sequences
    (layers/attention.py:99-145), causal=False: output and the gradient of every weight against the einsum definition in fp64."""
    from chambers_amd.layers.attention import MultiHeadAttention
    g = torch.Generator().manual_seed(37)
    b, tq, tk, d, heads, hd = 2, 10, 23, 128, 2, 64
    xq = torch.randn(b, tq, d, generator=g).cuda().requires_grad_(True)
    mem = torch.randn(b, tk, d, generator=g).cuda().requires_grad_(True)
    vmask = torch.rand(b, tk, generator=g) > 0.25
    vmask[:, 0] = True
    qmask = torch.rand(b, tq, generator=g) > 0.2
    mha = MultiHeadAttention(head_dim=hd, num_heads=heads, dropout_rate=0.0)
    mha([xq, mem, mem], mask=[qmask, vmask])
    names = ["w_query", "b_query", "w_value", "b_value", "w_key", "b_key", "w_projection", "b_projection"]
    with torch.no_grad():
        for nm in names:
            if nm.startswith("b_"):
                getattr(mha, nm).value.copy_(0.05 * torch.randn(getattr(mha, nm).shape, generator=g))
    for nm in names:
        getattr(mha, nm).zero_grad()
    xq.grad = mem.grad = None
    out = mha([xq, mem, mem], mask=[qmask, vmask], training=False)
    dy = torch.randn(out.shape, generator=g).cuda()
    out.backward(dy)
    w = {nm: getattr(mha, nm).value.detach().double().requires_grad_(True) for nm in names}
    xr, mr = bf(xq.detach()).double().requires_grad_(True), bf(mem.detach()).double().requires_grad_(True)
    rb = lambda t: bf(t.detach().float()).double() + (t - t.detach())   # noqa: E731  value rounded to bf16, gradient straight through (once)
    query = torch.einsum("btd,dnh->bnth", xr, rb(w["w_query"])) + w["b_query"]
    value = torch.einsum("btd,dnh->bnth", mr, rb(w["w_value"])) + w["b_value"]
    keyt = torch.einsum("btd,dnh->bnth", mr, rb(w["w_key"])) + w["b_key"]
    att = _keras_attention(rb(query), rb(keyt), rb(value), vmask.cuda(), qmask.cuda(), False)
    ref = torch.einsum("bnth,ndh->btd", rb(att), rb(w["w_projection"])) + w["b_projection"]
    ref.backward(bf(dy).double())
    assert out.dtype == torch.float32 and rel_l2(out.detach(), ref.detach()) < 4e-3
    assert rel_l2(xq.grad, xr.grad) < 1.5e-2 and rel_l2(mem.grad, mr.grad) < 1.5e-2
    for nm in names:
        if nm != "b_key":          # exactly zero in exact arithmetic (softmax shift invariance): noise against noise
            assert rel_l2(getattr(mha, nm).value.grad, w[nm].grad) < 1.5e-2, nm
    assert mha.compute_mask([xq, mem, mem], mask=[qmask, vmask]) is qmask


def test_encoder_layer_with_a_padding_mask():
    """EncoderLayer.call(x, mask) (layers/transformer.py:53-68: self-attention with mask=[mask, mask]): padded positions do not
    influence the other positions' outputs, and the layer stays differentiable."""
    from chambers_amd.layers.transformer import EncoderLayer
    g = torch.Generator().manual_seed(41)
    b, t, d = 2, 12, 128
    layer = EncoderLayer(embed_dim=d, num_heads=2, ff_dim=256, pre_norm=True)
    x = torch.randn(b, t, d, generator=g).cuda()
    mask = torch.ones(b, t, dtype=torch.bool)
    mask[:, 9:] = False
    y1 = layer(x, mask=mask, training=False)
    x2 = x.clone()
    x2[:, 9:] = torch.randn(b, 3, d, generator=g).cuda() * 5.0        # change only the padded positions
    y2 = layer(x2, mask=mask, training=False)
    assert torch.allclose(y1[:, :9].detach(), y2[:, :9].detach(), rtol=0, atol=0)
    xg = x.clone().requires_grad_(True)
    layer(xg, mask=mask, training=False)[:, :9].sum().backward()
    assert xg.grad is not None and float(xg.grad[:, :9].abs().sum()) > 0 and layer.dense1.kernel.value.grad is not None
