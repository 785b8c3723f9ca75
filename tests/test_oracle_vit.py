"""CPU tier: pins the ViT oracle to the one equivalence the reference states for this path — a chambers EncoderLayer
equals the timm Block under the weight mapping of test_units/manual_test_vit_weights.py:27-76 (norm1/MHA/MLP atol 1e-5,
whole block 1e-4, :252-279) — plus autograd sanity and the AdamW restatement against a hand-computed step."""
import numpy as np
import torch

from oracle import vit_ref


def _timm_weights(d, ff, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * 0.05  # noqa: E731
    return {"norm1.weight": 1 + r(d), "norm1.bias": r(d), "attn.qkv.weight": r(3 * d, d), "attn.qkv.bias": r(3 * d),
            "attn.proj.weight": r(d, d), "attn.proj.bias": r(d), "norm2.weight": 1 + r(d), "norm2.bias": r(d),
            "mlp.fc1.weight": r(ff, d), "mlp.fc1.bias": r(ff), "mlp.fc2.weight": r(d, ff), "mlp.fc2.bias": r(d)}


def test_encoder_layer_equals_timm_block_under_reference_mapping():
    d, heads, ff = 192, 3, 768
    w = _timm_weights(d, ff)
    x = torch.randn(2, 128, d, generator=torch.Generator().manual_seed(1))     # [2,128,D] as in the manual script (:245)
    ref = vit_ref.timm_block(x, w, heads)
    p = {"encoder/layer_0/" + k: v for k, v in vit_ref.timm_to_chambers_block(w, heads).items()}
    cfg = {"dropout_rate": 0.0, "n_heads": heads, "norm_epsilon": 1e-6}
    out = vit_ref.encoder_layer(x, p, "encoder/layer_0/", cfg, {}, 0, False)
    assert torch.allclose(out, ref, atol=1e-4)                                  # whole block: :274-276
    h_ref = torch.nn.functional.layer_norm(x, (d,), w["norm1.weight"], w["norm1.bias"], 1e-6)
    h = vit_ref.layer_norm(x, p["encoder/layer_0/norm1/gamma"], p["encoder/layer_0/norm1/beta"], 1e-6)
    assert torch.allclose(h, h_ref, atol=1e-5)                                  # norm1: :258-260


def test_full_model_shapes_grads_and_dropout_determinism():
    import sys
    from chambers_amd.engine import ViTConfig, init_keras_weights
    cfg = ViTConfig(16, 128, 2, 2, 256, image_size=(32, 48), classes=7)
    kw = init_keras_weights(cfg, seed=3)
    p = {k: torch.tensor(v, requires_grad=True) for k, v in kw.items()}
    x = torch.randn(3, 32, 48, 3, generator=torch.Generator().manual_seed(0))
    keys = {s: 1000 + s for s in range(7)}
    a = vit_ref.vit_forward(p, x, cfg.as_oracle_cfg(), keys=keys)
    b = vit_ref.vit_forward(p, x, cfg.as_oracle_cfg(), keys=keys)
    c = vit_ref.vit_forward(p, x, cfg.as_oracle_cfg(), keys=None)
    assert a.shape == (3, 7) and torch.equal(a, b) and not torch.allclose(a, c)
    vit_ref.sparse_ce_from_logits(a, torch.tensor([0, 3, 6])).backward()
    assert all(v.grad is not None and torch.isfinite(v.grad).all() for v in p.values())
    # d(b_key) vanishes (softmax shift invariance) — the property the GPU parity tests rely on
    gk = p["encoder/layer_0/multi_head_attention/b_key"].grad.abs().max()
    gw = p["encoder/layer_0/multi_head_attention/w_key"].grad.abs().max()
    assert gk < 1e-5 * max(1.0, float(gw))


def test_adamw_restatement_known_answer():
    p = {"w": torch.tensor([1.0, -2.0])}
    g = {"w": torch.tensor([0.5, 0.25])}
    m, v = {"w": torch.zeros(2)}, {"w": torch.zeros(2)}
    vit_ref.adamw_step(p, g, m, v, 1, lr=1e-3, weight_decay=0.01)
    # decay first: w = w - 0.01 w = [0.99, -1.98]; m = 0.1 g; v = 0.001 g^2; lr_t = 1e-3*sqrt(0.001)/0.1
    # update = lr_t * m / (sqrt(v) + 1e-7) ~= 1e-3 * sign(g)
    np.testing.assert_allclose(p["w"].numpy(), [0.99 - 1e-3, -1.98 - 1e-3], rtol=0, atol=2e-6)
    np.testing.assert_allclose(m["w"].numpy(), [0.05, 0.025], rtol=1e-6)
    np.testing.assert_allclose(v["w"].numpy(), [0.00025, 0.0000625], rtol=2e-5)   # (1 - 0.999f) in float32, as keras computes it


def test_timm_state_dict_import_matches_timm_forward():
    """chambers_amd.utils.weights.timm_state_dict_to_keras: a random timm-layout ViT evaluated the timm way (F.linear on the
    original tensors, fused qkv, conv patch embedding) equals the oracle's chambers graph on the converted weights — the
    equivalence the reference asserts in test_units/manual_test_vit_weights.py:252-341, here for the whole model."""
    from chambers_amd.utils.weights import timm_state_dict_to_keras
    g = torch.Generator().manual_seed(3)
    d, heads, ff, p, L, classes = 64, 2, 96, 8, 2, 5
    hgt, wid = 24, 16
    n = (hgt // p) * (wid // p) + 1
    r = lambda *s, sc=0.1: torch.randn(*s, generator=g) * sc   # noqa: E731
    sd = {"patch_embed.proj.weight": r(d, 3, p, p), "patch_embed.proj.bias": r(d), "cls_token": r(1, 1, d), "pos_embed": r(1, n, d),
          "norm.weight": 1 + r(d), "norm.bias": r(d), "head.weight": r(classes, d), "head.bias": r(classes)}
    for i in range(L):
        t = "blocks.%d." % i
        sd.update({t + "norm1.weight": 1 + r(d), t + "norm1.bias": r(d), t + "attn.qkv.weight": r(3 * d, d), t + "attn.qkv.bias": r(3 * d),
                   t + "attn.proj.weight": r(d, d), t + "attn.proj.bias": r(d), t + "norm2.weight": 1 + r(d), t + "norm2.bias": r(d),
                   t + "mlp.fc1.weight": r(ff, d), t + "mlp.fc1.bias": r(ff), t + "mlp.fc2.weight": r(d, ff), t + "mlp.fc2.bias": r(d)})
    x = torch.randn(2, hgt, wid, 3, generator=g)                                  # NHWC, already normalised
    # timm forward: Conv2d(stride = kernel = p) on NCHW, flatten, cls token, pos embed, blocks, norm, cls pooling, head
    t = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=p)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat([sd["cls_token"].expand(2, -1, -1), t], dim=1) + sd["pos_embed"]
    for i in range(L):
        pre = "blocks.%d." % i
        t = vit_ref.timm_block(t, {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, heads)
    t = torch.nn.functional.layer_norm(t, (d,), sd["norm.weight"], sd["norm.bias"], 1e-6)
    ref = torch.nn.functional.linear(t[:, 0], sd["head.weight"], sd["head.bias"])
    kw = timm_state_dict_to_keras(sd, heads)
    cfg = {"patch_size": p, "n_encoder_layers": L, "n_heads": heads, "dropout_rate": 0.0}
    out = vit_ref.vit_forward({k: torch.tensor(v) for k, v in kw.items()}, x, cfg)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5), float((out - ref).abs().max())
    assert kw["encoder/layer_1/multi_head_attention/w_key"].shape == (d, heads, d // heads)
    assert kw["patch_embeddings/embedding/kernel"].shape == (p, p, 3, d)
