"""Full-size BASELINE configs 4 and 5 at their per-GPU shape, full depth, on the GPU - through properties that
need no CPU oracle at that size, as tests/test_full_step_gpu.py does for config 3:
  * config 4: ViT-L/16 224^2, batch 512 (24 blocks, D = 1024, 16 heads);
  * config 5: ViT-B/16 384^2 (577 tokens: streaming attention forward, two-pass backward, M = 73,856 = 288.5 row tiles launched over
    padded tiles), batch 128, AutoAugment policy v0;
  gradient linearity over the batch (mean-loss gradient of the batch = mean of its halves' gradients, per-sample losses identical)
  and the loss falling over a few full train steps with dropout 0.1 and the fused augmentation stage."""
import numpy as np
import pytest
import torch

from conftest import fp_check

pytestmark = pytest.mark.gpu

VITL16 = dict(patch_size=16, patch_dim=1024, n_encoder_layers=24, n_heads=16, ff_dim=4096, image_size=(224, 224), classes=1000)
VITB16_384 = dict(patch_size=16, patch_dim=768, n_encoder_layers=12, n_heads=12, ff_dim=3072, image_size=(384, 384), classes=1000)


def rel_l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-300))


def _linearity(cfg_kw, batch, tag, bound):
    from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights
    cfg = ViTConfig(**dict(cfg_kw, dropout_rate=0.0))
    kw = init_keras_weights(cfg, seed=21)
    g = np.random.Generator(np.random.PCG64(5))
    h, w = cfg.image_size
    images = torch.as_tensor(g.integers(0, 256, size=(batch, h, w, 3), dtype=np.uint8), device="cuda")
    labels = torch.as_tensor(g.integers(0, 1000, size=(batch,)).astype(np.int32), device="cuda")

    def grads(bsz, slices):
        eng = ViTEngine(cfg, bsz, training=True, seed=0)
        eng.load_keras_weights(kw)
        out = []
        for s in slices:
            eng.forward(images[s], training=True)
            loss = eng.loss(labels[s]).clone()
            eng.backward()
            eng.reducer.finish()
            out.append((loss, eng.G.clone()))
            eng.G.zero_()
            eng._g_clean = True
        m, mg = eng.M, eng.Mg
        del eng
        torch.cuda.empty_cache()
        return out, m, mg

    ((loss_full, g_full),), m_full, mg_full = grads(batch, [slice(0, batch)])
    half = batch // 2
    ((loss_a, g_a), (loss_b, g_b)), _, _ = grads(half, [slice(0, half), slice(half, batch)])
    assert torch.isfinite(g_full).all() and float(g_full.abs().max()) > 0
    fp_check("%s | per-sample loss, batch %d vs 2 x %d" % (tag, batch, half), rel_l2(loss_full, torch.cat([loss_a, loss_b])), 1e-7)
    fp_check("%s | gradient linearity over the batch" % tag, rel_l2(g_full, 0.5 * (g_a + g_b)), bound)
    return m_full, mg_full


def test_config4_vit_l16_gradient_is_linear_over_the_batch():
    _linearity(VITL16, 512, "config4 ViT-L/16 b512", 1.5e-6)


def test_config5_vit_b16_384_gradient_is_linear_over_the_batch():
    m, mg = _linearity(VITB16_384, 128, "config5 ViT-B/16 384 b128", 1.5e-6)
    assert m == 128 * 577 and m % 256 != 0 and mg == (m + 255) // 256 * 256      # the ragged token count runs over padded full tiles


@pytest.mark.parametrize("name,cfg_kw,batch,scheme", [("config4", VITL16, 512, "randaugment"), ("config5", VITB16_384, 128, "autoaugment")])
def test_full_size_train_steps_reduce_the_loss(name, cfg_kw, batch, scheme):
    from chambers_amd import augmentations as aug
    from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights
    cfg = ViTConfig(**dict(cfg_kw, dropout_rate=0.1))
    eng = ViTEngine(cfg, batch, training=True, seed=1)
    eng.load_keras_weights(init_keras_weights(cfg, seed=22))
    g = np.random.Generator(np.random.PCG64(6))
    h, w = cfg.image_size
    images = torch.as_tensor(g.integers(0, 256, size=(batch, h, w, 3), dtype=np.uint8), device="cuda")
    labels = torch.as_tensor(g.integers(0, 1000, size=(batch,)).astype(np.int32), device="cuda")
    if scheme == "randaugment":
        plan = aug.RandAugment(2, 9).plan(images.shape, [{"op": 12, "negate": False}, {"op": 7, "negate": True}])       # Solarize -> ShearX
    else:
        plan = aug.AutoAugment().plan(images.shape, {"policy": 2, "apply": (True, True), "negate": (False, True)})       # Color -> Rotate
    losses = []
    for _ in range(5):
        losses.append(float(eng.train_step(images, labels, augment=plan, learning_rate=2e-4, weight_decay=0.05).mean().item()))
    assert all(np.isfinite(losses)), losses
    assert abs(losses[0] - np.log(1000.0)) < 0.6, losses
    assert losses[-1] < losses[0] - 0.03, losses
    if eng.Mg != eng.M:          # config 5: the pad rows of the gradient-side token matrices are still exactly zero after five steps
        for nm in ("dz", "da1", "dh", "do", "dqkv", "dx"):
            assert not bool(getattr(eng, nm)[eng.M:].any()), nm
