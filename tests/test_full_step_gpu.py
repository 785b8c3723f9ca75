"""The full BASELINE config-3 step on the GPU (ViT-B/16, 224x224, batch 512 - what bench.py times), checked through properties
that need no CPU oracle at that size (size-independent properties at full sizes):
  * linearity over the batch: the gradient of the mean loss over 512 images = the mean of the gradients over its two halves
    (each half run through a batch-256 engine with the same weights; dropout off, so a sample's arithmetic does not depend on its
    position in the batch), per-sample losses identical;
  * the whole train step (fused RandAugment -> forward -> CE -> backward -> AdamW, dropout 0.1) drives the loss down on a fixed batch."""
import numpy as np
import pytest
import torch

from conftest import fp_check

pytestmark = pytest.mark.gpu

VITB16 = dict(patch_size=16, patch_dim=768, n_encoder_layers=12, n_heads=12, ff_dim=3072, image_size=(224, 224), classes=1000)


def rel_l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-300))


def test_config3_gradient_is_linear_over_the_batch():
    from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights
    cfg = ViTConfig(**dict(VITB16, dropout_rate=0.0))
    kw = init_keras_weights(cfg, seed=11)
    g = np.random.Generator(np.random.PCG64(3))
    images = torch.as_tensor(g.integers(0, 256, size=(512, 224, 224, 3), dtype=np.uint8), device="cuda")
    labels = torch.as_tensor(g.integers(0, 1000, size=(512,)).astype(np.int32), device="cuda")

    def grads(batch, sl, overlap=False):
        eng = ViTEngine(cfg, batch, training=True, seed=0, overlap_wgrad=overlap)
        eng.load_keras_weights(kw)
        out = []
        for s in sl:
            eng.forward(images[s], training=True)
            loss = eng.loss(labels[s]).clone()
            eng.backward()
            eng.reducer.finish()
            out.append((loss, eng.G.clone()))
            eng.G.zero_()
            eng._g_clean = True
        del eng
        torch.cuda.empty_cache()
        return out

    (loss_full, g_full), = grads(512, [slice(0, 512)])
    (loss_a, g_a), (loss_b, g_b) = grads(256, [slice(0, 256), slice(256, 512)])
    assert torch.isfinite(g_full).all() and float(g_full.abs().max()) > 0
    # a sample's forward does not depend on the batch it is in: per-sample losses agree to fp32 rounding of the loss kernel
    fp_check("config3 full step | per-sample loss, batch 512 vs 2 x 256", rel_l2(loss_full, torch.cat([loss_a, loss_b])), 1e-7)     # measured 0.0
    # d(mean over 512) = (d(mean over first 256) + d(mean over second 256)) / 2; the weight-gradient GEMMs split the token axis
    # differently for the two batch sizes, so the fp32 sums differ in order only (measured 4.65e-7; bound = x 1.5)
    fp_check("config3 full step | gradient linearity over the batch", rel_l2(g_full, 0.5 * (g_a + g_b)), 7e-7)
    # the same backward with the weight-gradient GEMMs on the side stream (persistent kernels, operand rings re-used over 12 blocks)
    (loss_ov, g_ov), = grads(512, [slice(0, 512)], overlap=True)
    assert torch.equal(loss_ov, loss_full)
    fp_check("config3 full step | gradient, side-stream weight gradients vs one stream", rel_l2(g_ov, g_full), 3e-7)      # measured 1.9e-7: the order of the bias-gradient atomics


def test_config3_train_steps_reduce_the_loss():
    from chambers_amd import augmentations as aug
    from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights
    cfg = ViTConfig(**dict(VITB16, dropout_rate=0.1))
    eng = ViTEngine(cfg, 512, training=True, seed=1)
    eng.load_keras_weights(init_keras_weights(cfg, seed=12))
    g = np.random.Generator(np.random.PCG64(4))
    images = torch.as_tensor(g.integers(0, 256, size=(512, 224, 224, 3), dtype=np.uint8), device="cuda")
    labels = torch.as_tensor(g.integers(0, 1000, size=(512,)).astype(np.int32), device="cuda")
    layer = aug.RandAugment(2, 9)
    dec = [{"op": 12, "negate": False}, {"op": 7, "negate": True}]        # Solarize -> ShearX, the same chain every step
    plan = layer.plan(images.shape, dec)
    losses = []
    for _ in range(6):
        losses.append(float(eng.train_step(images, labels, augment=plan, learning_rate=3e-4, weight_decay=0.05).mean().item()))
    assert all(np.isfinite(losses)), losses
    assert abs(losses[0] - np.log(1000.0)) < 0.5, losses          # a fresh classifier starts near ln(classes)
    assert losses[-1] < losses[0] - 0.05, losses                  # six steps on one batch must reduce its loss
