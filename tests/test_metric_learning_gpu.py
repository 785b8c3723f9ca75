"""Metric-learning head: L2Normalization, MultiSimilarityLoss + MultiSimilarityMiner against the oracle
(values and gradients), and one image-retrieval style training loop through the ViT engine (feature head -> L2 -> MS loss)."""
import numpy as np
import pytest
import torch

from oracle import metric_ref

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_l2_normalization_forward_backward():
    from chambers_amd.layers.normalization import L2Normalization
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 48, generator=g)
    x[5] = 0.0                                   # hits the 1e-12 floor
    layer = L2Normalization(axis=-1)
    y = layer(x.cuda())
    xr = x.clone().requires_grad_(True)
    yr = metric_ref.l2_normalize(xr)
    assert torch.allclose(y.cpu(), yr.detach(), rtol=1e-6, atol=1e-7)
    dy = torch.randn(37, 48, generator=g)
    yr.backward(dy)
    dx = layer.backward(dy.cuda())
    assert torch.allclose(dx.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    assert layer.get_config()["axis"] == -1


@pytest.mark.parametrize("miner,ignore_diag,ignore_neg", [(0.1, True, True), (None, True, True), (0.1, False, True), (0.05, True, False)])
def test_multi_similarity_loss_matches_oracle(miner, ignore_diag, ignore_neg):
    from chambers_amd.losses import MultiSimilarityLoss
    from chambers_amd.miners import MultiSimilarityMiner
    g = torch.Generator().manual_seed(1)
    n, d = 96, 40
    emb = metric_ref.l2_normalize(torch.randn(n, d, generator=g))
    labels = torch.randint(0, 12, (n,), generator=g)
    labels[3], labels[40] = -1, -1              # ignored anchors
    labels[7] = 99                              # a class of one: no positives
    loss_fn = MultiSimilarityLoss(ignore_diag=ignore_diag, ignore_negative_labels=ignore_neg,
                                  miner=None if miner is None else MultiSimilarityMiner(miner))
    er = emb.clone().requires_grad_(True)
    rows_ref = metric_ref.multi_similarity_loss(labels, er, ignore_diag=ignore_diag, ignore_negative_labels=ignore_neg, miner_margin=miner)
    rows_ref.mean().backward()
    rows = loss_fn.per_sample(labels, emb.cuda())
    assert torch.allclose(rows.cpu(), rows_ref.detach(), rtol=2e-4, atol=1e-5), float((rows.cpu() - rows_ref.detach()).abs().max())
    value, grad = loss_fn.value_and_gradient(labels.cuda(), emb.cuda())
    assert abs(float(value) - float(rows_ref.detach().mean())) < 1e-5 * max(1.0, abs(float(rows_ref.detach().mean())))
    assert rel_l2(grad, er.grad) < 1e-4, rel_l2(grad, er.grad)
    assert abs(float(loss_fn(labels, emb.cuda())) - float(value)) < 1e-6
    cfg = loss_fn.get_config()
    assert cfg["pos_scale"] == 2.0 and cfg["neg_scale"] == 40.0 and cfg["threshold"] == 0.5 and (cfg["miner"] is None) == (miner is None)


def test_retrieval_training_loop_through_the_engine():
    """uint8 images -> ViT (tanh feature head, no classification top) -> L2Normalization -> MultiSimilarityLoss -> backward -> AdamW:
    the loss falls on a small memorisable set."""
    from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights
    from chambers_amd.layers.normalization import L2Normalization
    from chambers_amd.losses import MultiSimilarityLoss
    cfg = ViTConfig(16, 128, 2, 2, 256, dropout_rate=0.0, image_size=(32, 32), classes=10, include_top=False, feature_dim=64)
    eng = ViTEngine(cfg, 16, training=True, seed=1)
    eng.load_keras_weights(init_keras_weights(cfg, seed=5))
    g = np.random.Generator(np.random.PCG64(0))
    base = g.integers(64, 192, size=(1, 32, 32, 3))                       # classes differ from a shared image by a faint pattern,
    protos = base + g.integers(-12, 13, size=(4, 32, 32, 3))              # buried under stronger per-sample noise: at initialisation
    labels = np.repeat(np.arange(4), 4)                                   # the embeddings do not separate them (the miner keeps pairs)
    images = np.clip(protos[labels] + g.integers(-40, 41, size=(16, 32, 32, 3)), 0, 255).astype(np.uint8)
    x, y = torch.as_tensor(images, device="cuda"), torch.as_tensor(labels, device="cuda")
    l2, loss_fn = L2Normalization(axis=-1), MultiSimilarityLoss()
    history = []
    for _ in range(60):
        feats = eng.forward(x, training=True)
        value, grad = loss_fn.value_and_gradient(y, l2(feats))
        eng.backward(l2.backward(grad))
        eng.adamw_step(learning_rate=1e-3)
        history.append(float(value))
    assert np.isfinite(history).all() and history[0] > 0.1 and min(history[-5:]) < 0.5 * history[0], history[::10]
