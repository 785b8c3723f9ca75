"""GPU tier: data-parallel equivalence of the engine.  Two ranks (gloo process group, both on cuda:0 — RCCL refuses two ranks on
one device, and the box has one) each run forward + loss + backward on half of a batch; after the bucketed exchange
(GradBucketReducer: queued per bucket, flushed beside the attention backward) gradient x grad_scale must equal the gradient of a
single process on the whole batch, and one AdamW step must leave identical parameters on both ranks."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(patch_size=16, patch_dim=128, n_encoder_layers=3, n_heads=2, ff_dim=256, dropout_rate=0.0, image_size=(64, 64), classes=10)
BATCH = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _data():
    g = np.random.Generator(np.random.PCG64(5))
    return g.integers(0, 256, size=(BATCH, 64, 64, 3), dtype=np.uint8), g.integers(0, 10, size=(BATCH,))


def _grad(eng, images, labels):
    eng.forward(torch.as_tensor(images, device="cuda"), training=True)
    loss = eng.loss(torch.as_tensor(labels, device="cuda"))
    eng.backward()
    eng.reducer.finish()
    torch.cuda.synchronize()
    return loss.float().cpu().numpy(), (eng.G * eng.reducer.grad_scale).cpu().numpy()


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights
    cfg = ViTConfig(**CFG)
    per = BATCH // world
    eng = ViTEngine(cfg, per, training=True, seed=0)
    eng.load_keras_weights(init_keras_weights(cfg, seed=7))
    assert eng.reducer.active and eng.reducer.world == world
    images, labels = _data()
    sl = slice(rank * per, (rank + 1) * per)
    loss, grad = _grad(eng, images[sl], labels[sl])
    assert eng.reducer.handles == [] and eng.reducer.queued == []
    eng.adamw_step(learning_rate=1e-3, weight_decay=0.05)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), loss=loss, grad=grad, params=eng.P.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_one_process_on_the_whole_batch(tmp_path):
    import torch.multiprocessing as mp
    from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in (0, 1))
    np.testing.assert_array_equal(r0["grad"], r1["grad"])          # both ranks hold the same reduced gradient ...
    np.testing.assert_array_equal(r0["params"], r1["params"])      # ... and the same parameters after the step
    cfg = ViTConfig(**CFG)
    eng = ViTEngine(cfg, BATCH, training=True, seed=0)
    eng.load_keras_weights(init_keras_weights(cfg, seed=7))
    assert not eng.reducer.active
    images, labels = _data()
    loss, grad = _grad(eng, images, labels)
    np.testing.assert_allclose(np.concatenate([r0["loss"], r1["loss"]]), loss, rtol=2e-3, atol=2e-3)
    # mean-loss gradient of the whole batch = average of the two half-batch gradients (bf16 operands, fp32 sums in another order)
    scale = np.abs(grad).max()
    assert scale > 0
    np.testing.assert_allclose(r0["grad"], grad, rtol=0, atol=2e-2 * scale)
    assert np.abs(r0["grad"] - grad).mean() < 2e-3 * scale
