"""GPU tier: the data-parallel train step with 2 ranks (both on cuda:0, gloo transport — RCCL refuses two ranks on one
device; the 8-GPU RCCL run is the driver's).  Checks the engine-level contract: ranks train on different shards, exchange
ONE bucketed gradient all-reduce per step, and end with bit-identical parameters that equal a single-process step on the
averaged gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _cfg():
    from chambers_amd.engine import ViTConfig
    return ViTConfig(16, 128, 2, 2, 256, dropout_rate=0.1, image_size=(32, 32), classes=10)


def _data(rank, bsz):
    g = np.random.Generator(np.random.PCG64(100 + rank))
    return g.integers(0, 256, size=(bsz, 32, 32, 3), dtype=np.uint8), g.integers(0, 10, size=(bsz,))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chambers_amd.engine import ViTEngine, init_keras_weights
    cfg = _cfg()
    eng = ViTEngine(cfg, 4, training=True, seed=11)
    eng.load_keras_weights(init_keras_weights(cfg, seed=1234))
    assert eng.reducer.active and eng.reducer.world == world
    img, lab = _data(rank, 4)
    eng.forward(torch.as_tensor(img, device="cuda"), training=True)
    eng.loss(torch.as_tensor(lab, device="cuda"))
    eng.backward()
    eng.reducer.finish()
    np.save(os.path.join(out_dir, "g%d.npy" % rank), eng.G.cpu().numpy())          # already summed over ranks
    eng.adamw_step(learning_rate=1e-3, weight_decay=0.01)
    np.save(os.path.join(out_dir, "p%d.npy" % rank), eng.P.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_step_keeps_replicas_identical(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_array_equal(p0, p1)
    # single process: local gradients of both shards, summed on the host, must match the exchanged buffer
    from chambers_amd.engine import ViTEngine, init_keras_weights
    cfg = _cfg()
    tot = None
    for rank in range(world):
        eng = ViTEngine(cfg, 4, training=True, seed=11)
        eng.load_keras_weights(init_keras_weights(cfg, seed=1234))
        img, lab = _data(rank, 4)
        eng.forward(torch.as_tensor(img, device="cuda"), training=True)
        eng.loss(torch.as_tensor(lab, device="cuda"))
        eng.backward()
        g = eng.G.cpu().numpy().astype(np.float64)
        tot = g if tot is None else tot + g
    rel = np.linalg.norm(g0 - tot) / np.linalg.norm(tot)
    assert rel < 1e-5, rel            # fp32 atomics in the weight-gradient kernels: order-dependent last bits only
