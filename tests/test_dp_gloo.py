"""CPU tier, world_size 2 over gloo: the data-parallel gradient exchange of the engine (GradBucketReducer) — bucketed,
asynchronous all-reduce of contiguous slices of the flat gradient buffer, averaging folded into grad_scale — and the
sharding arithmetic bench.py relies on (weak scaling: every rank holds its own batch, parameters replicated)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chambers_amd.engine import GradBucketReducer, ViTConfig, build_param_table
    cfg = ViTConfig(16, 128, 2, 2, 256, image_size=(32, 32), classes=10)
    specs, total, buckets = build_param_table(cfg)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(total, generator=g)
    local = flat.clone()
    red = GradBucketReducer(flat, buckets)
    assert red.active and red.world == world and abs(red.grad_scale - 1.0 / world) < 1e-12
    # backward marks buckets final in order 0..L+1; collectives are asynchronous until finish()
    # bucket_ready() queues, flush() issues (the engine flushes beside the attention backward and at the end of backward)
    for k in range(len(buckets)):
        red.bucket_ready(k)
        if k % 2:
            red.flush()       # adjacent queued buckets go out as one collective
            assert red.queued == [] and len(red.handles) == (k + 1) // 2
    red.flush()
    assert len(red.handles) == (len(buckets) + 1) // 2
    red.finish()
    assert red.handles == []
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered)
    assert torch.allclose(flat, expect, atol=1e-6)
    # parameters stay bit-identical across ranks when every rank applies the same averaged gradient
    p = torch.ones(total)
    p -= 0.1 * flat * red.grad_scale
    ps = [torch.empty_like(p) for _ in range(world)]
    dist.all_gather(ps, p)
    assert torch.equal(ps[0], ps[1])
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    np.testing.assert_array_equal(a, b)


def test_reducer_is_a_noop_single_process():
    from chambers_amd.engine import GradBucketReducer
    flat = torch.arange(10.0)
    red = GradBucketReducer(flat, [(0, 4), (4, 10)])
    assert not red.active and red.grad_scale == 1.0
    red.bucket_ready(0)
    red.bucket_ready(1)
    red.finish()
    assert torch.equal(flat, torch.arange(10.0))


def test_buckets_cover_the_flat_buffer_in_backward_order():
    from chambers_amd.engine import ViTConfig, build_param_table
    cfg = ViTConfig(16, 768, 12, 12, 3072)
    specs, total, buckets = build_param_table(cfg)
    assert buckets[0][0] == 0 and buckets[-1][1] == total
    names = [s.name for s in specs]
    assert names[0].startswith("predictions") and names[-1] == "patch_embeddings/embedding/bias"
    first_block = [n for n in names if n.startswith("encoder/layer_")][0]
    assert first_block.startswith("encoder/layer_11/")          # last block's gradients are final first
    sizes_mb = [(hi - lo) * 4 / 2 ** 20 for lo, hi in buckets]
    # one ViT-B block = 7.09 M params = 27 MiB fp32, in two buckets split where its attention backward starts
    assert 19 < sizes_mb[1] < 22 and 6 < sizes_mb[2] < 8 and 25 < sizes_mb[1] + sizes_mb[2] < 30
    assert all(a[1] == b[0] for a, b in zip(buckets, buckets[1:]))     # contiguous: adjacent buckets merge into one collective


def _worker_bf16(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chambers_amd.engine import GradBucketReducer
    g = torch.Generator().manual_seed(7 + rank)
    local = torch.randn(5000, generator=g)
    buckets = [(0, 1024), (1024, 3072), (3072, 5000)]
    outs = {}
    for payload in ("fp32", "bf16"):
        flat = local.clone()
        red = GradBucketReducer(flat, buckets, payload=payload)
        assert red.active and red.payload == payload
        for k in range(len(buckets)):
            red.bucket_ready(k)
            if k == 0:
                red.flush()
        red.finish()
        assert red.staged == [] and red.handles == []
        assert red.bytes_reduced == 5000 * (4 if payload == "fp32" else 2) and red.n_collectives == 2
        outs[payload] = flat
    both = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    # the bf16 exchange is exactly: every rank's contribution rounded to bf16 once, summed, the sum rounded to bf16 once
    expect = (both[0].to(torch.bfloat16) + both[1].to(torch.bfloat16)).to(torch.float32)
    assert torch.equal(outs["bf16"], expect)
    assert torch.allclose(outs["fp32"], both[0] + both[1], atol=1e-6)
    # and so differs from the fp32 exchange by bf16 roundings only: |diff| <= 2^-8 (|a| + |b|) + 2^-8 |a + b| (three half-ulp bounds)
    bound = 2.0 ** -8 * (both[0].abs() + both[1].abs() + (both[0] + both[1]).abs()) + 1e-30
    assert bool(((outs["bf16"] - outs["fp32"]).abs() <= bound).all())
    rel = float((outs["bf16"] - outs["fp32"]).norm() / outs["fp32"].norm())
    assert 1e-4 < rel < 6e-3
    np.save(os.path.join(out_dir, "bf16_rank%d.npy" % rank), outs["bf16"].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_payload_world2(tmp_path):
    """the optional bf16 payload halves the bytes and differs from the fp32 exchange by one rounding of each
    contribution and one of the sum; both ranks end with identical values."""
    world = 2
    mp.spawn(_worker_bf16, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    np.testing.assert_array_equal(np.load(tmp_path / "bf16_rank0.npy"), np.load(tmp_path / "bf16_rank1.npy"))


def _worker_forced(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chambers_amd.engine import GradBucketReducer
    flat = torch.arange(16.0)
    assert not GradBucketReducer(flat, [(0, 16)]).active                  # a one-rank group is a no-op unless forced
    for payload in ("fp32", "bf16"):
        red = GradBucketReducer(flat, [(0, 6), (6, 16)], force=True, payload=payload)
        assert red.active and red.world == 1 and red.grad_scale == 1.0
        red.bucket_ready(0)
        red.flush()
        red.bucket_ready(1)
        red.finish()
        assert red.n_collectives == 2 and torch.equal(flat, torch.arange(16.0))      # small integers are exact in bf16
    dist.destroy_process_group()


def test_forced_single_rank_group_issues_real_collectives():
    mp.spawn(_worker_forced, args=(1, _free_port()), nprocs=1, join=True)
