"""GPU tier: ragged-batch resize (`chb_resize_ragged`) and the DeviceBatcher hand-over against the oracle's per-image
tf.image.resize restatement (oracle/augment_ref.py: resize), bit-exact; mnist sample of the reference's tests as input files."""
import os

import numpy as np
import pytest
import torch

from oracle import augment_ref as R

pytestmark = pytest.mark.gpu

MNIST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_data", "mnist", "train")


def _ragged(rng, sizes):
    return [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]


def _run(images, oh, ow, method, out_dtype):
    from chambers_amd import kernels as K
    from chambers_amd.data.device import DeviceBatcher
    total, offs, hw = DeviceBatcher.pack(images)
    packed = np.concatenate([im.reshape(-1) for im in images])
    assert packed.size == total
    out = K.resize_ragged(torch.as_tensor(packed, device="cuda"), torch.as_tensor(offs, device="cuda"), torch.as_tensor(hw, device="cuda"),
                          oh, ow, method, out_dtype)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("method", ["bilinear", "nearest"])
@pytest.mark.parametrize("out", [(224, 224), (64, 96), (17, 4)])
def test_ragged_resize_matches_the_oracle_per_image(method, out):
    rng = np.random.default_rng(11)
    sizes = [(1, 1), (28, 28), (333, 500), (500, 375), (64, 96), (2, 1023), (719, 3), (224, 224)]
    images = _ragged(rng, sizes)
    got32 = _run(images, out[0], out[1], method, torch.float32)
    got8 = _run(images, out[0], out[1], method, torch.uint8)
    assert got32.shape == (len(sizes), out[0], out[1], 3) and got8.dtype == np.uint8
    for k, im in enumerate(images):
        ref = R.resize(im[None], out[0], out[1], method)[0]
        np.testing.assert_array_equal(got32[k], ref.astype(np.float32))
        np.testing.assert_array_equal(got8[k], ref.astype(np.float32).astype(np.uint8))      # tf.cast(float -> uint8): truncation


def test_ragged_equals_the_uniform_resize():
    from chambers_amd import kernels as K
    rng = np.random.default_rng(2)
    x = rng.integers(0, 256, size=(6, 120, 200, 3), dtype=np.uint8)
    got = _run(list(x), 224, 224, "bilinear", torch.float32)
    ref = K.resize(torch.as_tensor(x, device="cuda"), 224, 224, "bilinear").cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_argument_errors():
    from chambers_amd import kernels as K
    p = torch.zeros(64, dtype=torch.uint8, device="cuda")
    o = torch.zeros(1, dtype=torch.int64, device="cuda")
    hw = torch.tensor([[4, 4]], dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        K.resize_ragged(p, o, hw, 8, 6)                      # OW % 4
    with pytest.raises(ValueError):
        K.resize_ragged(p, o, hw, 8, 8, "bicubic")
    with pytest.raises(ValueError):
        K.resize_ragged(p, o, hw.to(torch.int64), 8, 8)
    assert K.resize_ragged(p[:0], o[:0], hw[:0], 8, 8).shape == (0, 8, 8, 3)


def test_device_batcher_over_the_class_dataset():
    from chambers_amd.data import InterleaveImageClassDataset, match_nested_set, read_and_decode_image, match_img_files
    from chambers_amd.data.device import DeviceBatcher
    dirs = sorted(match_nested_set(MNIST))
    td = InterleaveImageClassDataset(class_dirs=dirs, labels=list(range(10)), class_cycle_length=5, images_per_block=2)
    batches = list(DeviceBatcher(td, batch_size=8, size=(32, 32), out_dtype=torch.uint8, depth=2))
    assert [int(b[0].shape[0]) for b in batches] == [8, 8, 4]
    labels = torch.cat([b[1] for b in batches]).cpu().tolist()
    assert labels == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9]
    first = read_and_decode_image(match_img_files(dirs[0])[0], channels=3)
    ref = R.resize(first[None], 32, 32, "bilinear")[0].astype(np.float32).astype(np.uint8)
    np.testing.assert_array_equal(batches[0][0][0].cpu().numpy(), ref)
    # slot reuse (depth 2, three batches) must not disturb earlier outputs
    again = list(DeviceBatcher(td, batch_size=8, size=(32, 32), out_dtype=torch.uint8, depth=1))
    for a, b in zip(batches, again):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert len(list(DeviceBatcher(td, batch_size=8, size=(32, 32), drop_remainder=True))) == 2
