"""Input side of the augmentation path: Resizing / ResizingMinMax / CenterCrop / RandomCrop / RandomFlip /
Rescaling.  CPU tier: the oracle against hand-computed values and the reference's own ResizingMinMax shape tests
(test_units/augmentations/test_image_augmentations.py:66-80).  GPU tier: bit-exact parity of the HIP kernels with the oracle."""
import numpy as np
import pytest
import torch

from oracle import augment_ref as A

IMG = np.array([[139, 186, 208, 200], [175, 201, 198, 200], [166, 191, 193, 195], [124, 155, 172, 151]], dtype=np.uint8)
IMG = np.stack([IMG, IMG, IMG], axis=-1)[None]
IMG_NOT_SQUARE = IMG[:, :, :3, :]                                   # test_image_augmentations.py:5-17
REF_SHAPES = [(dict(min_side=100), (1, 133, 100, 3)), (dict(max_side=100), (1, 100, 75, 3)),
              (dict(min_side=100, max_side=100), (1, 100, 75, 3)), (dict(min_side=100, max_side=50), (1, 50, 37, 3))]   # :66-80


# ------------------------------------------------------------------------------------ CPU tier
@pytest.mark.parametrize("kw,shape", REF_SHAPES)
def test_resizing_minmax_reference_shapes_oracle_and_host(kw, shape):
    from chambers_amd.augmentations import ResizingMinMax
    h, w = A.resizing_minmax_size(4, 3, **kw)
    assert (1, h, w, 3) == shape
    assert ResizingMinMax(**kw).target_size(4, 3) == (h, w)
    assert A.resize(IMG_NOT_SQUARE, h, w).shape == shape
    with pytest.raises(ValueError):
        ResizingMinMax()


def test_resize_oracle_known_values():
    x = np.array([[[[0.0], [10.0]], [[20.0], [30.0]]]], dtype=np.float32)          # 2x2 ramp
    up = A.resize(x, 4, 4)[0, :, :, 0]
    # half-pixel centres: sample positions -0.25, 0.25, 0.75, 1.25 -> clamped at the ends, quarter steps inside
    np.testing.assert_allclose(up[0], [0.0, 2.5, 7.5, 10.0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(up[:, 0], [0.0, 5.0, 15.0, 20.0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(up[1, 1], 0.75 * (0.75 * 0 + 0.25 * 10) + 0.25 * (0.75 * 20 + 0.25 * 30), atol=1e-6)
    assert A.resize(x, 2, 2).dtype == np.float32 and np.array_equal(A.resize(x, 2, 2), x)       # identity size
    down = A.resize(x, 1, 1)
    assert float(down.ravel()[0]) == 15.0                                                     # centre of the ramp
    near = A.resize(IMG, 2, 2, "nearest")
    assert near.dtype == np.uint8 and np.array_equal(near[0, :, :, 0], IMG[0, 1::2, 1::2, 0])  # floor((o+.5)*2) = 1, 3
    u8 = A.resize(IMG, 8, 6)
    assert u8.dtype == np.float32 and float(u8.min()) >= 124.0 and float(u8.max()) <= 208.0


def test_crop_flip_rescale_oracle():
    g = np.random.Generator(np.random.PCG64(0))
    x = g.integers(0, 256, size=(3, 6, 8, 3), dtype=np.uint8)
    assert A.center_crop_offsets(6, 8, 3, 4) == (1, 2) and A.center_crop_offsets(7, 8, 4, 3) == (1, 2)
    c = A.crop_flip(x, 4, 4, A.center_crop_offsets(6, 8, 4, 4))
    assert np.array_equal(c, x[:, 1:5, 2:6])
    f = A.crop_flip(x, 6, 8, (0, 0), flips=np.array([1, 2, 3], dtype=np.uint8))
    assert np.array_equal(f[0], x[0, :, ::-1]) and np.array_equal(f[1], x[1, ::-1]) and np.array_equal(f[2], x[2, ::-1, ::-1])
    r = A.rescale(x, 1.0 / 127.5, -1.0)
    assert r.dtype == np.float32 and float(r.min()) >= -1.0 and float(r.max()) <= 1.0


def test_layer_configs_and_validation():
    from chambers_amd import augmentations as aug
    assert aug.Resizing(224, 224).get_config()["interpolation"] == "bilinear"
    assert aug.CenterCrop(10, 12).compute_output_shape((None, 20, 20, 3)) == (None, 10, 12, 3)
    assert set(aug.RandomCrop(8, 8, seed=1).get_config()) >= {"height", "width", "seed"}
    assert aug.RandomFlip("horizontal").get_config()["mode"] == "horizontal"
    with pytest.raises(ValueError):
        aug.RandomFlip("diagonal")
    with pytest.raises(NotImplementedError):
        aug.Resizing(4, 4, interpolation="bicubic")
    assert aug.Rescaling(1 / 255.0).get_config()["offset"] == 0.0


# ------------------------------------------------------------------------------------ GPU tier
def _dev(x):
    return torch.as_tensor(x, device="cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("kw,shape", REF_SHAPES)
def test_hip_resizing_minmax_reference_shapes(kw, shape):
    from chambers_amd.augmentations import ResizingMinMax
    out = ResizingMinMax(**kw)(_dev(IMG_NOT_SQUARE))
    assert tuple(out.shape) == shape and out.dtype == torch.float32
    np.testing.assert_array_equal(out.cpu().numpy(), A.resize(IMG_NOT_SQUARE, shape[1], shape[2]))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,out_hw", [((2, 37, 53, 3), (224, 224)), ((3, 300, 260, 3), (224, 224)), ((1, 224, 224, 3), (224, 224)),
                                          ((2, 16, 16, 1), (5, 7)), ((1, 9, 10, 4), (18, 20))])
def test_hip_resize_matches_oracle(shape, out_hw):
    from chambers_amd import augmentations as aug
    g = np.random.Generator(np.random.PCG64(3))
    x = g.integers(0, 256, size=shape, dtype=np.uint8)
    for data in (x, (g.normal(0, 50, size=shape)).astype(np.float32)):
        out = aug.Resizing(*out_hw)(_dev(data))
        np.testing.assert_array_equal(out.cpu().numpy(), A.resize(data, *out_hw))            # same fp32 operation order: bit-exact
        near = aug.Resizing(*out_hw, interpolation="nearest")(_dev(data))
        assert near.dtype == _dev(data).dtype
        np.testing.assert_array_equal(near.cpu().numpy(), A.resize(data, *out_hw, "nearest"))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 40, 48, 3), (3, 33, 31, 3), (2, 20, 24, 1)])
def test_hip_crop_flip_rescale_match_oracle(shape):
    from chambers_amd import augmentations as aug
    g = np.random.Generator(np.random.PCG64(4))
    b, h, w, _ = shape
    for data in (g.integers(0, 256, size=shape, dtype=np.uint8), g.normal(size=shape).astype(np.float32)):
        xd = _dev(data)
        th, tw = h - 8, w - 12
        np.testing.assert_array_equal(aug.CenterCrop(th, tw)(xd).cpu().numpy(), A.crop_flip(data, th, tw, A.center_crop_offsets(h, w, th, tw)))
        np.testing.assert_array_equal(aug.RandomCrop(th, tw)(xd, training=True, offset=(3, 5)).cpu().numpy(), A.crop_flip(data, th, tw, (3, 5)))
        np.testing.assert_array_equal(aug.RandomCrop(th, tw)(xd, training=False).cpu().numpy(), aug.CenterCrop(th, tw)(xd).cpu().numpy())
        fh, fv = g.uniform(size=b) < 0.5, g.uniform(size=b) < 0.5
        out = aug.RandomFlip()(xd, training=True, flip_horizontal=fh, flip_vertical=fv)
        np.testing.assert_array_equal(out.cpu().numpy(), A.crop_flip(data, h, w, (0, 0), fh.astype(np.uint8) | (fv.astype(np.uint8) << 1)))
        assert torch.equal(aug.RandomFlip()(xd, training=False), xd)
        only_h = aug.RandomFlip("horizontal")(xd, training=True, flip_horizontal=np.ones(b, bool))
        np.testing.assert_array_equal(only_h.cpu().numpy(), data[:, :, ::-1])
        np.testing.assert_array_equal(aug.Rescaling(1 / 127.5, offset=-1.0)(xd).cpu().numpy(), A.rescale(data, 1 / 127.5, -1.0))
        from chambers_amd import kernels as K
        offs = np.stack([g.integers(0, 9, size=b), g.integers(0, 13, size=b)], axis=1).astype(np.int32)
        np.testing.assert_array_equal(K.crop_flip(xd, th, tw, offsets=offs).cpu().numpy(), A.crop_flip(data, th, tw, offs))   # per-image windows
    with pytest.raises(ValueError):
        aug.CenterCrop(h + 1, w)(_dev(np.zeros(shape, np.uint8)))


@pytest.mark.gpu
def test_hip_input_pipeline_into_randaugment():
    """decode-sized batch -> Resizing -> RandomCrop -> RandomFlip -> uint8 -> RandAugment: the pieces compose on the device."""
    from chambers_amd import augmentations as aug
    g = np.random.Generator(np.random.PCG64(5))
    x = g.integers(0, 256, size=(4, 300, 280, 3), dtype=np.uint8)
    y = aug.Resizing(256, 256)(_dev(x))
    y = aug.RandomCrop(224, 224)(y, training=True, offset=(7, 9))
    y = aug.RandomFlip("horizontal")(y, training=True, flip_horizontal=np.array([1, 0, 1, 0], bool))
    ref = A.crop_flip(A.crop_flip(A.resize(x, 256, 256), 224, 224, (7, 9)), 224, 224, (0, 0), np.array([1, 0, 1, 0], np.uint8))
    np.testing.assert_array_equal(y.cpu().numpy(), ref)
    u8 = y.clamp(0, 255).to(torch.uint8)                                      # truncating cast back to the uint8 the schemes expect
    out = aug.RandAugment(2, 9)(u8, training=True, decisions=[{"op": 2, "negate": False, "centers": np.zeros((4, 2), np.int32)},
                                                               {"op": 11, "negate": False, "centers": np.zeros((4, 2), np.int32)}])
    np.testing.assert_array_equal(out.cpu().numpy(), A.rand_augment(ref.astype(np.uint8), 2, 9, [{"op": 2}, {"op": 11}]))
