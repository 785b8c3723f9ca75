"""GPU parity: every augmentation kernel (through the chambers-API layer classes, i.e. through the
C ABI) against the NumPy oracle — bit-exact, uint8.  Sizes the oracle finishes in seconds; the
BASELINE-size batch is covered by size-independent properties."""
import numpy as np
import pytest
import torch

from oracle import augment_ref as A

pytestmark = pytest.mark.gpu

SHAPES = [(3, 32, 48, 3), (2, 37, 53, 3), (1, 224, 224, 3), (5, 16, 16, 3), (1, 1, 1, 3), (2, 3, 5, 3)]


def _img(shape, seed=0):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=shape, dtype=np.uint8)


def _dev(x):
    return torch.as_tensor(x, device="cuda")


def _eq(out, ref):
    got = out.cpu().numpy()
    assert got.dtype == np.uint8 and got.shape == ref.shape
    bad = int((got != ref).sum())
    assert bad == 0, "%d / %d bytes differ (max |d| = %d)" % (bad, ref.size, int(np.abs(got.astype(int) - ref.astype(int)).max()))


@pytest.mark.parametrize("shape", SHAPES)
def test_pointwise_ops(shape):
    from chambers_amd import augmentations as aug
    x = _img(shape, 1)
    xd = _dev(x)
    _eq(aug.Invert()(xd), A.invert(x))
    for bits in (0, 1, 3, 4, 8):
        _eq(aug.Posterize(bits)(xd), A.posterize(x, bits))
    for thr in (0, 77, 128, 230, 256):
        _eq(aug.Solarize(thr)(xd), A.solarize(x, thr))
    for add, thr in ((99, 128), (0, 128), (-30, 200), (110, 255)):
        _eq(aug.SolarizeAdd(add, thr)(xd), A.solarize_add(x, add, thr))
    for f in (0.0, 0.1, 0.5, 0.99, 1.0, 1.72, 1.9):
        _eq(aug.Brightness(f)(xd), A.brightness(x, f))
        _eq(aug.Color(f)(xd), A.color(x, f))
        _eq(aug.Contrast(f)(xd), A.contrast(x, f))


@pytest.mark.parametrize("shape", SHAPES)
def test_stat_ops(shape):
    from chambers_amd import augmentations as aug
    x = _img(shape, 2)
    # make one image low-dynamic-range and one constant so AutoContrast/Equalize hit their identity branches
    x[0] = (x[0] // 4) + 17
    if shape[0] > 1:
        x[1] = 93
    xd = _dev(x)
    _eq(aug.AutoContrast()(xd), A.autocontrast(x))
    _eq(aug.Equalize()(xd), A.equalize(x))


def test_equalize_skewed_histograms():
    from chambers_amd import augmentations as aug
    g = np.random.Generator(np.random.PCG64(7))
    x = np.clip(g.normal(100, 12, size=(4, 64, 64, 3)), 0, 255).astype(np.uint8)
    x[1, :, :, 0] = 255
    x[2, :32] = 0
    _eq(aug.Equalize()(_dev(x)), A.equalize(x))


@pytest.mark.parametrize("shape", SHAPES)
def test_sharpness(shape):
    from chambers_amd import augmentations as aug
    x = _img(shape, 3)
    for f in (0.0, 0.3, 1.0, 1.72):
        _eq(aug.Sharpness(f)(_dev(x)), A.sharpness(x, f))


@pytest.mark.parametrize("shape", SHAPES)
def test_warps(shape):
    from chambers_amd import augmentations as aug
    x = _img(shape, 4)
    xd = _dev(x)
    h, w = shape[1], shape[2]
    for neg in (False, True):
        _eq(aug.ShearX(0.27, fill_value=128)(xd, negate=neg), A.projective_transform(x, A.shear_x_transform(0.27, neg), 128))
        _eq(aug.ShearY(0.27, fill_value=128)(xd, negate=neg), A.projective_transform(x, A.shear_y_transform(0.27, neg), 128))
        _eq(aug.TranslateX(90.0, fill_value=128)(xd, negate=neg), A.projective_transform(x, A.translate_x_transform(90.0, neg), 128))
        _eq(aug.TranslateY(7.0, fill_value=128)(xd, negate=neg), A.projective_transform(x, A.translate_y_transform(7.0, neg), 128))
        _eq(aug.TranslateX(2.5, fill_value=0)(xd, negate=neg), A.projective_transform(x, A.translate_x_transform(2.5, neg), 0))
        for deg in (27.0, 0.0, 90.0, 3.0):
            _eq(aug.Rotate(deg, fill_value=128)(xd, negate=neg), A.projective_transform(x, A.rotate_transform(deg, neg, h, w), 128))


def test_per_image_and_projective_transforms():
    from chambers_amd import kernels as K
    x = _img((4, 40, 36, 3), 5)
    g = np.random.Generator(np.random.PCG64(5))
    t = np.tile(np.array([1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float32), (4, 1))
    t[:, :6] += g.normal(0, 0.2, size=(4, 6)).astype(np.float32)
    t[:, 6:] = g.normal(0, 0.004, size=(4, 2)).astype(np.float32)   # true projective rows
    out = K.aug_affine(_dev(x), _dev(t), fill=55)
    _eq(out, A.projective_transform(x, t, 55))


@pytest.mark.parametrize("shape", SHAPES)
def test_cutout(shape):
    from chambers_amd import augmentations as aug
    x = _img(shape, 6)
    b, h, w, _ = shape
    g = np.random.Generator(np.random.PCG64(6))
    centers = np.stack([g.integers(0, h, size=b), g.integers(0, w, size=b)], axis=1).astype(np.int32)
    for m in (0, 2, 8, 72):
        _eq(aug.CutOut(m, 128)(_dev(x), centers=centers), A.cutout(x, m, centers, 128))
    corners = np.array([[0, 0]] * b, dtype=np.int32)
    _eq(aug.CutOut(6, 1)(_dev(x), centers=corners), A.cutout(x, 6, corners, 1))
    with pytest.raises(ValueError):
        aug.CutOut(3)(_dev(x), centers=corners)


@pytest.mark.parametrize("mode", ["tf", "torch", "caffe"])
def test_normalization_matches_oracle_and_reference_kat(mode):
    from chambers_amd import augmentations as aug
    from tests.test_oracle_kat import IMG, TARGETS
    out = aug.ImageNetNormalization(mode)(_dev(IMG)).cpu().numpy()
    np.testing.assert_array_equal(out[0, ..., 0], np.array(TARGETS[mode], dtype=np.float32))   # reference KAT, exact
    x = _img((3, 31, 20, 3), 8)
    got = aug.ImageNetNormalization(mode)(_dev(x)).cpu().numpy()
    np.testing.assert_array_equal(got, A.imagenet_normalize(x, mode))
    xf = x.astype(np.float32) * 0.5
    got = aug.ImageNetNormalization(mode)(_dev(xf)).cpu().numpy()
    np.testing.assert_array_equal(got, A.imagenet_normalize(xf, mode))


def test_normalize_patchify_bf16():
    from chambers_amd import kernels as K
    x = _img((3, 64, 48, 3), 9)
    p = 16
    out = K.normalize_patchify(_dev(x), p, "tf").float().cpu().numpy()
    ref = A.imagenet_normalize(x, "tf").reshape(3, 4, p, 3, p, 3).transpose(0, 1, 3, 2, 4, 5).reshape(3 * 12, p * p * 3)
    ref_bf = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(out, ref_bf)
    outf = K.patchify_f32(_dev(A.imagenet_normalize(x, "tf")), p).float().cpu().numpy()
    np.testing.assert_array_equal(outf, ref_bf)


def test_empty_batch_and_bad_args():
    from chambers_amd import augmentations as aug
    e = torch.empty((0, 8, 8, 3), dtype=torch.uint8, device="cuda")
    for layer in (aug.Invert(), aug.Equalize(), aug.AutoContrast(), aug.Sharpness(0.5), aug.ShearX(0.1), aug.Color(0.4)):
        kw = {"negate": False} if isinstance(layer, aug.ShearX) else {}
        assert tuple(layer(e, **kw).shape) == (0, 8, 8, 3)
    with pytest.raises(ValueError):
        aug.ImageNetNormalization("bogus")
    with pytest.raises(ValueError):
        aug.Invert()(torch.zeros((8, 8, 3), dtype=torch.uint8, device="cuda"))   # InputSpec(ndim=4)
    with pytest.raises(ValueError):
        aug.RandAugment(2, 9)(torch.zeros((1, 8, 8, 3), dtype=torch.float32, device="cuda"), training=True)  # dtype uint8


def _decisions(n, b, h, w, g):
    return [{"op": int(g.integers(0, 16)), "negate": bool(g.uniform() < 0.5),
             "centers": np.stack([g.integers(0, h, size=b), g.integers(0, w, size=b)], axis=1).astype(np.int32)} for _ in range(n)]


def test_randaugment_and_autoaugment_schemes():
    from chambers_amd import augmentations as aug
    x = _img((4, 48, 40, 3), 10)
    g = np.random.Generator(np.random.PCG64(42))
    ra = aug.RandAugment(2, 9)
    _eq(ra(_dev(x), training=False), x)          # identity when not training (augmentation_schemes.py:204-213)
    for _ in range(24):
        dec = _decisions(2, 4, 48, 40, g)
        _eq(ra(_dev(x), training=True, decisions=dec), A.rand_augment(x, 2, 9, dec))
    # every op once, in order
    for op in range(16):
        dec = [{"op": op, "negate": True, "centers": np.array([[5, 5]] * 4, dtype=np.int32)}]
        _eq(aug.RandAugment(1, 9)(_dev(x), training=True, decisions=dec), A.rand_augment(x, 1, 9, dec))
    aa = aug.AutoAugment()
    for pol in range(25):
        for apply in ((True, True), (True, False), (False, True)):
            dec = {"policy": pol, "apply": apply, "negate": (bool(pol & 1), bool(pol & 2))}
            _eq(aa(_dev(x), training=True, decision=dec), A.auto_augment(x, dec))


def test_full_size_properties():
    """BASELINE-size batch [256,224,224,3]: properties that do not need the oracle."""
    from chambers_amd import augmentations as aug
    x = torch.randint(0, 256, (256, 224, 224, 3), dtype=torch.uint8, device="cuda")
    assert torch.equal(aug.Invert()(aug.Invert()(x)), x)                              # involution
    assert torch.equal(aug.Solarize(256)(x), x) and torch.equal(aug.Solarize(0)(x), aug.Invert()(x))
    p = aug.Posterize(3)(x)
    assert torch.equal(aug.Posterize(3)(p), p) and int((p & 31).max()) == 0           # idempotent, low bits cleared
    e = aug.Equalize()(x)
    lo = e.amin(dim=(1, 2)).min().item()
    assert lo == 0                                                                    # uniform noise stretches to 0
    ac = aug.AutoContrast()(x)
    assert ac.amin(dim=(1, 2)).max().item() == 0 and ac.amax(dim=(1, 2)).min().item() == 255
    t = aug.TranslateX(16.0, fill_value=128)
    back = aug.TranslateX(16.0, fill_value=128)(t(x, negate=False), negate=True)      # shift left then right
    assert torch.equal(back[:, :, 16:-16], x[:, :, 16:-16])
    c = aug.CutOut(72, 128)(x, centers=np.full((256, 2), 112, dtype=np.int32))
    assert int((c[:, 76:148, 76:148] != 128).sum()) == 0 and torch.equal(c[:, :76], x[:, :76])
    n = aug.ImageNetNormalization("tf")(x)
    assert n.dtype == torch.float32 and float(n.min()) >= -1.0 and float(n.max()) <= 1.0
    # checksum of checksums: rotating by 0 degrees is the identity
    assert torch.equal(aug.Rotate(0.0, fill_value=128)(x, negate=False), x)
