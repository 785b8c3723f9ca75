"""GPU parity of the fused scheme stage (chb_aug_fused): a batch-shared RandAugment / AutoAugment chain evaluated per output
pixel in one pass, optionally with ImageNetNormalization("tf") + the ViT patch gather folded in.  The bar is the chain run op
by op: bit-exact against the oracle (augmentation_schemes.py:151-160, 204-213 -> image_augmentations.py op by op), for every
ordered pair of the 16 RandAugment ops, longer chains, the 25 AutoAugment sub-policies and the BASELINE batch sizes."""
import itertools

import numpy as np
import pytest
import torch

from oracle import augment_ref as A

pytestmark = pytest.mark.gpu


def _img(shape, seed=0):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=shape, dtype=np.uint8)


def _dev(x):
    return torch.as_tensor(x, device="cuda")


def _decisions(g, ops, b, h, w):
    return [{"op": int(op), "negate": bool(g.uniform() < 0.5),
             "centers": np.stack([g.integers(0, h, size=b), g.integers(0, w, size=b)], axis=1).astype(np.int32)} for op in ops]


def _eq(out, ref, what):
    got = out.cpu().numpy()
    assert got.dtype == ref.dtype and got.shape == ref.shape, (got.dtype, got.shape, ref.dtype, ref.shape)
    diff = got != ref
    if diff.any():
        raise AssertionError("%s: %d / %d values differ (first at %s)" % (what, int(diff.sum()), ref.size, np.argwhere(diff)[0].tolist()))


def _patch_rows(x_u8, p):
    """ImageNetNormalization("tf") -> bf16 -> the [B*gh*gw, p*p*3] patch rows the ViT embeds (vision_transformer.py:235-248)."""
    b, h, w, _ = x_u8.shape
    gh, gw = h // p, w // p
    f = A.imagenet_normalize(x_u8[:, :gh * p, :gw * p], "tf").reshape(b, gh, p, gw, p, 3).transpose(0, 1, 3, 2, 4, 5).reshape(b * gh * gw, p * p * 3)
    return torch.from_numpy(np.ascontiguousarray(f)).to(torch.bfloat16)


def test_every_ordered_pair_of_randaugment_ops():
    """16 x 16 chains on a ragged 3-image batch (W % 4 != 0; one low-contrast image, one constant image)."""
    from chambers_amd import augmentations as aug
    shape = (3, 29, 38, 3)
    x = _img(shape, 3)
    x[1] = (x[1] // 6) + 30
    x[2] = 77
    xd = _dev(x)
    layer = aug.RandAugment(2, 9)
    g = np.random.Generator(np.random.PCG64(5))
    for a, b in itertools.product(range(16), range(16)):
        dec = _decisions(g, (a, b), *shape[:3])
        _eq(layer(xd, training=True, decisions=dec), A.rand_augment(x, 2, 9, dec), "%s -> %s" % (A.RANDAUGMENT_OPS[a], A.RANDAUGMENT_OPS[b]))


@pytest.mark.parametrize("n_ops", [1, 3, 4])
def test_longer_chains(n_ops):
    from chambers_amd import augmentations as aug
    shape = (4, 40, 48, 3)
    x = _img(shape, 13)
    x[0] = (x[0] // 4) + 100
    xd = _dev(x)
    layer = aug.RandAugment(n_ops, 9)
    g = np.random.Generator(np.random.PCG64(100 + n_ops))
    chains = [tuple(int(v) for v in g.integers(0, 16, size=n_ops)) for _ in range(40)]
    chains += [(6,) * n_ops, (1,) * n_ops, tuple([6, 1, 0, 15][:n_ops]), tuple([15, 6, 14, 1][:n_ops])]   # Sharpness / Equalize stacks
    for ops in chains:
        dec = _decisions(g, ops, *shape[:3])
        _eq(layer(xd, training=True, decisions=dec), A.rand_augment(x, n_ops, 9, dec), "chain %s" % (ops,))


def test_fused_equals_op_by_op_route_and_drawn_decisions():
    """The same host-generator seed gives the same output through the fused launch and through one launch per op (the draw
    order of plan() is the op-by-op route's); a 5-op chain is beyond the fused kernel and takes the op-by-op route."""
    from chambers_amd import augmentations as aug
    from chambers_amd import rng
    x = _dev(_img((5, 32, 32, 3), 17))
    for seed in range(12):
        layer = aug.RandAugment(2, 9)
        rng.set_seed(seed)
        fused = layer(x, training=True)
        layer._transform.fused = False
        rng.set_seed(seed)
        assert torch.equal(fused, layer(x, training=True)), seed
    g = np.random.Generator(np.random.PCG64(2))
    dec = _decisions(g, (6, 0, 14, 3, 15), 5, 32, 32)
    _eq(aug.RandAugment(5, 9)(x, training=True, decisions=dec), A.rand_augment(x.cpu().numpy(), 5, 9, dec), "5-op chain")


@pytest.mark.parametrize("shape", [(4, 64, 64, 3), (2, 33, 47, 3)])
def test_autoaugment_batch_shared_all_policies(shape):
    from chambers_amd import augmentations as aug
    x = _img(shape, 23)
    x[0] = (x[0] // 3) + 20
    layer = aug.AutoAugment()
    for policy in range(25):
        for apply in ((True, True), (True, False), (False, True), (False, False)):
            dec = {"policy": policy, "apply": apply, "negate": (bool(policy & 1), bool(policy & 2))}
            _eq(layer(_dev(x), training=True, decision=dec), A.auto_augment(x, dec), "policy %d apply %s" % (policy, apply))


@pytest.mark.parametrize("shape,p", [((3, 64, 48, 3), 16), ((2, 50, 70, 3), 16), ((2, 32, 32, 3), 8)])
def test_patch_rows_of_the_normalised_chain(shape, p):
    """patch > 0: the bf16 patch rows of the "tf"-normalised chain output, ragged edges dropped as the patch gather does."""
    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    x = _img(shape, 29)
    layer = aug.RandAugment(2, 9)
    g = np.random.Generator(np.random.PCG64(31))
    for ops in [(0, 15), (6, 7), (14, 1), (4, 5), (9, 6), (1, 0), (2, 11), (12, 13), (3, 8), (10, 6)]:
        dec = _decisions(g, ops, *shape[:3])
        plan = layer.plan(shape, dec)
        rows = K.aug_fused(_dev(x), plan, patch=p)
        ref = _patch_rows(A.rand_augment(x, 2, 9, dec), p)
        assert torch.equal(rows.cpu().view(torch.int16), ref.view(torch.int16)), ops


@pytest.mark.parametrize("size,batch", [(224, 512), (384, 64)])
def test_full_size_batches(size, batch):
    """BASELINE sizes (config 3: [512,224,224,3]; config 5 resolution).  The oracle checks a slice of the batch - every image's
    chain is independent apart from Contrast's batch constant, which the decisions carry through `plan`."""
    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    shape = (batch, size, size, 3)
    x = _img(shape, 37)
    xd = _dev(x)
    layer = aug.RandAugment(2, 9)
    g = np.random.Generator(np.random.PCG64(41))
    sl = slice(batch - 6, batch)
    for ops in [(15, 6), (1, 8), (6, 0), (14, 5), (11, 12)]:
        dec = _decisions(g, ops, *shape[:3])
        out = layer(xd, training=True, decisions=dec)
        dec_s = [dict(d, centers=d["centers"][sl]) for d in dec]
        _eq(out[sl], A.rand_augment(x[sl], 2, 9, dec_s), "ops %s at %d^2" % (ops, size))
        # the rest of the batch: against the op-by-op GPU route (itself oracle-checked in test_augment_gpu.py)
        layer._transform.fused = False
        assert torch.equal(out, layer(xd, training=True, decisions=dec)), ops
        layer._transform.fused = True
        rows = K.aug_fused(xd, layer.plan(shape, dec), patch=16)
        assert torch.equal(rows, K.normalize_patchify(out, 16, "tf")), ops


@pytest.mark.parametrize("shape", [(2, 21, 260, 3), (2, 18, 250, 3), (1, 9, 516, 3)])
def test_wide_rows_sharpness_windows_and_row_warps(shape):
    """Rows wider than one 62-quad window of the Sharpness launch (the halo crosses a window seam), W % 4 != 0, warps whose source
    row does not depend on x (one run per quad, also nested and under / above a Sharpness), patch rows that crop the image."""
    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    x = _img(shape, 83)
    x[0, :, :40] = 200                      # a flat area: ties and fill-like values
    g = np.random.Generator(np.random.PCG64(89))
    chains = [(9, 6), (10, 6), (7, 6), (8, 6), (15, 6), (2, 6), (6, 2), (6, 14), (6, 5), (6, 9), (6, 1), (1, 6), (9, 10), (7, 9), (10, 7), (7, 7),
              (9, 6, 3), (2, 9, 6), (7, 14, 6, 4), (6, 3, 6), (10, 1, 9)]
    for ops in chains:
        layer = aug.RandAugment(len(ops), 9)
        for trial in range(2):              # both signs turn up
            dec = _decisions(g, ops, *shape[:3])
            plan = layer.plan(shape, dec)
            ref = A.rand_augment(x, len(ops), 9, dec)
            _eq(K.aug_fused(_dev(x), plan, scratch=True), ref, "cut %s" % (ops,))
            _eq(K.aug_fused(_dev(x), plan, scratch=False), ref, "uncut %s" % (ops,))
            rows = K.aug_fused(_dev(x), plan, patch=4)
            assert torch.equal(rows.cpu().view(torch.int16), _patch_rows(ref, 4).view(torch.int16)), ops


def test_uncut_chains_equal_cut_chains():
    """scratch=False: a Sharpness above other ops evaluates them at its nine taps instead of reading a materialised image - the
    same bytes either way (and the route a C caller without scratch memory takes)."""
    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    g = np.random.Generator(np.random.PCG64(73))
    for shape in [(3, 24, 28, 3), (2, 19, 30, 3)]:
        x = _img(shape, 71)
        for ops in [(5, 6), (6, 6), (15, 6), (1, 6), (14, 6, 6), (6, 3, 6, 6), (0, 6, 15, 6), (7, 15), (3, 4, 5, 1), (15, 1, 0), (2, 8, 12)]:
            layer = aug.RandAugment(len(ops), 9)
            dec = _decisions(g, ops, *shape[:3])
            plan = layer.plan(shape, dec)
            ref = A.rand_augment(x, len(ops), 9, dec)
            _eq(K.aug_fused(_dev(x), plan, scratch=False), ref, "uncut %s" % (ops,))
            _eq(K.aug_fused(_dev(x), plan, scratch=True), ref, "cut %s" % (ops,))
            if shape[1] >= 16 and shape[2] >= 16:
                rows = K.aug_fused(_dev(x), plan, patch=8, scratch=False)
                assert torch.equal(rows.cpu().view(torch.int16), _patch_rows(ref, 8).view(torch.int16)), ops


def test_contrast_constant_is_the_batch_tensors():
    """Batch-shared Contrast blends towards B*H*W/256 clipped to 255 (image_augmentations.py:253-257), not one image's H*W/256."""
    from chambers_amd import augmentations as aug
    x = _img((2, 16, 16, 3), 43)                     # 2*16*16/256 = 2; one image alone would give 1
    dec = [{"op": 4}, {"op": 2}]
    out = aug.RandAugment(2, 9)(_dev(x), training=True, decisions=dec).cpu().numpy()
    np.testing.assert_array_equal(out, A.invert(A.blend(np.full_like(x, 2), x, 9 / 10 * 1.8 + 0.1)))


def test_engine_consumes_the_plan():
    """ViTEngine.forward(images, augment=plan) == forward(scheme(images)): the chain runs inside the patchify pass."""
    from chambers_amd import augmentations as aug
    from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights
    cfg = ViTConfig(patch_size=16, patch_dim=128, n_encoder_layers=1, n_heads=2, ff_dim=128, image_size=(64, 64), classes=10, dropout_rate=0.0)
    eng = ViTEngine(cfg, 4, training=False, seed=0)
    eng.load_keras_weights(init_keras_weights(cfg, seed=1))
    x = _dev(_img((4, 64, 64, 3), 47))
    layer = aug.RandAugment(2, 9)
    g = np.random.Generator(np.random.PCG64(53))
    for ops in [(15, 6), (1, 14), (4, 9)]:
        dec = _decisions(g, ops, 4, 64, 64)
        a = eng.forward(layer(x, training=True, decisions=dec), training=False).clone()
        b = eng.forward(x, training=False, augment=layer.plan(x.shape, dec)).clone()
        assert torch.equal(a, b), ops


def test_edge_shapes_and_errors():
    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    g = np.random.Generator(np.random.PCG64(59))
    for shape in [(2, 1, 1, 3), (3, 3, 5, 3), (1, 2, 9, 3)]:
        x = _img(shape, 61)
        for ops in [(6, 15), (0, 1), (14, 7), (6, 6)]:
            dec = _decisions(g, ops, *shape[:3])
            _eq(aug.RandAugment(2, 9)(_dev(x), training=True, decisions=dec), A.rand_augment(x, 2, 9, dec), "%s %s" % (shape, ops))
    e = torch.empty((0, 8, 8, 3), dtype=torch.uint8, device="cuda")
    assert aug.RandAugment(2, 9)(e, training=True).shape == e.shape
    x = _dev(_img((2, 8, 8, 3), 1))
    plan = aug.RandAugment(2, 9).plan(x.shape, _decisions(g, (14, 2), 2, 8, 8))
    plan.centers[0] = np.zeros((3, 2), np.int32)
    with pytest.raises(ValueError):
        K.aug_fused(x, plan)
    with pytest.raises(ValueError):
        K.aug_fused(x, K.AugPlan([]))
    with pytest.raises(ValueError):
        K.aug_fused(x, aug.RandAugment(2, 9).plan(x.shape, _decisions(g, (3, 2), 2, 8, 8)), patch=6)     # patch must be a multiple of 4
