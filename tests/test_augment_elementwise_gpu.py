"""GPU parity of the per-image (elementwise=True) augmentation path: RandomChoice / RandAugment / AutoAugment with
`elementwise=True` run ONE dispatch launch per slot (chb_aug_dispatch) and must equal the oracle applied image by image
to batch-1 tensors — the semantics of the reference's tf.map_fn (image_augmentations.py:563-570, 606-617;
augmentation_schemes.py:138-149,193): per-image op index, per-image sign draw, per-image cutout centre, and Contrast's
constant computed from ONE image (196 for 224x224).  Bit-exact, uint8."""
import numpy as np
import pytest
import torch

from oracle import augment_ref as A

pytestmark = pytest.mark.gpu


def _img(shape, seed=0):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=shape, dtype=np.uint8)


def _dev(x):
    return torch.as_tensor(x, device="cuda")


def _eq(out, ref, what=""):
    got = out.cpu().numpy()
    assert got.dtype == np.uint8 and got.shape == ref.shape
    diff = got != ref
    if diff.any():
        bad_images = sorted(set(np.nonzero(diff)[0].tolist()))
        raise AssertionError("%s: %d / %d bytes differ; images %s" % (what, int(diff.sum()), ref.size, bad_images[:16]))


def _rand_decisions(g, n_slots, b, h, w, ops=None):
    out = []
    for _ in range(b):
        out.append([{"op": int(g.integers(0, 16)) if ops is None else int(ops[int(g.integers(0, len(ops)))]),
                     "negate": bool(g.uniform() < 0.5),
                     "centers": np.array([[int(g.integers(0, h)), int(g.integers(0, w))]], dtype=np.int32)} for _ in range(n_slots)])
    return out


@pytest.mark.parametrize("shape,n_slots", [((48, 64, 64, 3), 2), ((7, 37, 50, 3), 3), ((5, 9, 10, 3), 2), ((3, 20, 24, 3), 1),
                                           ((2, 1, 1, 3), 2), ((4, 3, 5, 3), 2), ((3, 256, 256, 3), 2)])
def test_randaugment_elementwise_matches_oracle_per_image(shape, n_slots):
    from chambers_amd import augmentations as aug
    b, h, w, _ = shape
    x = _img(shape, 11)
    x[0] = (x[0] // 5) + 40                      # low dynamic range: AutoContrast / Equalize do real work
    if b > 1:
        x[1] = 93                                # constant image: their identity branches
    g = np.random.Generator(np.random.PCG64(1000 + b))
    dec = _rand_decisions(g, n_slots, b, h, w)
    for n in range(min(b, 16)):                  # make sure every op occurs at least once in every case that has room
        dec[n][0]["op"] = n % 16
    layer = aug.RandAugment(n_slots, 9, elementwise=True)
    _eq(layer(_dev(x), training=True, decisions=dec), A.rand_augment_elementwise(x, n_slots, 9, dec), "RandAugment elementwise")
    # identity when not training (augmentation_schemes.py:204-213)
    assert torch.equal(layer(_dev(x), training=False), _dev(x))


def test_randaugment_elementwise_full_batch_512x224():
    """BASELINE config 3 size: [512,224,224,3], RandAugment(2, 9), per-image decisions from seed 42."""
    from chambers_amd import augmentations as aug
    shape = (512, 224, 224, 3)
    x = _img(shape, 0)
    g = np.random.Generator(np.random.PCG64(42))
    dec = _rand_decisions(g, 2, 512, 224, 224)
    out = aug.RandAugment(2, 9, elementwise=True)(_dev(x), training=True, decisions=dec)
    _eq(out, A.rand_augment_elementwise(x, 2, 9, dec), "RandAugment elementwise 512x224x224")


@pytest.mark.parametrize("shape", [(6, 40, 56, 3), (9, 64, 64, 3), (3, 224, 224, 3)])
@pytest.mark.parametrize("second", [1, 0])          # Equalize / AutoContrast: the table ops, whose histogram pass evaluates the level below
def test_cutout_first_chains_under_a_table_op_small_shapes(shape, second):
    """the shape of chain the hipcc 7.2 byte-select fault corrupted (CutOut FIRST, then Equalize / AutoContrast: the
    histogram pass counted CutOut's value for the wrong pixels, in two of three channels) - per-image route, every image its own
    centre, at SMALL shapes; in round 3 only the 512 x 224 x 224 batch caught it.  Mixed with chains that start with another op, so
    the sorted launches see several groups.  Bit-exact against the oracle, patch-row output included."""
    from chambers_amd import augmentations as aug
    b, h, w, _ = shape
    x = _img(shape, 21 + second)
    x[0] = (x[0] // 3) + 60
    g = np.random.Generator(np.random.PCG64(77 + b))
    dec = []
    for n in range(b):
        first = 14 if n % 3 != 2 else int(g.integers(0, 16))          # CutOut first in two of three images
        dec.append([{"op": first, "negate": bool(g.uniform() < 0.5), "centers": np.array([[int(g.integers(0, h)), int(g.integers(0, w))]], dtype=np.int32)},
                    {"op": second, "negate": False, "centers": np.zeros((1, 2), np.int32)}])
    layer = aug.RandAugment(2, 9, elementwise=True)
    _eq(layer(_dev(x), training=True, decisions=dec), A.rand_augment_elementwise(x, 2, 9, dec), "CutOut-first chains under a table op")


def test_contrast_constant_and_signs_are_per_image():
    """What elementwise mode changes semantically: Contrast's constant is H*W/256 of ONE image
    (196 at 224x224; the batch-shared mode clips B*H*W/256 to 255), and the sign of a warp is drawn per image."""
    from chambers_amd import augmentations as aug
    shape = (4, 224, 224, 3)
    x = _img(shape, 5)
    dec = [[{"op": 4, "negate": False, "centers": np.zeros((1, 2), np.int32)}] for _ in range(4)]        # Contrast everywhere
    out = aug.RandAugment(1, 9, elementwise=True)(_dev(x), training=True, decisions=dec).cpu().numpy()
    np.testing.assert_array_equal(out, A.blend(np.full_like(x, 196), x, 9 / 10 * 1.8 + 0.1))
    shared = aug.RandAugment(1, 9)(_dev(x), training=True, decisions=dec[0]).cpu().numpy()
    np.testing.assert_array_equal(shared, A.blend(np.full_like(x, 255), x, 9 / 10 * 1.8 + 0.1))
    assert (out != shared).any()
    dec = [[{"op": 7, "negate": bool(n & 1), "centers": np.zeros((1, 2), np.int32)}] for n in range(4)]   # ShearX, alternating sign
    out = aug.RandAugment(1, 9, elementwise=True)(_dev(x), training=True, decisions=dec).cpu().numpy()
    for n in range(4):
        np.testing.assert_array_equal(out[n:n + 1], A.projective_transform(x[n:n + 1], A.shear_x_transform(0.27, bool(n & 1)), 128))


@pytest.mark.parametrize("shape", [(100, 64, 64, 3), (6, 33, 47, 3), (3, 224, 224, 3)])
def test_autoaugment_elementwise_all_policies(shape):
    from chambers_amd import augmentations as aug
    b = shape[0]
    x = _img(shape, 21)
    x[0] = (x[0] // 3) + 10
    g = np.random.Generator(np.random.PCG64(77))
    dec = [{"policy": n % 25, "apply": (bool(g.uniform() < 0.7), bool(g.uniform() < 0.7)),
            "negate": (bool(g.uniform() < 0.5), bool(g.uniform() < 0.5))} for n in range(b)]
    _eq(aug.AutoAugment(elementwise=True)(_dev(x), training=True, decision=dec), A.auto_augment_elementwise(x, dec), "AutoAugment elementwise")


def test_random_choice_elementwise_draw_order_and_generic_route():
    """Without explicit decisions the layer draws, image by image and slot by slot, the transform index and then whatever the
    chosen transform draws itself (sign, cutout centre): replaying the host generator reproduces the output through the
    oracle.  A RandomChoice over layers that cannot describe themselves to the dispatch kernel takes the image-by-image route."""
    from chambers_amd import augmentations as aug
    from chambers_amd import rng
    shape = (9, 32, 40, 3)
    x = _img(shape, 31)
    layer = aug.RandAugment(2, 9, elementwise=True)
    rng.set_seed(123)
    out = layer(_dev(x), training=True)
    g = np.random.Generator(np.random.PCG64(123))
    dec = []
    for n in range(shape[0]):
        ds = []
        for _ in range(2):
            d = {"op": int(g.integers(0, 16))}
            name = A.RANDAUGMENT_OPS[d["op"]]
            if name in ("ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate"):
                d["negate"] = bool(g.uniform() < 0.5)
            if name == "CutOut":
                d["centers"] = np.array([[int(g.integers(0, shape[1])), int(g.integers(0, shape[2]))]], dtype=np.int32)
            ds.append(d)
        dec.append(ds)
    _eq(out, A.rand_augment_elementwise(x, 2, 9, dec), "drawn decisions")

    choices = [[n & 1] for n in range(shape[0])]
    ref = np.concatenate([A.invert(x[n:n + 1]) if not (n & 1) else A.solarize(x[n:n + 1], 100) for n in range(shape[0])])
    choice = aug.RandomChoice([aug.Invert(), aug.Solarize(100)], n_transforms=1, elementwise=True)
    _eq(choice(_dev(x), choices=choices), ref, "RandomChoice elementwise (dispatch kernel)")

    from chambers_amd import kernels as K
    from chambers_amd._keras_like import Layer

    class UserInvert(Layer):          # a user-supplied layer that cannot describe itself to the dispatch kernel
        _forward_kwargs = True

        def call(self, inputs, **kwargs):
            return K.aug_pointwise(inputs, K.PW_INVERT)

    choice = aug.RandomChoice([UserInvert(), aug.Solarize(100)], n_transforms=1, elementwise=True)
    assert not hasattr(choice.transforms[0], "dispatch_item")
    _eq(choice(_dev(x), choices=choices), ref, "RandomChoice elementwise (image-by-image route)")


def test_elementwise_golden_fixture():
    import importlib.util
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden, "make_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    from chambers_amd import augmentations as aug
    with np.load(os.path.join(golden, "augment_ops.npz")) as z:
        ref = {k: z[k] for k in z.files}
    for tag, shape in gen.IMAGE_SHAPES.items():
        x = _dev(ref["x_" + tag])
        out = aug.RandAugment(2, 9, elementwise=True)(x, training=True, decisions=gen.elementwise_randaugment_decisions(shape))
        _eq(out, ref["randaugment_elementwise_" + tag], "golden randaugment " + tag)
        out = aug.AutoAugment(elementwise=True)(x, training=True, decision=gen.elementwise_autoaugment_decisions(shape))
        _eq(out, ref["autoaugment_elementwise_" + tag], "golden autoaugment " + tag)


def test_per_image_chains_as_patch_rows_and_through_the_engine():
    """chb_aug_fused_items with patch > 0 = the "tf"-normalised patch rows of the per-image chains; ViTEngine.forward(images,
    augment=items_plan) = forward(scheme(images))."""
    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights
    shape = (6, 64, 48, 3)
    x = _img(shape, 5)
    g = np.random.Generator(np.random.PCG64(7))
    for trial in range(6):
        dec = [[{"op": int(g.integers(0, 16)), "negate": bool(g.uniform() < 0.5), "centers": (int(g.integers(0, shape[1])), int(g.integers(0, shape[2])))}
                for _ in range(2)] for _ in range(shape[0])]
        layer = aug.RandAugment(2, 9, elementwise=True)
        ref = A.rand_augment_elementwise(x, 2, 9, dec)
        plan = layer.items_plan(shape, dec)
        _eq(K.aug_fused_items(_dev(x), plan), ref, "per-image chains, uint8")
        rows = K.aug_fused_items(_dev(x), plan, patch=16)
        assert torch.equal(rows, K.normalize_patchify(_dev(ref), 16, "tf")), trial
    cfg = ViTConfig(patch_size=16, patch_dim=128, n_encoder_layers=1, n_heads=2, ff_dim=128, image_size=(64, 64), classes=10, dropout_rate=0.0)
    eng = ViTEngine(cfg, 4, training=False, seed=0)
    eng.load_keras_weights(init_keras_weights(cfg, seed=1))
    xi = _dev(_img((4, 64, 64, 3), 9))
    layer = aug.AutoAugment(elementwise=True)
    dec = [{"policy": int(g.integers(0, 25)), "apply": (bool(g.uniform() < 0.5), bool(g.uniform() < 0.5)), "negate": (bool(g.uniform() < 0.5), bool(g.uniform() < 0.5))}
           for _ in range(4)]
    a = eng.forward(layer(xi, training=True, decision=dec), training=False).clone()
    b = eng.forward(xi, training=False, augment=layer.items_plan(xi.shape, dec)).clone()
    assert torch.equal(a, b)


def test_per_image_cutout_without_a_centre_table_is_left_out():
    """chb_aug_fused_items cannot see the device records on the host: a CutOut record at a level whose centre table is NULL must not
    dereference it - the image passes that level untouched."""
    import ctypes
    from chambers_amd import _lib
    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    shape = (3, 20, 24, 3)
    x = _img(shape, 21)
    dec = [[{"op": 14, "negate": False, "centers": (5, 6)}, {"op": 2, "negate": False, "centers": (1, 1)}] for _ in range(3)]
    plan = aug.RandAugment(2, 9, elementwise=True).items_plan(shape, dec)
    xd = _dev(x)
    dev_items, _centers, tables = plan.resident(xd.device)[:3]
    out = torch.empty_like(xd)
    cptr = (ctypes.c_void_p * 2)()          # no centre tables at all
    _lib.call("chb_aug_fused_items", _lib.ptr(xd), _lib.ptr(out), 3, 20, 24, 2, _lib.ptr(dev_items), ctypes.cast(cptr, ctypes.c_void_p), tables, None, 0, K._s())
    _eq(out, A.invert(x), "CutOut without centres -> Invert alone")


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n_slots,patch", [((64, 48, 64, 3), 2, 0), ((40, 32, 48, 3), 3, 16), ((9, 21, 30, 3), 4, 0), ((1, 16, 16, 3), 2, 0)])
def test_sorted_groups_give_the_bytes_of_the_single_launch(shape, n_slots, patch):
    """chb_aug_fused_items_sorted (one launch per group of chains, histograms over the images that need one, more slices for fewer
    images) against chb_aug_fused_items (every image through the general evaluators) and the oracle, table ops at every level."""
    import ctypes
    from chambers_amd import _lib
    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    b, h, w, _ = shape
    g = np.random.Generator(np.random.PCG64(700 + n_slots + b))
    x = _img(shape, 31 + b)
    dec = _rand_decisions(g, n_slots, b, h, w)
    layer = aug.RandAugment(n_slots, 9, elementwise=True)
    plan = layer.items_plan(shape, dec)
    xd = _dev(x)
    dev_items, centers, tables, order, counts = plan.resident(xd.device, h, w)
    assert int(counts[:K.ITEMS_GROUPS].sum()) == b
    cptr = (ctypes.c_void_p * n_slots)()
    for l, c_ in enumerate(centers):
        if c_ is not None:
            cptr[l] = c_.data_ptr()
    nt = bin(tables).count("1")
    ws = torch.empty(max(_lib.aug_fused_workspace_ints(b, h, w, nt), 1), dtype=torch.int32, device=xd.device)
    if patch:
        one = torch.empty((b * (h // patch) * (w // patch), patch * patch * 3), dtype=torch.bfloat16, device=xd.device)
    else:
        one = torch.empty_like(xd)
    _lib.call("chb_aug_fused_items", _lib.ptr(xd), _lib.ptr(one), b, h, w, n_slots, _lib.ptr(dev_items), ctypes.cast(cptr, ctypes.c_void_p), tables,
              _lib.ptr(ws), patch, K._s())
    grouped = K.aug_fused_items(xd, plan, patch=patch or None)
    torch.cuda.synchronize()
    assert torch.equal(grouped.view(torch.uint8) if not patch else grouped.view(torch.int16), one.view(torch.uint8) if not patch else one.view(torch.int16))
    if not patch:
        _eq(grouped, A.rand_augment_elementwise(x, n_slots, 9, dec), "sorted groups vs oracle")


@pytest.mark.gpu
def test_sorted_entry_checks_its_group_counts():
    import ctypes
    from chambers_amd import _lib
    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    shape = (4, 16, 16, 3)
    g = np.random.Generator(np.random.PCG64(5))
    dec = _rand_decisions(g, 2, 4, 16, 16)
    plan = aug.RandAugment(2, 9, elementwise=True).items_plan(shape, dec)
    xd = _dev(_img(shape, 3))
    dev_items, centers, tables, order, counts = plan.resident(xd.device, 16, 16)
    cptr = (ctypes.c_void_p * 2)()
    for l, c_ in enumerate(centers):
        if c_ is not None:
            cptr[l] = c_.data_ptr()
    ws = torch.empty(max(_lib.aug_fused_workspace_ints(4, 16, 16, 2), 1), dtype=torch.int32, device=xd.device)
    out = torch.empty_like(xd)
    bad = counts.copy()
    bad[0] += 1                                    # group sizes of row 0 must add up to B
    lib = _lib.load()
    rc = lib.chb_aug_fused_items_sorted(_lib.ptr(xd), _lib.ptr(out), 4, 16, 16, 2, _lib.ptr(dev_items), ctypes.cast(cptr, ctypes.c_void_p), tables,
                                        _lib.ptr(ws), 0, _lib.ptr(order), bad.ctypes.data, K._s())
    assert rc == _lib.CHB_EINVAL
    rc = lib.chb_aug_fused_items_sorted(_lib.ptr(xd), _lib.ptr(out), 4, 16, 16, 2, _lib.ptr(dev_items), ctypes.cast(cptr, ctypes.c_void_p), tables,
                                        _lib.ptr(ws), 0, None, counts.ctypes.data, K._s())
    assert rc == _lib.CHB_EINVAL


def test_sorted_groups_inside_a_captured_graph_and_on_a_side_stream():
    """The group launches fork onto the library's side streams and join again inside the call: captured into a HIP graph they must
    replay as one unit (new input bytes in the same buffers -> new output), and on a caller's non-default stream they must order with it."""
    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    shape = (48, 32, 48, 3)
    b, h, w, _ = shape
    g = np.random.Generator(np.random.PCG64(77))
    dec = _rand_decisions(g, 2, b, h, w)
    for n in range(16):
        dec[n][0]["op"] = n                       # every op, tables included
        dec[16 + n][1]["op"] = n
    layer = aug.RandAugment(2, 9, elementwise=True)
    plan = layer.items_plan(shape, dec)
    x1, x2 = _img(shape, 1), _img(shape, 2)
    xd = _dev(x1).clone()
    out = torch.empty_like(xd)
    plan.resident(xd.device, h, w)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            K.aug_fused_items(xd, plan, out=out)
        direct = out.clone()
    side.synchronize()
    _eq(direct, A.rand_augment_elementwise(x1, 2, 9, dec), "sorted groups on a side stream")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        K.aug_fused_items(xd, plan, out=out)
    for x in (x2, x1):
        xd.copy_(_dev(x))
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        _eq(out, A.rand_augment_elementwise(x, 2, 9, dec), "sorted groups replayed from a graph")
