"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle):
CPU tier — the oracle still reproduces every file, and the reference-derived ImageNetNormalization answers stored beside
the oracle's agree exactly; GPU tier — the HIP path through the chambers API / C ABI matches the files (bit-exact for every
uint8 operation and the dropout mask; fp32-oracle tolerances for the bf16 ViT path, stated at each assert)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _load(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


# ------------------------------------------------------------------------------------ CPU tier: oracle vs files
def test_oracle_reproduces_augment_fixture():
    ref, now = _load("augment_ops.npz"), _gen().augment_fixture()
    assert set(ref) == set(now)
    for k in ref:
        assert ref[k].dtype == now[k].dtype, k
        np.testing.assert_array_equal(now[k], ref[k], err_msg=k)
    for mode in ("tf", "torch", "caffe"):   # reference-derived known answers == the oracle on the reference's own test image
        np.testing.assert_array_equal(ref["norm_kat_oracle_" + mode], ref["norm_kat_reference_" + mode])


def test_oracle_reproduces_dropout_fixture():
    ref, now = _load("dropout_mask.npz"), _gen().dropout_fixture()
    for k in ref:
        np.testing.assert_array_equal(now[k], ref[k], err_msg=k)


def test_oracle_reproduces_vit_fixture():
    ref, now = _load("vit_tiny_step.npz"), _gen().vit_fixture()
    assert set(ref) == set(now)
    for k in ref:
        if ref[k].dtype.kind == "f":   # torch-CPU kernels differ by instruction set between hosts: fp32 round-off only
            if k.endswith("b_key"):
                # d(loss)/d(b_key) is exactly zero (softmax shift invariance): the stored gradient is fp32 round-off (~1e-8)
                # and Adam turns it into a ~1e-6 step; bound both instead of comparing noise with noise
                if k.startswith("g/"):
                    assert float(np.abs(now[k]).max()) < 1e-6 and float(np.abs(ref[k]).max()) < 1e-6, k
                else:
                    assert float(np.abs(now[k] - ref[k]).max()) < 2e-5, k
                continue
            if k.startswith("w1/"):
                # Adam's first step is lr * g / (|g| + eps'): where |g| is round-off sized the step follows the round-off
                assert float(np.abs(now[k] - ref[k]).max()) <= 1e-4 and rel_l2(now[k], ref[k]) < 1e-4, k
                continue
            scale = float(np.abs(ref[k]).max()) + 1e-30
            assert float(np.abs(now[k] - ref[k]).max()) <= 2e-5 * scale and rel_l2(now[k], ref[k]) < 1e-5, k
        else:
            np.testing.assert_array_equal(now[k], ref[k], err_msg=k)


# ------------------------------------------------------------------------------------ GPU tier: HIP path vs files
def _dev(x):
    return torch.as_tensor(x, device="cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_augment_ops_match_golden(tag):
    from chambers_amd import augmentations as aug
    gen, ref = _gen(), _load("augment_ops.npz")
    x, centers = _dev(ref["x_" + tag]), ref["centers_" + tag]
    for name, op, kw, neg in gen.AUG_CASES:
        layer = getattr(aug, op)(**kw)
        if op in ("ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate"):
            out = layer(x, negate=neg)
        elif op == "CutOut":
            out = layer(x, centers=centers)
        else:
            out = layer(x)
        np.testing.assert_array_equal(out.cpu().numpy(), ref["%s_%s" % (name, tag)], err_msg="%s_%s" % (name, tag))
    dec = [{"op": 7, "negate": True, "centers": centers}, {"op": 14, "negate": False, "centers": centers}]
    np.testing.assert_array_equal(aug.RandAugment(2, 9)(x, training=True, decisions=dec).cpu().numpy(), ref["randaugment_" + tag])
    for pol, negate in ((3, (False, True)), (22, (False, False))):
        out = aug.AutoAugment()(x, training=True, decision={"policy": pol, "apply": (True, True), "negate": negate})
        np.testing.assert_array_equal(out.cpu().numpy(), ref["autoaugment_p%d_%s" % (pol, tag)])
    oh, ow = gen.INPUT_SIDE_SIZE
    shape = ref["x_" + tag].shape
    ch, cw = shape[1] - 4, shape[2] - 2
    np.testing.assert_array_equal(aug.Resizing(oh, ow)(x).cpu().numpy(), ref["resize_bilinear_" + tag])
    np.testing.assert_array_equal(aug.Resizing(oh, ow, interpolation="nearest")(x).cpu().numpy(), ref["resize_nearest_" + tag])
    np.testing.assert_array_equal(aug.CenterCrop(ch, cw)(x).cpu().numpy(), ref["centercrop_" + tag])
    np.testing.assert_array_equal(aug.RandomCrop(ch, cw)(x, training=True, offset=gen.INPUT_SIDE_OFFSET).cpu().numpy(), ref["randomcrop_" + tag])
    bits = np.arange(shape[0]) % 4
    flipped = aug.RandomFlip()(x, training=True, flip_horizontal=(bits & 1).astype(bool), flip_vertical=(bits >> 1).astype(bool))
    np.testing.assert_array_equal(flipped.cpu().numpy(), ref["flip_" + tag])
    np.testing.assert_array_equal(aug.Rescaling(1.0 / 255.0, offset=-0.5)(x).cpu().numpy(), ref["rescale_" + tag])
    for mode in ("tf", "torch", "caffe"):
        out = aug.ImageNetNormalization(mode)(x).cpu().numpy()
        np.testing.assert_array_equal(out, ref["normalize_%s_%s" % (mode, tag)], err_msg=mode)
        kat = aug.ImageNetNormalization(mode)(_dev(ref["norm_kat_x"])).cpu().numpy()[0, ..., 0]
        np.testing.assert_array_equal(kat, ref["norm_kat_reference_" + mode])     # the reference's own known answers


@pytest.mark.gpu
def test_hip_dropout_mask_matches_golden():
    from chambers_amd import kernels as K
    from chambers_amd import rng
    ref = _load("dropout_mask.npz")
    for i in range(3):
        seed, step, site = (int(v) for v in ref["case%d_params" % i])
        key = rng.site_key(seed, step, site)
        assert key == int(ref["case%d_key" % i][0])
        mask = K.dropout_mask(4096, float(ref["case%d_rate" % i][0]), key).cpu().numpy()
        np.testing.assert_array_equal(np.packbits(mask.astype(np.uint8)), ref["case%d_keep_bits" % i])


@pytest.mark.gpu
def test_hip_vit_step_matches_golden():
    from chambers_amd.engine import ViTConfig, ViTEngine, keras_to_internal
    gen, ref = _gen(), _load("vit_tiny_step.npz")
    c = gen.VIT_CFG
    cfg = ViTConfig(c["patch_size"], c["patch_dim"], c["n_encoder_layers"], c["n_heads"], c["ff_dim"], c["dropout_rate"],
                    image_size=c["image_size"], classes=c["classes"])
    kw = {k[2:]: v for k, v in ref.items() if k.startswith("w/")}
    images, labels = _dev(ref["images"]), _dev(ref["labels"])
    inf = ViTEngine(cfg, 3, training=False)
    inf.load_keras_weights(kw)
    # bf16 operands / fp32 accumulate against the plain fp32 oracle: 2e-2 (the documented bf16-vs-fp32 gap)
    assert rel_l2(inf.forward(images, training=False).cpu().numpy(), ref["logits_inference"]) < 2e-2
    eng = ViTEngine(cfg, 3, training=True, seed=gen.VIT_SEED)
    eng.load_keras_weights(kw)
    logits = eng.forward(images, training=True).cpu().numpy()
    loss = eng.loss(labels).cpu().numpy()
    eng.backward()
    assert rel_l2(logits, ref["logits_training"]) < 2e-2          # same dropout masks as the fixture (counter-hash keys)
    assert rel_l2(loss, ref["loss_per_sample"]) < 2e-2
    grads = eng.export_keras_grads()
    for k in kw:
        if k.endswith("b_key"):
            continue   # exactly zero in exact arithmetic (softmax shift invariance): rounding noise on both sides
        assert rel_l2(grads[k], ref["g/" + k]) < 6e-2, (k, rel_l2(grads[k], ref["g/" + k]))
    # optimizer alone: the fixture's gradients into the flat buffer, one fused AdamW step, against the fixture's new weights
    gi = keras_to_internal({k: ref["g/" + k] for k in kw}, cfg)
    eng.G.zero_()
    for s in eng.specs:
        eng.g(s.name).copy_(torch.as_tensor(np.ascontiguousarray(gi[s.name]), device="cuda"))
    eng.adamw_step(learning_rate=1e-3, weight_decay=0.01)
    new = eng.export_keras_weights()
    for k in kw:
        np.testing.assert_allclose(new[k], ref["w1/" + k], rtol=1e-5, atol=2e-7, err_msg=k)   # same (fixture) gradients in: exact rule
