"""Generates the golden fixtures in this directory from the CPU oracle (oracle/*.py) ONLY.

    python tests/golden/make_golden.py          # rewrites augment_ops.npz, vit_tiny_step.npz, dropout_mask.npz

The reference cannot run in this environment (TensorFlow is not installed - SURVEY.md 8c), so, as that section prescribes,
the fixtures are produced by the oracle; the only reference-derived vectors are the ImageNetNormalization known answers of
/root/reference/test_units/augmentations/test_image_augmentations.py:5-64, which are stored here as data next to the
oracle's output for the same input (tests assert the two agree exactly).  tests/test_golden.py checks (a) that the oracle
still reproduces every file (CPU tier) and (b) that the HIP path matches them (GPU tier: bit-exact for uint8 work)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import augment_ref as A      # noqa: E402
from oracle import rng_ref, vit_ref      # noqa: E402

# ---- the op list with the parameters RandAugment(magnitude 9) / AutoAugment produce, plus edge values ------------------
AUG_CASES = [
    ("invert", "Invert", {}, False), ("posterize3", "Posterize", {"bits": 3}, False), ("posterize0", "Posterize", {"bits": 0}, False),
    ("solarize230", "Solarize", {"threshold": 230}, False), ("solarize256", "Solarize", {"threshold": 256}, False),
    ("solarizeadd99", "SolarizeAdd", {"addition": 99}, False), ("brightness1.72", "Brightness", {"factor": 1.72}, False),
    ("brightness0.3", "Brightness", {"factor": 0.3}, False), ("contrast1.72", "Contrast", {"factor": 1.72}, False),
    ("color1.72", "Color", {"factor": 1.72}, False), ("color0.5", "Color", {"factor": 0.5}, False),
    ("sharpness1.72", "Sharpness", {"factor": 1.72}, False), ("sharpness0.4", "Sharpness", {"factor": 0.4}, False),
    ("autocontrast", "AutoContrast", {}, False), ("equalize", "Equalize", {}, False),
    ("shearx0.27", "ShearX", {"level": 0.27, "fill_value": 128}, False), ("shearx0.27neg", "ShearX", {"level": 0.27, "fill_value": 128}, True),
    ("sheary0.27", "ShearY", {"level": 0.27, "fill_value": 128}, False), ("translatex9neg", "TranslateX", {"pixels": 9.0, "fill_value": 128}, True),
    ("translatey2.5", "TranslateY", {"pixels": 2.5, "fill_value": 0}, False), ("rotate27", "Rotate", {"degrees": 27.0, "fill_value": 128}, False),
    ("rotate27neg", "Rotate", {"degrees": 27.0, "fill_value": 128}, True), ("cutout8", "CutOut", {"mask_size": 8, "constant_values": 128}, False),
]
INPUT_SIDE_SIZE, INPUT_SIDE_OFFSET = (30, 36), (2, 1)
IMAGE_SHAPES = {"a": (2, 20, 24, 3), "b": (3, 9, 10, 3)}    # W % 4 == 0 (vector paths) and a ragged shape (generic paths)
NORM_KAT_IMG = np.array([[139, 186, 208, 200], [175, 201, 198, 200], [166, 191, 193, 195], [124, 155, 172, 151]], dtype=np.uint8)
NORM_KAT = {   # reference-derived (test_image_augmentations.py:21-64), channel 0 of the broadcast grey image
    "caffe": [[35.060997, 82.061, 104.061, 96.061], [71.061, 97.061, 94.061, 96.061], [62.060997, 87.061, 89.061, 91.061],
              [20.060997, 51.060997, 68.061, 47.060997]],
    "tf": [[0.0901961327, 0.458823562, 0.631372571, 0.568627477], [0.372549057, 0.576470613, 0.552941203, 0.568627477],
           [0.301960826, 0.498039246, 0.513725519, 0.529411793], [-0.0274509788, 0.215686321, 0.349019647, 0.184313774]],
    "torch": [[0.262436897, 1.06730032, 1.44404483, 1.30704677], [0.878928, 1.32417154, 1.27279735, 1.30704677],
              [0.724805236, 1.15292406, 1.1871736, 1.22142303], [0.00556548592, 0.536432922, 0.827553749, 0.467933923]],
}
VIT_CFG = {"patch_size": 16, "patch_dim": 64, "n_encoder_layers": 2, "n_heads": 1, "ff_dim": 128, "dropout_rate": 0.1,
           "image_size": (32, 48), "classes": 8, "norm_epsilon": 1e-6, "pooling": "cls"}
VIT_SEED, VIT_STEP = 7, 0


def elementwise_randaugment_decisions(shape):
    """Fixed per-image decisions that walk all 16 ops over the images of the two fixture batches (image n: ops
    (5n + 4) % 16 then (5n + 9) % 16 ... so Contrast, both statistics ops, every warp and CutOut occur)."""
    b, h, w, _ = shape
    table = [(4, 1), (0, 7), (14, 6), (15, 10), (12, 2)]       # Contrast+Equalize, AutoContrast+ShearX, CutOut+Sharpness, Rotate+TranslateY, Solarize+Invert
    return [[{"op": table[(n + k) % 5][j] if k == 0 else (3 * n + 5 * j + 9) % 16, "negate": bool((n + j) & 1),
              "centers": np.array([[(7 * n + 3 * j + 2) % h, (5 * n + j + 1) % w]], dtype=np.int32)} for j in range(2)]
            for n in range(b) for k in (0,)]


def elementwise_autoaugment_decisions(shape):
    b = shape[0]
    pols = [0, 12, 22, 15, 7]
    return [{"policy": pols[n % 5], "apply": (bool((n + 1) & 1), True), "negate": (bool(n & 1), bool((n >> 1) & 1))} for n in range(b)]


def augment_fixture():
    g = np.random.Generator(np.random.PCG64(0))
    out = {}
    for tag, shape in IMAGE_SHAPES.items():
        x = g.integers(0, 256, size=shape, dtype=np.uint8)
        x[0, :3, :, :] = 77                      # a flat band: AutoContrast identity channel rows / Equalize step == 0 paths see ties
        centers = np.stack([g.integers(0, shape[1], size=shape[0]), g.integers(0, shape[2], size=shape[0])], axis=1).astype(np.int32)
        out["x_" + tag], out["centers_" + tag] = x, centers
        for name, op, kw, neg in AUG_CASES:
            out["%s_%s" % (name, tag)] = A.apply_op(x, op, kw, negate=neg, centers=centers)
        dec = [{"op": 7, "negate": True, "centers": centers}, {"op": 14, "negate": False, "centers": centers}]   # ShearX then CutOut
        out["randaugment_%s" % tag] = A.rand_augment(x, 2, 9, dec)
        out["autoaugment_p3_%s" % tag] = A.auto_augment(x, {"policy": 3, "apply": (True, True), "negate": (False, True)})
        out["autoaugment_p22_%s" % tag] = A.auto_augment(x, {"policy": 22, "apply": (True, True), "negate": (False, False)})
        # elementwise=True (tf.map_fn over batch-1 tensors): per-image op / sign / centre; Contrast's constant is H*W/256 of ONE image
        out["randaugment_elementwise_%s" % tag] = A.rand_augment_elementwise(x, 2, 9, elementwise_randaugment_decisions(shape))
        out["autoaugment_elementwise_%s" % tag] = A.auto_augment_elementwise(x, elementwise_autoaugment_decisions(shape))
        for mode in ("tf", "torch", "caffe"):
            out["normalize_%s_%s" % (mode, tag)] = A.imagenet_normalize(x, mode)
        # input side (Resizing / CenterCrop / RandomCrop / RandomFlip / Rescaling)
        oh, ow = INPUT_SIDE_SIZE
        out["resize_bilinear_%s" % tag] = A.resize(x, oh, ow, "bilinear")
        out["resize_nearest_%s" % tag] = A.resize(x, oh, ow, "nearest")
        ch, cw = shape[1] - 4, shape[2] - 2
        out["centercrop_%s" % tag] = A.crop_flip(x, ch, cw, A.center_crop_offsets(shape[1], shape[2], ch, cw))
        out["randomcrop_%s" % tag] = A.crop_flip(x, ch, cw, INPUT_SIDE_OFFSET)
        out["flip_%s" % tag] = A.crop_flip(x, shape[1], shape[2], (0, 0), np.arange(shape[0], dtype=np.uint8) % 4)
        out["rescale_%s" % tag] = A.rescale(x, 1.0 / 255.0, -0.5)
    kat = np.stack([NORM_KAT_IMG] * 3, axis=-1)[None]
    out["norm_kat_x"] = kat
    for mode in ("tf", "torch", "caffe"):
        out["norm_kat_reference_" + mode] = np.array(NORM_KAT[mode], dtype=np.float32)
        out["norm_kat_oracle_" + mode] = A.imagenet_normalize(kat, mode)[0, ..., 0]
    return out


def vit_weights(cfg, g):
    """Deterministic weights with the reference's shapes / names (values: scaled normals, non-trivial biases and LayerNorm)."""
    d, h, ff, p = cfg["patch_dim"], cfg["n_heads"], cfg["ff_dim"], cfg["patch_size"]
    hd = d // h
    n_tok = (cfg["image_size"][0] // p) * (cfg["image_size"][1] // p) + 1
    f = lambda *s, sc=0.05: (g.normal(0, sc, size=s)).astype(np.float32)   # noqa: E731
    kw = {"patch_embeddings/embedding/kernel": f(p, p, 3, d, sc=0.03), "patch_embeddings/embedding/bias": f(d),
          "add_cls_token/embeddings": f(1, d, sc=0.02), "pos_embedding/embeddings": f(n_tok, d, sc=0.02)}
    for i in range(cfg["n_encoder_layers"]):
        pre = "encoder/layer_%d/" % i
        a = pre + "multi_head_attention/"
        for nm in ("query", "value", "key"):
            kw[a + "w_" + nm], kw[a + "b_" + nm] = f(d, h, hd, sc=0.08), f(h, 1, hd)
        kw[a + "w_projection"], kw[a + "b_projection"] = f(h, d, hd, sc=0.08), f(1, d)
        kw[pre + "norm1/gamma"], kw[pre + "norm1/beta"] = (1 + f(d, sc=0.1)).astype(np.float32), f(d)
        kw[pre + "dense1/kernel"], kw[pre + "dense1/bias"] = f(d, ff, sc=0.08), f(ff)
        kw[pre + "dense2/kernel"], kw[pre + "dense2/bias"] = f(ff, d, sc=0.08), f(d)
        kw[pre + "norm2/gamma"], kw[pre + "norm2/beta"] = (1 + f(d, sc=0.1)).astype(np.float32), f(d)
    kw["encoder/norm/gamma"], kw["encoder/norm/beta"] = (1 + f(d, sc=0.1)).astype(np.float32), f(d)
    kw["predictions/kernel"], kw["predictions/bias"] = f(d, cfg["classes"], sc=0.1), f(cfg["classes"])
    return kw


def vit_fixture():
    torch.set_num_threads(1)
    cfg = VIT_CFG
    g = np.random.Generator(np.random.PCG64(1))
    kw = vit_weights(cfg, g)
    images = g.integers(0, 256, size=(3,) + cfg["image_size"] + (3,), dtype=np.uint8)
    labels = g.integers(0, cfg["classes"], size=(3,)).astype(np.int64)
    x = torch.from_numpy(A.imagenet_normalize(images, "tf"))
    n_sites = 1 + 3 * cfg["n_encoder_layers"]
    keys = {s: rng_ref.site_key(VIT_SEED, VIT_STEP, s) for s in range(n_sites)}
    out = {"images": images, "labels": labels}
    out.update({"w/" + k: v for k, v in kw.items()})
    p = {k: torch.tensor(v) for k, v in kw.items()}
    out["logits_inference"] = vit_ref.vit_forward(p, x, cfg, keys=None, bf16=False).numpy()
    p = {k: torch.tensor(v, requires_grad=True) for k, v in kw.items()}
    logits = vit_ref.vit_forward(p, x, cfg, keys=keys, bf16=False)
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(labels), reduction="none")
    loss.mean().backward()
    out["logits_training"], out["loss_per_sample"] = logits.detach().numpy(), loss.detach().numpy()
    grads = {k: p[k].grad.clone() for k in kw}
    out.update({"g/" + k: v.numpy() for k, v in grads.items()})
    pw = {k: torch.tensor(v) for k, v in kw.items()}
    m = {k: torch.zeros_like(v) for k, v in pw.items()}
    v_ = {k: torch.zeros_like(v) for k, v in pw.items()}
    vit_ref.adamw_step(pw, grads, m, v_, 1, lr=1e-3, weight_decay=0.01)
    out.update({"w1/" + k: v.numpy() for k, v in pw.items()})
    return out


def dropout_fixture():
    out = {}
    for i, (seed, step, site, rate) in enumerate(((0, 0, 0, 0.1), (7, 3, 11, 0.1), (2 ** 40 + 5, 1000, 36, 0.5))):
        key = rng_ref.site_key(seed, step, site)
        out["case%d_key" % i] = np.array([key], dtype=np.uint64)
        out["case%d_params" % i] = np.array([seed, step, site], dtype=np.int64)
        out["case%d_rate" % i] = np.array([rate], dtype=np.float32)
        out["case%d_keep_bits" % i] = np.packbits(rng_ref.keep_mask(4096, key, rate).astype(np.uint8))
    return out


FILES = {"augment_ops.npz": augment_fixture, "vit_tiny_step.npz": vit_fixture, "dropout_mask.npz": dropout_fixture}

if __name__ == "__main__":
    for name, fn in FILES.items():
        data = fn()
        np.savez_compressed(os.path.join(HERE, name), **data)
        print("%-22s %4d arrays  %7.1f KiB" % (name, len(data), os.path.getsize(os.path.join(HERE, name)) / 1024))
