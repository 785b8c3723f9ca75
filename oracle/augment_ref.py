"""CPU oracle for the chambers augmentation hot path (TEST INFRASTRUCTURE ONLY).

This file is a NumPy restatement of the reference algorithm, written from the
reference *text* (TensorFlow / tensorflow-addons are not installed here, so the
reference itself cannot run).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product path
(``chambers_amd``) never does.

PARITY STATUS: **parity unpinned** for everything except ``imagenet_normalize``.
The reference's own tests hold known-answer vectors only for
``ImageNetNormalization`` (test_units/augmentations/test_image_augmentations.py:21-64);
those three vectors are checked in ``tests/test_oracle_kat.py``.  The 16
augmentation ops delegate their arithmetic to tensorflow==2.6.0 (pinned,
requirements.txt:2) and tensorflow-addons (unpinned, requirements.txt:3), whose
sources are not under /root/reference; their published algorithms are restated
below and each such function says so ("upstream restated").
Independent anchors (not the reference): tests/test_oracle_independent.py holds this file against Pillow - the implementation the
AutoAugment ops were defined through - bit for bit for Invert, Posterize, Solarize, Equalize, AutoContrast and Brightness, within
one grey level for Color and Sharpness, and against scipy.ndimage / torch for the warps and the resizes.

Every random decision (sign flips, chance draws, op choice, cutout centres) is an
explicit argument: the reference draws them from TF's stateful global RNG
(image_augmentations.py:54,523,608), which cannot be reproduced outside TF.

All images are uint8 NHWC ``[B, H, W, C]``.  Float arithmetic is done in
``np.float32`` one operation at a time (no fused multiply-add), which is what
the un-fused TF CPU kernels do.
"""
import math

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #
def _trunc_u8(x):
    """tf.cast(float32 -> uint8) for in-range values: truncation toward zero."""
    return x.astype(np.int32).astype(np.uint8)


def blend(image1, image2, factor):
    """image_augmentations.py:10-49."""
    if factor == 0.0:
        return np.array(image1, copy=True)
    if factor == 1.0:
        return np.array(image2, copy=True)
    i1 = image1.astype(F32)
    i2 = image2.astype(F32)
    difference = i2 - i1
    scaled = F32(factor) * difference
    temp = i1 + scaled
    if 0.0 < factor < 1.0:
        return _trunc_u8(temp)
    return _trunc_u8(np.clip(temp, F32(0.0), F32(255.0)))


def rgb_to_grayscale(images):
    """tf.image.rgb_to_grayscale on uint8 (upstream restated, TF 2.6
    python/ops/image_ops_impl.py): convert_image_dtype(uint8->f32) multiplies by
    1/255, tensordot with (0.2989, 0.5870, 0.1140), convert back multiplies by
    255.5 and truncates.  Summation order r,g,b left to right, no FMA."""
    flt = images.astype(F32) * F32(1.0 / 255.0)
    g = flt[..., 0] * F32(0.2989)
    g = g + flt[..., 1] * F32(0.5870)
    g = g + flt[..., 2] * F32(0.1140)
    g = g * F32(255.5)
    return _trunc_u8(g)[..., None]


# --------------------------------------------------------------------------- #
# integer / pointwise ops
# --------------------------------------------------------------------------- #
def invert(x):
    """Invert.call, image_augmentations.py:112-113."""
    return (255 - x.astype(np.int32)).astype(np.uint8)


def posterize(x, bits):
    """Posterize.call, image_augmentations.py:168,171-174.  shift = 8 - bits; TF's
    CPU shift functors clamp the shift count to bit-width-1 (upstream restated),
    which is reached by AutoAugment sub-policy 22 (bits=0 -> shift 8 -> 7)."""
    shift = min(max(8 - int(bits), 0), 7)
    return ((x >> shift) << shift).astype(np.uint8)


def solarize(x, threshold=128):
    """Solarize.call, image_augmentations.py:192-193.  threshold may be 256
    (augmentation_schemes.py:76 at magnitude 10): compared as integers, i.e.
    'always below' => identity."""
    xi = x.astype(np.int32)
    return np.where(xi < int(threshold), xi, 255 - xi).astype(np.uint8)


def solarize_add(x, addition=0, threshold=128):
    """SolarizeAdd.call, image_augmentations.py:212-215."""
    xi = x.astype(np.int64)
    added = np.clip(xi + int(addition), 0, 255)
    return np.where(xi < int(threshold), added, xi).astype(np.uint8)


def brightness(x, factor):
    """Brightness.call, image_augmentations.py:283-285."""
    return blend(np.zeros_like(x), x, factor)


def color(x, factor):
    """Color.call, image_augmentations.py:233-235."""
    degenerate = np.repeat(rgb_to_grayscale(x), 3, axis=-1)
    return blend(degenerate, x, factor)


def contrast_constant(n_pixels):
    """Contrast.call, image_augmentations.py:254-264: the 'mean' is
    sum(histogram)/256 == (number of grey pixels in the WHOLE tensor)/256,
    clipped to [0,255] and truncated (a quirk of the reference kept as is)."""
    mean = F32(n_pixels) / F32(256.0)
    mean = min(max(mean, F32(0.0)), F32(255.0))
    return int(mean)


def contrast(x, factor):
    """Contrast.call, image_augmentations.py:253-265 (x is the whole batch)."""
    b, h, w, _ = x.shape
    degenerate = np.full_like(x, contrast_constant(b * h * w))
    return blend(degenerate, x, factor)


def autocontrast(x):
    """AutoContrast.call, image_augmentations.py:68-87 (per image, per channel)."""
    lo = x.min(axis=(1, 2)).astype(F32)
    hi = x.max(axis=(1, 2)).astype(F32)
    rng = hi - lo
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = np.where(rng != 0, F32(255.0) / rng, F32(0.0)).astype(F32)
    offset = (-lo) * scale
    mask = (hi > lo).astype(F32)
    scale = scale * mask + (F32(1.0) - mask)
    offset = offset * mask
    y = x.astype(F32) * scale[:, None, None, :]
    y = y + offset[:, None, None, :]
    y = np.clip(y, F32(0.0), F32(255.0))
    return _trunc_u8(y)


def equalize(x):
    """Equalize.call -> tfa.image.equalize (upstream restated, tensorflow_addons
    image/color_ops.py `_scale_channel`): per image and channel, 256-bin
    histogram; step = (sum(nonzero bins) - last nonzero bin) // 255; identity if
    step == 0 else lut = clip((exclusive_cumsum(hist) + step//2) // step, 0, 255)."""
    out = np.empty_like(x)
    b, _, _, c = x.shape
    for n in range(b):
        for ch in range(c):
            plane = x[n, :, :, ch]
            hist = np.bincount(plane.ravel(), minlength=256).astype(np.int64)
            nz = hist[hist != 0]
            step = (nz.sum() - nz[-1]) // 255
            if step == 0:
                out[n, :, :, ch] = plane
            else:
                cum = np.cumsum(hist) - hist
                lut = np.clip((cum + step // 2) // step, 0, 255).astype(np.uint8)
                out[n, :, :, ch] = lut[plane]
    return out


def sharpness(x, factor, final_cast="truncate"):
    """Sharpness.call -> tfa.image.sharpness (upstream restated,
    tensorflow_addons image/color_ops.py `sharpness_image`): depthwise
    3x3 [[1,1,1],[1,5,1],[1,1,1]]/13 VALID in f32 (accumulated row-major, no
    FMA), truncated to uint8; the 1-pixel border keeps the original; then
    blend(degenerate, original, factor) with clip and truncating cast.

    OPEN CHOICE (parity unpinned, tensorflow-addons is not pinned in requirements.txt:3): the final blend goes through TFA's own
    `blend` (tensorflow_addons/image/compose_ops.py), not chambers' (image_augmentations.py:10-49).  SURVEY 8a row 24 restates
    it with a truncating cast, which is what this oracle and the HIP kernel implement (``final_cast="truncate"``); another
    recollection of that file has `tf.round` in front of the cast.  ``final_cast="round"`` evaluates that alternative so the
    known-answer test (tests/test_oracle_kat.py::test_sharpness_final_cast_is_an_open_choice) shows where the two differ."""
    b, h, w, c = x.shape
    xf = x.astype(F32)
    k = (np.array([[1, 1, 1], [1, 5, 1], [1, 1, 1]], dtype=F32) / F32(13.0)).astype(F32)
    result = np.array(x, copy=True)
    if h >= 3 and w >= 3:
        acc = np.zeros((b, h - 2, w - 2, c), dtype=F32)
        for ky in range(3):
            for kx in range(3):
                acc = acc + xf[:, ky:ky + h - 2, kx:kx + w - 2, :] * k[ky, kx]
        result[:, 1:h - 1, 1:w - 1, :] = _trunc_u8(acc)
    if factor == 0.0:
        return result
    if factor == 1.0:
        return np.array(x, copy=True)
    i1 = result.astype(F32)
    temp = i1 + F32(factor) * (xf - i1)
    if not (0.0 <= factor <= 1.0):
        temp = np.clip(temp, F32(0.0), F32(255.0))
    if final_cast == "round":
        temp = np.rint(temp)        # tf.round: half to even
    elif final_cast != "truncate":
        raise ValueError(final_cast)
    return _trunc_u8(temp)


# --------------------------------------------------------------------------- #
# geometric ops (tfa.image.transform == ImageProjectiveTransformV3, upstream restated)
# --------------------------------------------------------------------------- #
def _round_half_away(v):
    """std::round on float32."""
    t = np.trunc(v)
    frac = v - t  # exact in float32
    return np.where(np.abs(frac) >= F32(0.5), t + np.sign(v), t).astype(F32)


def projective_transform(x, transforms, fill_value=0):
    """tfa.image.transform(interpolation='nearest', fill_mode='constant')
    (upstream restated, tensorflow/core/kernels/image/image_ops.h
    ProjectiveGenerator): for OUTPUT pixel (ox, oy), in f32 without FMA,
        proj = c0*ox + c1*oy + 1;  proj == 0 -> fill
        ix = (a0*ox + a1*oy + a2) / proj ; iy = (b0*ox + b1*oy + b2) / proj
        read input[round(iy), round(ix)] (half away from zero) or fill if outside.
    ``transforms`` is [8] (shared by the batch) or [B, 8]."""
    b, h, w, c = x.shape
    t = np.asarray(transforms, dtype=F32).reshape(-1, 8)
    if t.shape[0] == 1:
        t = np.repeat(t, b, axis=0)
    ox = np.arange(w, dtype=F32)[None, :]
    oy = np.arange(h, dtype=F32)[:, None]
    out = np.empty_like(x)
    fill = np.uint8(int(fill_value))
    for n in range(b):
        a0, a1, a2, b0, b1, b2, c0, c1 = (F32(v) for v in t[n])
        proj = (c0 * ox + c1 * oy) + F32(1.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            ix = ((a0 * ox + a1 * oy) + a2) / proj
            iy = ((b0 * ox + b1 * oy) + b2) / proj
        rx = _round_half_away(ix)
        ry = _round_half_away(iy)
        ok = (proj != 0) & (rx >= 0) & (rx < w) & (ry >= 0) & (ry < h)
        rxi = np.where(ok, rx, 0).astype(np.int64)
        ryi = np.where(ok, ry, 0).astype(np.int64)
        gathered = x[n][ryi, rxi, :]
        out[n] = np.where(ok[..., None], gathered, fill)
    return out


def shear_x_transform(level, negate):
    """ShearX.call, image_augmentations.py:333-341 (sign: :52-56)."""
    lv = -level if negate else level
    return np.array([1.0, lv, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], dtype=F32)


def shear_y_transform(level, negate):
    """ShearY.call, image_augmentations.py:376-384."""
    lv = -level if negate else level
    return np.array([1.0, 0.0, 0.0, lv, 1.0, 0.0, 0.0, 0.0], dtype=F32)


def translate_x_transform(pixels, negate):
    """TranslateX.call :419-427 -> tfa.image.translate([-px, 0]) ->
    translations_to_projective_transforms: [1,0,-dx,0,1,-dy,0,0] (upstream restated)."""
    px = -pixels if negate else pixels
    dx = -px
    return np.array([1.0, 0.0, -dx, 0.0, 1.0, -0.0, 0.0, 0.0], dtype=F32)


def translate_y_transform(pixels, negate):
    """TranslateY.call :462-470."""
    px = -pixels if negate else pixels
    dy = -px
    return np.array([1.0, 0.0, -0.0, 0.0, 1.0, -dy, 0.0, 0.0], dtype=F32)


def rotate_transform(degrees, negate, height, width):
    """Rotate.call :138-146 -> tfa.image.rotate -> angles_to_projective_transforms
    (upstream restated), all in f32."""
    radians = degrees * math.pi / 180.0
    if negate:
        radians = -radians
    a = F32(radians)
    cos = F32(np.cos(a))
    sin = F32(np.sin(a))
    wm1 = F32(width - 1)
    hm1 = F32(height - 1)
    x_off = (wm1 - (cos * wm1 - sin * hm1)) / F32(2.0)
    y_off = (hm1 - (sin * wm1 + cos * hm1)) / F32(2.0)
    return np.array([cos, -sin, x_off, sin, cos, y_off, 0.0, 0.0], dtype=F32)


def cutout(x, mask_size, centers, constant_values=0):
    """CutOut.call :495-499 -> tfa.image.random_cutout/cutout (upstream restated):
    per image rows [cy - m/2, cy + m/2) and cols [cx - m/2, cx + m/2) clipped to
    the image are set to constant_values.  ``centers`` is int [B, 2] = (cy, cx)."""
    if int(mask_size) % 2 != 0:
        raise ValueError("mask_size should be divisible by 2")
    b, h, w, _ = x.shape
    half = int(mask_size) // 2
    out = np.array(x, copy=True)
    centers = np.asarray(centers).reshape(b, 2)
    for n in range(b):
        cy, cx = int(centers[n, 0]), int(centers[n, 1])
        y0, y1 = max(0, cy - half), min(h, cy + half)
        x0, x1 = max(0, cx - half), min(w, cx + half)
        if y1 > y0 and x1 > x0:
            out[n, y0:y1, x0:x1, :] = np.uint8(int(constant_values))
    return out


# --------------------------------------------------------------------------- #
# normalisation (the only reference-pinned function)
# --------------------------------------------------------------------------- #
def imagenet_normalize(x, mode="caffe"):
    """ImageNetNormalization.call, image_augmentations.py:629-682.
    Pinned by test_units/augmentations/test_image_augmentations.py:21-64."""
    if mode not in {"caffe", "tf", "torch"}:
        raise ValueError("Unknown mode " + str(mode))
    if mode == "tf":
        y = x.astype(F32)
        y = y / F32(127.5)
        return y - F32(1.0)
    if mode == "torch":
        y = x.astype(F32) / F32(255.0)
        mean = np.array([0.485, 0.456, 0.406], dtype=F32)
        std = np.array([0.229, 0.224, 0.225], dtype=F32)
        return (y - mean) / std
    y = x[..., ::-1].astype(F32)
    mean = np.array([103.939, 116.779, 123.68], dtype=F32)
    return y - mean


# --------------------------------------------------------------------------- #
# schemes (augmentation_schemes.py)
# --------------------------------------------------------------------------- #
FILL_VALUE = 128           # augmentation_schemes.py:9
MAX_MAGNITUDE = 10.0       # augmentation_schemes.py:10

AUTO_AUGMENT_POLICY_V0 = [  # augmentation_schemes.py:12-39
    [("Equalize", 0.8, None), ("ShearY", 0.8, 4)],
    [("Color", 0.4, 9), ("Equalize", 0.6, None)],
    [("Color", 0.4, 1), ("Rotate", 0.6, 8)],
    [("Solarize", 0.8, 3), ("Equalize", 0.4, 7)],
    [("Solarize", 0.4, 2), ("Solarize", 0.6, 2)],
    [("Color", 0.2, 0), ("Equalize", 0.8, None)],
    [("Equalize", 0.4, None), ("SolarizeAdd", 0.8, 3)],
    [("ShearX", 0.2, 9), ("Rotate", 0.6, 8)],
    [("Color", 0.6, 1), ("Equalize", 1.0, None)],
    [("Invert", 0.4, None), ("Rotate", 0.6, 0)],
    [("Equalize", 1.0, None), ("ShearY", 0.6, 3)],
    [("Color", 0.4, 7), ("Equalize", 0.6, None)],
    [("Posterize", 0.4, 6), ("AutoContrast", 0.4, None)],
    [("Solarize", 0.6, 8), ("Color", 0.6, 9)],
    [("Solarize", 0.2, 4), ("Rotate", 0.8, 9)],
    [("Rotate", 1.0, 7), ("TranslateY", 0.8, 9)],
    [("ShearX", 0.0, 0), ("Solarize", 0.8, 4)],
    [("ShearY", 0.8, 0), ("Color", 0.6, 4)],
    [("Color", 1.0, 0), ("Rotate", 0.6, 2)],
    [("Equalize", 0.8, None), ("Equalize", 0.0, None)],
    [("Equalize", 1.0, None), ("AutoContrast", 0.6, None)],
    [("ShearY", 0.4, 7), ("SolarizeAdd", 0.6, 7)],
    [("Posterize", 0.8, 2), ("Solarize", 0.6, 10)],
    [("Solarize", 0.6, 8), ("Equalize", 0.6, 1)],
    [("Color", 0.8, 6), ("Rotate", 0.4, 5)],
]

RANDAUGMENT_OPS = [  # augmentation_schemes.py:181-198
    "AutoContrast", "Equalize", "Invert", "Brightness", "Contrast", "Color",
    "Sharpness", "ShearX", "ShearY", "TranslateX", "TranslateY", "Posterize",
    "Solarize", "SolarizeAdd", "CutOut", "Rotate",
]


def magnitude_to_kwargs(name, magnitude):
    """augmentation_schemes.py:42-128."""
    if name in ("AutoContrast", "Equalize", "Invert"):
        return {}
    m = magnitude / MAX_MAGNITUDE
    if name in ("Brightness", "Contrast", "Color", "Sharpness"):
        return {"factor": m * 1.8 + 0.1}
    if name in ("ShearX", "ShearY"):
        return {"level": m * 0.3, "fill_value": FILL_VALUE}
    if name in ("TranslateX", "TranslateY"):
        return {"pixels": m * 100, "fill_value": FILL_VALUE}
    if name == "Posterize":
        return {"bits": int(m * 4)}
    if name == "Solarize":
        return {"threshold": int(m * 256)}
    if name == "SolarizeAdd":
        return {"addition": int(m * 110)}
    if name == "Rotate":
        return {"degrees": m * 30.0, "fill_value": FILL_VALUE}
    if name == "CutOut":
        return {"mask_size": int(m * 80), "constant_values": FILL_VALUE}
    raise KeyError(name)


def apply_op(x, name, kwargs, negate=False, centers=None):
    """One primitive op on a whole batch with explicit random decisions."""
    _, h, w, _ = x.shape
    if name == "AutoContrast":
        return autocontrast(x)
    if name == "Equalize":
        return equalize(x)
    if name == "Invert":
        return invert(x)
    if name == "Brightness":
        return brightness(x, kwargs["factor"])
    if name == "Contrast":
        return contrast(x, kwargs["factor"])
    if name == "Color":
        return color(x, kwargs["factor"])
    if name == "Sharpness":
        return sharpness(x, kwargs["factor"])
    if name == "Posterize":
        return posterize(x, kwargs["bits"])
    if name == "Solarize":
        return solarize(x, kwargs.get("threshold", 128))
    if name == "SolarizeAdd":
        return solarize_add(x, kwargs.get("addition", 0), kwargs.get("threshold", 128))
    if name == "CutOut":
        return cutout(x, kwargs["mask_size"], centers, kwargs.get("constant_values", 0))
    fill = kwargs.get("fill_value", 0.0)
    if name == "ShearX":
        t = shear_x_transform(kwargs["level"], negate)
    elif name == "ShearY":
        t = shear_y_transform(kwargs["level"], negate)
    elif name == "TranslateX":
        t = translate_x_transform(kwargs["pixels"], negate)
    elif name == "TranslateY":
        t = translate_y_transform(kwargs["pixels"], negate)
    elif name == "Rotate":
        t = rotate_transform(kwargs["degrees"], negate, h, w)
    else:
        raise KeyError(name)
    return projective_transform(x, t, fill)


def rand_augment(x, n_transforms, magnitude, decisions):
    """RandAugment.call + RandomChoice._random_transforms (augmentation_schemes.py:
    204-213, image_augmentations.py:606-617), batch-shared decisions.
    ``decisions`` = list of n dicts {"op": idx, "negate": bool, "centers": [B,2]}."""
    for i in range(n_transforms):
        d = decisions[i]
        name = RANDAUGMENT_OPS[d["op"]]
        x = apply_op(x, name, magnitude_to_kwargs(name, magnitude),
                     negate=d.get("negate", False), centers=d.get("centers"))
    return x


def auto_augment(x, decision):
    """AutoAugment.call (augmentation_schemes.py:132-160): one of 25 sub-policies,
    each two RandomChance(op, p) (image_augmentations.py:522-529).
    ``decision`` = {"policy": idx, "apply": (bool, bool), "negate": (bool, bool)}."""
    sub = AUTO_AUGMENT_POLICY_V0[decision["policy"]]
    for j, (name, _p, mag) in enumerate(sub):
        if decision["apply"][j]:
            x = apply_op(x, name, magnitude_to_kwargs(name, mag),
                         negate=decision["negate"][j])
    return x


def rand_augment_elementwise(x, n_transforms, magnitude, decisions):
    """RandAugment(elementwise=True): RandomChoice.call maps `_random_transforms` over the batch with tf.map_fn, each image
    expanded to a batch-1 tensor (image_augmentations.py:563-570).  So every image has its own op indices, its own sign
    draws and cutout centre, and Contrast's degenerate constant is that of ONE image (H*W/256 clipped, :260-262).
    ``decisions[n]`` = the list of n_transforms dicts for image n ("centers": [1,2] or [2])."""
    outs = []
    for n in range(x.shape[0]):
        ds = [dict(d, centers=np.asarray(d["centers"]).reshape(1, 2)) if d.get("centers") is not None else d for d in decisions[n]]
        outs.append(rand_augment(x[n:n + 1], n_transforms, magnitude, ds))
    return np.concatenate(outs, axis=0) if outs else np.array(x, copy=True)


def auto_augment_elementwise(x, decisions):
    """AutoAugment(elementwise=True) (augmentation_schemes.py:135,147-149): one decision (sub-policy, two chance draws, two
    sign draws) per image, each image a batch-1 tensor."""
    outs = [auto_augment(x[n:n + 1], decisions[n]) for n in range(x.shape[0])]
    return np.concatenate(outs, axis=0) if outs else np.array(x, copy=True)


# --------------------------------------------------------------------------- #
# input side: keras preprocessing layers re-exported by chambers.augmentations
# (augmentations/__init__.py:1-13) and chambers' ResizingMinMax (:686-748).
# tf.image.resize / Keras layer behaviour is upstream restated (TF 2.6).
# --------------------------------------------------------------------------- #
def resize(x, out_h, out_w, method="bilinear"):
    """tf.image.resize(images, [out_h, out_w], method) with TF2 semantics (half-pixel centres, antialias=False).
    bilinear -> float32, arithmetic order of the CPU kernel (compute_interpolation_weights + compute_lerp);
    nearest -> input dtype, index min(floor((o + 0.5) * scale), size - 1)."""
    b, h, w, c = x.shape
    sy, sx = F32(h) / F32(out_h), F32(w) / F32(out_w)
    oy = np.arange(out_h, dtype=F32)
    ox = np.arange(out_w, dtype=F32)
    if method == "nearest":
        iy = np.minimum(np.floor((oy + F32(0.5)) * sy).astype(np.int64), h - 1)
        ix = np.minimum(np.floor((ox + F32(0.5)) * sx).astype(np.int64), w - 1)
        return np.ascontiguousarray(x[:, iy][:, :, ix])
    if method != "bilinear":
        raise ValueError("unsupported interpolation %r" % (method,))
    fy = ((oy + F32(0.5)) * sy - F32(0.5)).astype(F32)
    fx = ((ox + F32(0.5)) * sx - F32(0.5)).astype(F32)
    fy0, fx0 = np.floor(fy), np.floor(fx)
    y0 = np.maximum(fy0.astype(np.int64), 0)
    y1 = np.minimum(np.ceil(fy).astype(np.int64), h - 1)
    x0 = np.maximum(fx0.astype(np.int64), 0)
    x1 = np.minimum(np.ceil(fx).astype(np.int64), w - 1)
    ty = (fy - fy0).astype(F32)[None, :, None, None]
    tx = (fx - fx0).astype(F32)[None, None, :, None]
    xf = x.astype(F32)
    tl, tr = xf[:, y0][:, :, x0], xf[:, y0][:, :, x1]
    bl, br = xf[:, y1][:, :, x0], xf[:, y1][:, :, x1]
    top = (tl + ((tr - tl) * tx).astype(F32)).astype(F32)
    bot = (bl + ((br - bl) * tx).astype(F32)).astype(F32)
    return (top + ((bot - top) * ty).astype(F32)).astype(F32)


def resizing_minmax_size(height, width, min_side=None, max_side=None):
    """ResizingMinMax.call size arithmetic, image_augmentations.py:712-731 (float32, truncating cast)."""
    if min_side is None and max_side is None:
        raise ValueError("Must specify either 'min_side' or 'max_side'.")
    h, w = F32(height), F32(width)
    if min_side is not None and max_side is not None:
        scale = np.minimum(F32(max_side) / np.maximum(w, h), F32(min_side) / np.minimum(w, h))
    elif min_side is not None:
        scale = F32(min_side) / np.minimum(w, h)
    else:
        scale = F32(max_side) / np.maximum(w, h)
    return int(F32(h * F32(scale))), int(F32(w * F32(scale)))


def crop_flip(x, out_h, out_w, offsets, flips=None):
    """CenterCrop / RandomCrop window + RandomFlip: offsets int [B,2] or [2] = (y, x); flips uint8 [B], bit 0 left-right,
    bit 1 up-down (applied inside the window)."""
    b = x.shape[0]
    offsets = np.broadcast_to(np.asarray(offsets, dtype=np.int64).reshape(-1, 2), (b, 2))
    out = np.empty((b, out_h, out_w) + x.shape[3:], dtype=x.dtype)
    for n in range(b):
        win = x[n, offsets[n, 0]:offsets[n, 0] + out_h, offsets[n, 1]:offsets[n, 1] + out_w]
        f = 0 if flips is None else int(flips[n])
        if f & 2:
            win = win[::-1]
        if f & 1:
            win = win[:, ::-1]
        out[n] = win
    return out


def center_crop_offsets(height, width, out_h, out_w):
    """keras CenterCrop: start = int((size - target) / 2)."""
    return int((height - out_h) / 2), int((width - out_w) / 2)


def rescale(x, scale, offset=0.0):
    """keras Rescaling: cast(x, float32) * scale + offset."""
    return ((x.astype(F32) * F32(scale)).astype(F32) + F32(offset)).astype(F32)
