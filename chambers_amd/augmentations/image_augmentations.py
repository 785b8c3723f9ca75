"""chambers.augmentations.image_augmentations on MI355X.

Same class names, constructor arguments and get_config() keys as the reference
(/root/reference/chambers/augmentations/image_augmentations.py, lines cited per class); the
arithmetic runs in hand-written HIP kernels (chambers_amd/csrc/augment.hip) behind the C ABI.
Inputs are uint8 NHWC torch tensors on the GPU.

Random decisions, which the reference draws from TF's stateful RNG inside `call`, are explicit:
each `call` accepts them as keyword arguments and, when they are omitted, draws them from the
host generator `chambers_amd.rng` (batch-shared, like the reference's non-elementwise mode).
"""
import math

import numpy as np
import torch

from .. import _lib
from .. import kernels as K
from .. import rng
from .._keras_like import InputSpec, Layer, deserialize, register_keras_serializable, serialize


def _randomly_negate_value(value, negate=None):
    """image_augmentations.py:52-56 — one draw per call."""
    if negate is None:
        negate = bool(rng.host_generator().uniform() < 0.5)
    return -value if negate else value


def _base_cfg(layer, config):
    base = Layer.get_config(layer)
    return dict(list(base.items()) + list(config.items()))


# ---------------------------------------------------------------- ops used by AutoAugment / RandAugment
@register_keras_serializable(package="Chambers")
class AutoContrast(Layer):
    """:63-90."""

    def __init__(self, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        return K.aug_autocontrast(inputs)

    def dispatch_item(self, height, width, **kwargs):
        return K.aug_item(_lib.AUG_AUTOCONTRAST)


@register_keras_serializable(package="Chambers")
class Equalize(Layer):
    """:94-103 (tfa.image.equalize)."""

    def __init__(self, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        return K.aug_equalize(inputs)

    def dispatch_item(self, height, width, **kwargs):
        return K.aug_item(_lib.AUG_EQUALIZE)


@register_keras_serializable(package="Chambers")
class Invert(Layer):
    """:107-116."""

    def __init__(self, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        return K.aug_pointwise(inputs, K.PW_INVERT)

    def dispatch_item(self, height, width, **kwargs):
        return K.aug_item(_lib.AUG_INVERT)


class _Warp(Layer):
    """Shared body of the five tfa.image.transform-backed ops."""

    def __init__(self, interpolation="nearest", fill_mode="constant", fill_value=0.0, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.interpolation = interpolation
        self.fill_mode = fill_mode
        self.fill_value = fill_value
        self.input_spec = InputSpec(ndim=4)

    def _warp(self, inputs, transform):
        if self.interpolation != "nearest" or self.fill_mode != "constant":
            raise ValueError("the MI355X warp kernel implements interpolation='nearest', fill_mode='constant' "
                             "(the only combination the augmentation schemes use, augmentation_schemes.py:7-9)")
        return K.aug_affine(inputs, transform, fill=int(self.fill_value))

    def _item(self, transform):
        if self.interpolation != "nearest" or self.fill_mode != "constant":
            raise ValueError("the MI355X warp kernel implements interpolation='nearest', fill_mode='constant'")
        t = np.asarray(transform, dtype=np.float32)
        return K.aug_item(_lib.AUG_AFFINE, i=(int(self.fill_value),), f=t[:6])

    def _warp_cfg(self):
        return {"interpolation": self.interpolation, "fill_mode": self.fill_mode, "fill_value": self.fill_value}


@register_keras_serializable(package="Chambers")
class Rotate(_Warp):
    """:120-160 (tfa.image.rotate)."""

    def __init__(self, degrees, interpolation="nearest", fill_mode="constant", fill_value=0.0, name=None, **kwargs):
        super().__init__(interpolation, fill_mode, fill_value, name=name, **kwargs)
        self.degrees = degrees
        self._radians = degrees * math.pi / 180.0

    @staticmethod
    def transform_for(radians, height, width):
        """tfa angles_to_projective_transforms, float32."""
        a = np.float32(radians)
        cos, sin = np.float32(np.cos(a)), np.float32(np.sin(a))
        wm1, hm1 = np.float32(width - 1), np.float32(height - 1)
        x_off = (wm1 - (cos * wm1 - sin * hm1)) / np.float32(2.0)
        y_off = (hm1 - (sin * wm1 + cos * hm1)) / np.float32(2.0)
        return np.array([cos, -sin, x_off, sin, cos, y_off, 0.0, 0.0], dtype=np.float32)

    def call(self, inputs, negate=None, **kwargs):
        radians = _randomly_negate_value(self._radians, negate)
        return self._warp(inputs, self.transform_for(radians, inputs.shape[1], inputs.shape[2]))

    def dispatch_item(self, height, width, negate=None, **kwargs):
        return self._item(self.transform_for(_randomly_negate_value(self._radians, negate), height, width))

    def get_config(self):
        return _base_cfg(self, dict({"degrees": self.degrees}, **self._warp_cfg()))


@register_keras_serializable(package="Chambers")
class Posterize(Layer):
    """:164-182."""

    def __init__(self, bits, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.bits = bits
        self._shift = 8 - bits
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        # TF's shift functors clamp the count to bit-width-1 (reached by AutoAugment sub-policy 22)
        return K.aug_pointwise(inputs, K.PW_POSTERIZE, i0=min(max(int(self._shift), 0), 7))

    def dispatch_item(self, height, width, **kwargs):
        return K.aug_item(_lib.AUG_POSTERIZE, i=(min(max(int(self._shift), 0), 7),))

    def get_config(self):
        return _base_cfg(self, {"bits": self.bits})


@register_keras_serializable(package="Chambers")
class Solarize(Layer):
    """:186-201."""

    def __init__(self, threshold=128, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.threshold = threshold
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        return K.aug_pointwise(inputs, K.PW_SOLARIZE, i0=int(self.threshold))

    def dispatch_item(self, height, width, **kwargs):
        return K.aug_item(_lib.AUG_SOLARIZE, i=(int(self.threshold),))

    def get_config(self):
        return _base_cfg(self, {"threshold": self.threshold})


@register_keras_serializable(package="Chambers")
class SolarizeAdd(Layer):
    """:205-223."""

    def __init__(self, addition=0, threshold=128, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.addition = addition
        self.threshold = threshold
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        return K.aug_pointwise(inputs, K.PW_SOLARIZE_ADD, i0=int(self.threshold), i1=int(self.addition))

    def dispatch_item(self, height, width, **kwargs):
        return K.aug_item(_lib.AUG_SOLARIZE_ADD, i=(int(self.threshold), int(self.addition)))

    def get_config(self):
        return _base_cfg(self, {"addition": self.addition, "threshold": self.threshold})


def _blend_passthrough(inputs, factor):
    """blend() short-cuts, :28-31."""
    return factor == 1.0


@register_keras_serializable(package="Chambers")
class Color(Layer):
    """:227-243."""

    def __init__(self, factor, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.factor = factor
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        if self.factor == 1.0:
            return inputs.clone()
        return K.aug_pointwise(inputs, K.PW_COLOR, factor=self.factor)

    def dispatch_item(self, height, width, **kwargs):
        return K.aug_item(_lib.AUG_COLOR, f=(self.factor,))     # blend()'s factor 0 / 1 short-cuts are what the arithmetic gives

    def get_config(self):
        return _base_cfg(self, {"factor": self.factor})


@register_keras_serializable(package="Chambers")
class Contrast(Layer):
    """:247-273.  The degenerate 'mean' is (pixels in the WHOLE input tensor)/256, as in the reference."""

    def __init__(self, factor, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.factor = factor
        self.input_spec = InputSpec(ndim=4)

    @staticmethod
    def degenerate_constant(n_pixels):
        mean = np.float32(n_pixels) / np.float32(256.0)
        return int(min(max(mean, np.float32(0.0)), np.float32(255.0)))

    def call(self, inputs, **kwargs):
        if self.factor == 1.0:
            return inputs.clone()
        b, h, w, _ = inputs.shape
        const = self.degenerate_constant(b * h * w)
        if self.factor == 0.0:
            return torch.full_like(inputs, const)
        return K.aug_pointwise(inputs, K.PW_CONTRAST, factor=self.factor, i0=const)

    def dispatch_item(self, height, width, **kwargs):
        # per image = a batch-1 tensor: the "mean" is H*W/256 (196 for 224x224, clipped to 255 from 256x256 on)
        return K.aug_item(_lib.AUG_CONTRAST, i=(self.degenerate_constant(height * width),), f=(self.factor,))

    def get_config(self):
        return _base_cfg(self, {"factor": self.factor})


@register_keras_serializable(package="Chambers")
class Brightness(Layer):
    """:277-293."""

    def __init__(self, factor, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.factor = factor
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        if self.factor == 1.0:
            return inputs.clone()
        if self.factor == 0.0:
            return torch.zeros_like(inputs)
        return K.aug_pointwise(inputs, K.PW_BRIGHTNESS, factor=self.factor)

    def dispatch_item(self, height, width, **kwargs):
        return K.aug_item(_lib.AUG_BRIGHTNESS, f=(self.factor,))

    def get_config(self):
        return _base_cfg(self, {"factor": self.factor})


@register_keras_serializable(package="Chambers")
class Sharpness(Layer):
    """:297-312 (tfa.image.sharpness)."""

    def __init__(self, factor, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.factor = factor
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        if self.factor == 1.0:
            return inputs.clone()
        return K.aug_sharpness(inputs, self.factor)

    def dispatch_item(self, height, width, **kwargs):
        if self.factor == 1.0:
            return K.aug_item(_lib.AUG_IDENTITY)
        return K.aug_item(_lib.AUG_SHARPNESS, f=(self.factor,))

    def get_config(self):
        return _base_cfg(self, {"factor": self.factor})


@register_keras_serializable(package="Chambers")
class ShearX(_Warp):
    """:316-355."""

    def __init__(self, level, interpolation="nearest", fill_mode="constant", fill_value=0.0, name=None, **kwargs):
        super().__init__(interpolation, fill_mode, fill_value, name=name, **kwargs)
        self.level = level

    def call(self, inputs, negate=None, **kwargs):
        level = _randomly_negate_value(self.level, negate)
        return self._warp(inputs, [1.0, level, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])

    def dispatch_item(self, height, width, negate=None, **kwargs):
        return self._item([1.0, _randomly_negate_value(self.level, negate), 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])

    def get_config(self):
        return _base_cfg(self, dict({"level": self.level}, **self._warp_cfg()))


@register_keras_serializable(package="Chambers")
class ShearY(_Warp):
    """:359-398."""

    def __init__(self, level, interpolation="nearest", fill_mode="constant", fill_value=0.0, name=None, **kwargs):
        super().__init__(interpolation, fill_mode, fill_value, name=name, **kwargs)
        self.level = level

    def call(self, inputs, negate=None, **kwargs):
        level = _randomly_negate_value(self.level, negate)
        return self._warp(inputs, [1.0, 0.0, 0.0, level, 1.0, 0.0, 0.0, 0.0])

    def dispatch_item(self, height, width, negate=None, **kwargs):
        return self._item([1.0, 0.0, 0.0, _randomly_negate_value(self.level, negate), 1.0, 0.0, 0.0, 0.0])

    def get_config(self):
        return _base_cfg(self, dict({"level": self.level}, **self._warp_cfg()))


@register_keras_serializable(package="Chambers")
class TranslateX(_Warp):
    """:402-441 (tfa.image.translate([-pixels, 0]) -> transform [1,0,-dx,0,1,-dy,0,0])."""

    def __init__(self, pixels, interpolation="nearest", fill_mode="constant", fill_value=0.0, name=None, **kwargs):
        super().__init__(interpolation, fill_mode, fill_value, name=name, **kwargs)
        self.pixels = pixels

    def call(self, inputs, negate=None, **kwargs):
        pixels = _randomly_negate_value(self.pixels, negate)
        dx = -pixels
        return self._warp(inputs, [1.0, 0.0, -dx, 0.0, 1.0, -0.0, 0.0, 0.0])

    def dispatch_item(self, height, width, negate=None, **kwargs):
        dx = -_randomly_negate_value(self.pixels, negate)
        return self._item([1.0, 0.0, -dx, 0.0, 1.0, -0.0, 0.0, 0.0])

    def get_config(self):
        return _base_cfg(self, dict({"pixels": self.pixels}, **self._warp_cfg()))


@register_keras_serializable(package="Chambers")
class TranslateY(_Warp):
    """:445-484."""

    def __init__(self, pixels, interpolation="nearest", fill_mode="constant", fill_value=0.0, name=None, **kwargs):
        super().__init__(interpolation, fill_mode, fill_value, name=name, **kwargs)
        self.pixels = pixels

    def call(self, inputs, negate=None, **kwargs):
        pixels = _randomly_negate_value(self.pixels, negate)
        dy = -pixels
        return self._warp(inputs, [1.0, 0.0, -0.0, 0.0, 1.0, -dy, 0.0, 0.0])

    def dispatch_item(self, height, width, negate=None, **kwargs):
        dy = -_randomly_negate_value(self.pixels, negate)
        return self._item([1.0, 0.0, -0.0, 0.0, 1.0, -dy, 0.0, 0.0])

    def get_config(self):
        return _base_cfg(self, dict({"pixels": self.pixels}, **self._warp_cfg()))


@register_keras_serializable(package="Chambers")
class CutOut(Layer):
    """:488-507 (tfa.image.random_cutout: one centre per image)."""

    def __init__(self, mask_size, constant_values=0, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.mask_size = mask_size
        self.constant_values = constant_values
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, centers=None, **kwargs):
        b, h, w, _ = inputs.shape
        if centers is None:
            g = rng.host_generator()
            centers = np.stack([g.integers(0, h, size=b), g.integers(0, w, size=b)], axis=1).astype(np.int32)
        return K.aug_cutout(inputs, centers, self.mask_size, self.constant_values)

    def dispatch_item(self, height, width, centers=None, **kwargs):
        if int(self.mask_size) % 2 != 0:
            raise ValueError("mask_size should be divisible by 2")
        if centers is None:
            g = rng.host_generator()
            cy, cx = int(g.integers(0, height)), int(g.integers(0, width))
        else:
            cy, cx = (int(v) for v in np.asarray(centers).reshape(-1)[:2])
        return K.aug_item(_lib.AUG_CUTOUT, i=(cy, cx, int(self.mask_size) // 2, int(self.constant_values) & 0xff))

    def get_config(self):
        return _base_cfg(self, {"mask_size": self.mask_size, "constant_values": self.constant_values})


def batch_item(transform, batch, height, width, **kwargs):
    """(op record, per-image cutout centres or None) of `transform` applied to a whole [batch, height, width, 3] tensor: the
    record of dispatch_item (which describes a batch of ONE image) with the two things that differ for a batch put right -
    Contrast's constant counts the pixels of the whole tensor (:253-257), CutOut draws one centre per image (:497-503)."""
    if isinstance(transform, RandomChance):
        apply = kwargs.pop("apply", None)
        if apply is None:
            apply = bool(rng.host_generator().uniform() < transform.probability)
        if not apply:
            return K.aug_item(_lib.AUG_IDENTITY), None
        return batch_item(transform.transform, batch, height, width, **kwargs)
    if isinstance(transform, Contrast):
        return K.aug_item(_lib.AUG_CONTRAST, i=(transform.degenerate_constant(batch * height * width),), f=(transform.factor,)), None
    if isinstance(transform, CutOut):
        centers = kwargs.get("centers")
        if centers is None:
            g = rng.host_generator()
            centers = np.stack([g.integers(0, height, size=batch), g.integers(0, width, size=batch)], axis=1)
        if not isinstance(centers, torch.Tensor):        # a device tensor = resident decisions, used as they are
            centers = np.ascontiguousarray(centers, dtype=np.int32).reshape(batch, 2)
        return transform.dispatch_item(height, width, centers=(0, 0)), centers
    return transform.dispatch_item(height, width, **kwargs), None


# ---------------------------------------------------------------- combinators
@register_keras_serializable(package="Chambers")
class RandomChance(Layer):
    """:514-545 — one uniform draw per call (shared by the batch)."""

    _forward_kwargs = True

    def __init__(self, transform, probability, name=None, **kwargs):
        if name is None and transform.name is not None:
            name = "random_chance_" + transform.name
        super().__init__(name=name, **kwargs)
        self.transform = transform
        self.probability = probability

    def _sublayers(self):
        return [self.transform]

    def call(self, inputs, apply=None, **kwargs):
        if apply is None:
            apply = bool(rng.host_generator().uniform() < self.probability)
        return self.transform(inputs, **kwargs) if apply else inputs

    def dispatch_item(self, height, width, apply=None, **kwargs):
        if apply is None:
            apply = bool(rng.host_generator().uniform() < self.probability)
        return self.transform.dispatch_item(height, width, **kwargs) if apply else K.aug_item(_lib.AUG_IDENTITY)

    def compute_output_shape(self, input_shape):
        return self.transform.compute_output_shape(input_shape)

    def get_config(self):
        return _base_cfg(self, {"transform": serialize(self.transform), "probability": self.probability})

    @classmethod
    def from_config(cls, config):
        config["transform"] = deserialize(config["transform"])
        return cls(**config)


@register_keras_serializable(package="Chambers")
class RandomChoice(Layer):
    """:549-617.  `choices` (list of n_transforms indices) and per-slot kwargs may be given
    explicitly; elementwise=True applies the draws per image (tf.map_fn over batch-1 tensors)."""

    _forward_kwargs = True

    def __init__(self, transforms, n_transforms, elementwise=False, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.transforms = transforms
        self.n_transforms = n_transforms
        self.elementwise = elementwise

    def _sublayers(self):
        return list(self.transforms)

    def call(self, inputs, choices=None, slot_kwargs=None, **kwargs):
        if self.elementwise:
            return self._elementwise(inputs, choices, slot_kwargs)
        return self._random_transforms(inputs, choices, slot_kwargs)

    def _elementwise(self, inputs, choices, slot_kwargs):
        """tf.map_fn over batch-1 tensors (:565-567): every image draws its own transform index per slot, and the chosen
        transform sees a batch of one (its sign draw, cutout centre and Contrast's constant are per image).  When every
        transform can describe itself as an op record, the whole batch runs at once, each workgroup evaluating its own image's
        chain and the images sorted by what their chain needs (chb_aug_fused_items_sorted; chains longer than the fused kernels hold: one dispatch launch per
        slot, chb_aug_dispatch); transforms that cannot (user-supplied layers) take the image-by-image route."""
        b = inputs.shape[0]
        if b == 0:
            return inputs
        if not all(hasattr(t, "dispatch_item") for t in self.transforms) or inputs.dim() != 4 or inputs.shape[-1] != 3:
            outs = []
            for n in range(b):
                ch = None if choices is None else choices[n]
                sk = None if slot_kwargs is None else slot_kwargs[n]
                outs.append(self._random_transforms(inputs[n:n + 1], ch, sk))
            return K.concat_batch(outs)
        h, w = int(inputs.shape[1]), int(inputs.shape[2])
        items = self.elementwise_items(b, h, w, choices, slot_kwargs)
        if self.n_transforms <= K.FUSED_MAX_OPS:
            return K.aug_fused_items(inputs, items)        # every image's own chain, one launch per group of chains (chb_aug_fused_items_sorted)
        x = inputs
        for i in range(self.n_transforms):
            x = K.aug_dispatch(x, items[i])
        return x

    def elementwise_items(self, b, h, w, choices=None, slot_kwargs=None):
        """[n_transforms, B] op records of one elementwise call, drawn in the order of the reference's map_fn body: image by image,
        slot by slot (the transform index, then whatever the chosen transform draws)."""
        items = np.zeros((self.n_transforms, b), dtype=K.AUG_ITEM_DTYPE)
        for n in range(b):
            for i in range(self.n_transforms):
                idx = int(rng.host_generator().integers(0, len(self.transforms))) if choices is None else int(choices[n][i])
                kw = {} if slot_kwargs is None else dict(slot_kwargs[n][i])
                items[i, n] = self.transforms[idx].dispatch_item(h, w, **kw)
        return items

    def _random_transforms(self, inputs, choices=None, slot_kwargs=None):
        if self._can_fuse(inputs):
            return K.aug_fused(inputs, self.plan(inputs.shape, choices, slot_kwargs))
        for i in range(self.n_transforms):
            if choices is None:
                idx = int(rng.host_generator().integers(0, len(self.transforms)))
            else:
                idx = int(choices[i])
            kw = {} if slot_kwargs is None else dict(slot_kwargs[i])
            inputs = self.transforms[idx](inputs, **kw)
        return inputs

    fused = True      # batch-shared chains of describable ops run as ONE launch (chb_aug_fused); False = one launch per op

    def _can_fuse(self, inputs):
        return (self.fused and 1 <= self.n_transforms <= K.FUSED_MAX_OPS and inputs.dim() == 4 and inputs.shape[-1] == 3
                and inputs.dtype == torch.uint8 and inputs.shape[0] > 0 and all(hasattr(t, "dispatch_item") for t in self.transforms))

    def plan(self, input_shape, choices=None, slot_kwargs=None):
        """Resolve one batch-shared call to op records (K.AugPlan), drawing what was not given in the order the op-by-op route
        draws it: per slot the transform index, then whatever the chosen transform draws (sign, chance, cutout centres)."""
        b, h, w = int(input_shape[0]), int(input_shape[1]), int(input_shape[2])
        items, centers = [], []
        for i in range(self.n_transforms):
            idx = int(rng.host_generator().integers(0, len(self.transforms))) if choices is None else int(choices[i])
            kw = {} if slot_kwargs is None else dict(slot_kwargs[i])
            item, cen = batch_item(self.transforms[idx], b, h, w, **kw)
            items.append(item)
            centers.append(cen)
        return K.AugPlan(items, centers)

    def compute_output_shape(self, input_shape):
        shapes = [t.compute_output_shape(input_shape) for t in self.transforms]
        shapes = np.array(shapes, dtype=float)
        shape0 = shapes[0]
        identical = np.all(shapes == shape0, axis=0).astype(int)
        out = (shape0 * identical).astype(int).tolist()
        return [None if d == 0 else d for d in out]

    def get_config(self):
        return _base_cfg(self, {"transforms": [serialize(t) for t in self.transforms], "n_transforms": self.n_transforms,
                                "elementwise": self.elementwise})

    @classmethod
    def from_config(cls, config):
        config["transforms"] = [deserialize(t) for t in config["transforms"]]
        return cls(**config)


@register_keras_serializable(package="Chambers")
class ImageNetNormalization(Layer):
    """:621-682."""

    def __init__(self, mode="caffe", name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        if mode not in {"caffe", "tf", "torch"}:
            raise ValueError("Unknown mode " + str(mode))
        self.mode = mode
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        return K.normalize(inputs, self.mode)

    def get_config(self):
        return _base_cfg(self, {"mode": self.mode})


for _cls in (AutoContrast, Equalize, Invert, Rotate, Posterize, Solarize, SolarizeAdd, Color, Contrast, Brightness, Sharpness,
             ShearX, ShearY, TranslateX, TranslateY, CutOut):
    _cls._forward_kwargs = True
