"""Input side of the augmentation path on MI355X: the Keras preprocessing layers the reference re-exports from
`chambers.augmentations` (augmentations/__init__.py:1-13: Resizing, CenterCrop, RandomCrop, RandomFlip, Rescaling) and its
own `ResizingMinMax` (augmentations/image_augmentations.py:686-748).  Same constructor arguments and get_config() keys;
arithmetic in chambers_amd/csrc/imageio.hip (tf.image.resize TF2 semantics: half-pixel centres, no antialias).

Random decisions are explicit keyword arguments, drawn from `chambers_amd.rng` when omitted: RandomCrop draws ONE window for
the whole batch (tf.image.random_crop on the batched tensor), RandomFlip flips every image independently.  At inference
(training falsy) RandomCrop is the centre crop (inputs at least as large as the target) and RandomFlip the identity.
Not built: RandomRotation / RandomZoom / RandomTranslation / RandomContrast / RandomHeight / RandomWidth."""
import numpy as np
import torch

from .. import kernels as K
from .. import rng
from .._keras_like import InputSpec, Layer, register_keras_serializable


def _is_training(training):
    return bool(training)


def _cfg(layer, config):
    base = Layer.get_config(layer)
    return dict(list(base.items()) + list(config.items()))


@register_keras_serializable(package="Chambers")
class Resizing(Layer):
    """keras Resizing(height, width, interpolation="bilinear"): tf.image.resize; bilinear returns float32."""

    def __init__(self, height, width, interpolation="bilinear", name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        if interpolation not in ("bilinear", "nearest"):
            raise NotImplementedError("interpolation %r: bilinear and nearest are built" % (interpolation,))
        self.target_height, self.target_width, self.interpolation = int(height), int(width), interpolation
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        return K.resize(inputs, self.target_height, self.target_width, self.interpolation)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], self.target_height, self.target_width, input_shape[3])

    def get_config(self):
        return _cfg(self, {"height": self.target_height, "width": self.target_width, "interpolation": self.interpolation})


@register_keras_serializable(package="Chambers")
class ResizingMinMax(Layer):
    """image_augmentations.py:686-748: smallest side -> min_side or largest side -> max_side (whichever shrinks more when both
    are given), aspect ratio kept; size arithmetic in float32 with a truncating cast (:712-731)."""

    def __init__(self, min_side=None, max_side=None, interpolation="bilinear", name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        if min_side is None and max_side is None:
            raise ValueError("Must specify either 'min_side' or 'max_side'.")
        self.min_side, self.max_side, self.interpolation = min_side, max_side, interpolation
        self.input_spec = InputSpec(ndim=4)

    def target_size(self, height, width):
        h, w = np.float32(height), np.float32(width)
        if self.min_side is not None and self.max_side is not None:
            scale = np.minimum(np.float32(self.max_side) / np.maximum(w, h), np.float32(self.min_side) / np.minimum(w, h))
        elif self.min_side is not None:
            scale = np.float32(self.min_side) / np.minimum(w, h)
        else:
            scale = np.float32(self.max_side) / np.maximum(w, h)
        scale = np.float32(scale)
        return int(np.float32(h * scale)), int(np.float32(w * scale))

    def call(self, inputs, **kwargs):
        new_h, new_w = self.target_size(inputs.shape[1], inputs.shape[2])
        return Resizing(height=new_h, width=new_w, interpolation=self.interpolation)(inputs)

    def compute_output_shape(self, input_shape):
        return [input_shape[0], self.min_side, self.max_side, input_shape[3]]     # as the reference (:739-740)

    def get_config(self):
        return _cfg(self, {"min_side": self.min_side, "max_side": self.max_side, "interpolation": self.interpolation})


@register_keras_serializable(package="Chambers")
class CenterCrop(Layer):
    """keras CenterCrop(height, width): the central window, start = int((size - target) / 2)."""

    def __init__(self, height, width, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.target_height, self.target_width = int(height), int(width)
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, **kwargs):
        h, w = inputs.shape[1], inputs.shape[2]
        if h < self.target_height or w < self.target_width:
            raise ValueError("CenterCrop target %s exceeds the input %s" % ((self.target_height, self.target_width), (h, w)))
        off = (int((h - self.target_height) / 2), int((w - self.target_width) / 2))
        return K.crop_flip(inputs, self.target_height, self.target_width, offsets=off)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], self.target_height, self.target_width, input_shape[3])

    def get_config(self):
        return _cfg(self, {"height": self.target_height, "width": self.target_width})


@register_keras_serializable(package="Chambers")
class RandomCrop(Layer):
    """keras RandomCrop(height, width, seed=None): training — one uniformly drawn window for the whole batch; inference — the
    centre crop."""

    def __init__(self, height, width, seed=None, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.height, self.width, self.seed = int(height), int(width), seed
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, training=True, offset=None, **kwargs):
        h, w = inputs.shape[1], inputs.shape[2]
        if h < self.height or w < self.width:
            raise ValueError("RandomCrop target %s exceeds the input %s" % ((self.height, self.width), (h, w)))
        if not _is_training(training):
            offset = (int((h - self.height) / 2), int((w - self.width) / 2))
        elif offset is None:
            g = rng.host_generator()
            offset = (int(g.integers(0, h - self.height + 1)), int(g.integers(0, w - self.width + 1)))
        return K.crop_flip(inputs, self.height, self.width, offsets=(int(offset[0]), int(offset[1])))

    def compute_output_shape(self, input_shape):
        return (input_shape[0], self.height, self.width, input_shape[3])

    def get_config(self):
        return _cfg(self, {"height": self.height, "width": self.width, "seed": self.seed})


@register_keras_serializable(package="Chambers")
class RandomFlip(Layer):
    """keras RandomFlip(mode="horizontal_and_vertical", seed=None): every image flipped left-right / up-down independently with
    probability 1/2 each; identity at inference."""

    def __init__(self, mode="horizontal_and_vertical", seed=None, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        if mode not in ("horizontal", "vertical", "horizontal_and_vertical"):
            raise ValueError("RandomFlip layer %s received an unknown mode argument %s" % (name, mode))
        self.mode, self.seed = mode, seed
        self.horizontal = mode in ("horizontal", "horizontal_and_vertical")
        self.vertical = mode in ("vertical", "horizontal_and_vertical")
        self.input_spec = InputSpec(ndim=4)

    def call(self, inputs, training=True, flip_horizontal=None, flip_vertical=None, **kwargs):
        if not _is_training(training):
            return inputs
        b = inputs.shape[0]
        g = rng.host_generator()

        def bits(given, enabled):
            """per-image decisions as uint8 {0,1}: a device tensor stays on the device (no host round trip), anything else is numpy"""
            if not enabled:
                return np.zeros(b, dtype=np.uint8)
            if isinstance(given, torch.Tensor):
                return given.to(device=inputs.device, dtype=torch.uint8)
            return (np.asarray(given, dtype=bool) if given is not None else g.uniform(size=b) < 0.5).astype(np.uint8)

        fh, fv = bits(flip_horizontal, self.horizontal), bits(flip_vertical, self.vertical)
        if isinstance(fh, torch.Tensor) or isinstance(fv, torch.Tensor):
            fh = fh if isinstance(fh, torch.Tensor) else torch.as_tensor(fh, device=inputs.device)
            fv = fv if isinstance(fv, torch.Tensor) else torch.as_tensor(fv, device=inputs.device)
            flips = fh | (fv << 1)
        else:
            flips = fh | (fv << 1)
        return K.crop_flip(inputs, inputs.shape[1], inputs.shape[2], offsets=(0, 0), flips=flips)

    def compute_output_shape(self, input_shape):
        return input_shape

    def get_config(self):
        return _cfg(self, {"mode": self.mode, "seed": self.seed})


@register_keras_serializable(package="Chambers")
class Rescaling(Layer):
    """keras Rescaling(scale, offset=0.): float32(inputs) * scale + offset."""

    def __init__(self, scale, offset=0.0, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.scale, self.offset = scale, offset

    def call(self, inputs, **kwargs):
        return K.rescale(inputs, self.scale, self.offset)

    def compute_output_shape(self, input_shape):
        return input_shape

    def get_config(self):
        return _cfg(self, {"scale": self.scale, "offset": self.offset})
