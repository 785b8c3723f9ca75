"""chambers.augmentations.augmentation_schemes on MI355X: RandAugment / AutoAugment.

Mirrors /root/reference/chambers/augmentations/augmentation_schemes.py (policy :12-39,
magnitude maps :42-102, `_get_transform` :105-128, AutoAugment :132-171, RandAugment :175-225).
The reference selects ops with n*K `tf.cond`s over TF's global RNG; here the selection is made on
the host (explicit `decisions`, or drawn from `chambers_amd.rng`) and only the chosen op's HIP
kernel is launched.
"""
import numpy as np
import torch

from .. import rng
from .._keras_like import InputSpec, Layer, Sequential, register_keras_serializable
from . import image_augmentations

_INTERPOLATION_MODE = "nearest"
_FILL_MODE = "constant"
_FILL_VALUE = 128
_MAX_MAGNITUDE = 10.0

# AutoAugment policy "v0" (augmentation_schemes.py:12-39): 25 sub-policies of two (op, probability, magnitude) steps; "-" = the op
# takes no magnitude.
_POLICY_V0_TABLE = """
    Equalize 0.8 -         | ShearY 0.8 4
    Color 0.4 9            | Equalize 0.6 -
    Color 0.4 1            | Rotate 0.6 8
    Solarize 0.8 3         | Equalize 0.4 7
    Solarize 0.4 2         | Solarize 0.6 2
    Color 0.2 0            | Equalize 0.8 -
    Equalize 0.4 -         | SolarizeAdd 0.8 3
    ShearX 0.2 9           | Rotate 0.6 8
    Color 0.6 1            | Equalize 1.0 -
    Invert 0.4 -           | Rotate 0.6 0
    Equalize 1.0 -         | ShearY 0.6 3
    Color 0.4 7            | Equalize 0.6 -
    Posterize 0.4 6        | AutoContrast 0.4 -
    Solarize 0.6 8         | Color 0.6 9
    Solarize 0.2 4         | Rotate 0.8 9
    Rotate 1.0 7           | TranslateY 0.8 9
    ShearX 0.0 0           | Solarize 0.8 4
    ShearY 0.8 0           | Color 0.6 4
    Color 1.0 0            | Rotate 0.6 2
    Equalize 0.8 -         | Equalize 0.0 -
    Equalize 1.0 -         | AutoContrast 0.6 -
    ShearY 0.4 7           | SolarizeAdd 0.6 7
    Posterize 0.8 2        | Solarize 0.6 10
    Solarize 0.6 8         | Equalize 0.6 1
    Color 0.8 6            | Rotate 0.4 5
"""


def _parse_policy(table):
    def step(text):
        name, prob, mag = text.split()
        return (name, float(prob), None if mag == "-" else int(mag))
    return [[step(part) for part in line.split("|")] for line in table.strip().splitlines()]


_AUTO_AUGMENT_POLICY_V0 = _parse_policy(_POLICY_V0_TABLE)


def _magnitude_to_enhance_kwargs(magnitude):
    return {"factor": magnitude / _MAX_MAGNITUDE * 1.8 + 0.1}


def _warp_kwargs():
    return {"interpolation": _INTERPOLATION_MODE, "fill_mode": _FILL_MODE, "fill_value": _FILL_VALUE}


def _magnitude_to_shear_kwargs(magnitude):
    return dict({"level": magnitude / _MAX_MAGNITUDE * 0.3}, **_warp_kwargs())


def _magnitude_to_translate_kwargs(magnitude):
    return dict({"pixels": magnitude / _MAX_MAGNITUDE * 100}, **_warp_kwargs())


def _magnitude_to_posterize_kwargs(magnitude):
    return {"bits": int(magnitude / _MAX_MAGNITUDE * 4)}


def _magnitude_to_solarize_kwargs(magnitude):
    return {"threshold": int(magnitude / _MAX_MAGNITUDE * 256)}


def _magnitude_to_solarizeadd_kwargs(magnitude):
    return {"addition": int(magnitude / _MAX_MAGNITUDE * 110)}


def _magnitude_to_rotate_kwargs(magnitude):
    return dict({"degrees": magnitude / _MAX_MAGNITUDE * 30.0}, **_warp_kwargs())


def _magnitude_to_cutout_kwargs(magnitude):
    return {"mask_size": int(magnitude / _MAX_MAGNITUDE * 80), "constant_values": _FILL_VALUE}


def _get_transform(transform_name, magnitude):
    magnitude_fn_map = {
        "AutoContrast": lambda magnitude: {},
        "Equalize": lambda magnitude: {},
        "Invert": lambda magnitude: {},
        "Brightness": _magnitude_to_enhance_kwargs,
        "Contrast": _magnitude_to_enhance_kwargs,
        "Color": _magnitude_to_enhance_kwargs,
        "Sharpness": _magnitude_to_enhance_kwargs,
        "ShearX": _magnitude_to_shear_kwargs,
        "ShearY": _magnitude_to_shear_kwargs,
        "TranslateX": _magnitude_to_translate_kwargs,
        "TranslateY": _magnitude_to_translate_kwargs,
        "Posterize": _magnitude_to_posterize_kwargs,
        "Solarize": _magnitude_to_solarize_kwargs,
        "SolarizeAdd": _magnitude_to_solarizeadd_kwargs,
        "CutOut": _magnitude_to_cutout_kwargs,
        "Rotate": _magnitude_to_rotate_kwargs,
    }
    transform = getattr(image_augmentations, transform_name)
    kwargs = magnitude_fn_map[transform_name](magnitude)
    return transform(**kwargs)


def _is_training(training):
    # keras learning_phase() defaults to 0 (inference) outside fit()
    return bool(training) if training is not None else False


@register_keras_serializable(package="Chambers")
class AutoAugment(Layer):
    """ Applies a random augmentation pair to each image """

    def __init__(self, elementwise=False, name=None, **kwargs):
        super(AutoAugment, self).__init__(name=name, **kwargs)
        self.elementwise = elementwise
        self.transforms = [
            Sequential([
                image_augmentations.RandomChance(_get_transform(t1, m1), p1),
                image_augmentations.RandomChance(_get_transform(t2, m2), p2),
            ])
            for (t1, p1, m1), (t2, p2, m2) in _AUTO_AUGMENT_POLICY_V0
        ]
        self._transform = image_augmentations.RandomChoice(self.transforms, n_transforms=1, elementwise=elementwise)
        self.input_spec = InputSpec(ndim=4, dtype=torch.uint8)

    def _sublayers(self):
        return [self._transform]

    def draw_decision(self):
        """One sub-policy choice + the two chance draws + sign draws, from the host generator."""
        g = rng.host_generator()
        policy = int(g.integers(0, len(self.transforms)))
        sub = _AUTO_AUGMENT_POLICY_V0[policy]
        apply = tuple(bool(g.uniform() < p) for (_t, p, _m) in sub)
        negate = tuple(bool(g.uniform() < 0.5) for _ in sub)
        return {"policy": policy, "apply": apply, "negate": negate}

    def plan(self, input_shape, decision=None):
        """The batch-shared call resolved to op records (kernels.AugPlan): the drawn sub-policy's two RandomChance steps."""
        from .. import kernels as K
        decision = decision if decision is not None else self.draw_decision()
        b, h, w = int(input_shape[0]), int(input_shape[1]), int(input_shape[2])
        seq = self.transforms[int(decision["policy"])]
        pairs = [image_augmentations.batch_item(chance, b, h, w, apply=bool(decision["apply"][j]), negate=bool(decision["negate"][j]))
                 for j, chance in enumerate(seq.layers)]
        return K.AugPlan([p[0] for p in pairs], [p[1] for p in pairs])

    fused = True      # one launch for the pair (chb_aug_fused); False = one launch per applied op

    def _apply(self, inputs, decision):
        if self.fused and inputs.dim() == 4 and inputs.shape[-1] == 3 and inputs.shape[0] > 0:
            from .. import kernels as K
            return K.aug_fused(inputs, self.plan(inputs.shape, decision))
        seq = self.transforms[int(decision["policy"])]
        x = inputs
        for j, chance in enumerate(seq.layers):
            x = chance(x, apply=bool(decision["apply"][j]), negate=bool(decision["negate"][j]))
        return x

    def call(self, inputs, training=None, decision=None, **kwargs):
        if not _is_training(training):
            return inputs
        if self.elementwise:
            return self._elementwise(inputs, decision)
        return self._apply(inputs, decision if decision is not None else self.draw_decision())

    def _elementwise(self, inputs, decisions):
        """elementwise=True (:135, RandomChoice :563-570): every image draws its own sub-policy, its two chance draws and its
        sign draws.  The two steps of the 25 sub-policies run inside the patchify pass, each workgroup its own image's pair, the images sorted by what the pair needs (chb_aug_fused_items_sorted)."""
        from .. import kernels as K
        b = inputs.shape[0]
        if b == 0:
            return inputs
        h, w = int(inputs.shape[1]), int(inputs.shape[2])
        return K.aug_fused_items(inputs, self.elementwise_items(b, h, w, decisions))

    def elementwise_items(self, b, h, w, decisions=None):
        """[2, B] op records of one elementwise call: every image its own sub-policy, chance draws and signs."""
        from .. import kernels as K
        items = np.zeros((2, b), dtype=K.AUG_ITEM_DTYPE)
        for n in range(b):
            d = decisions[n] if decisions is not None else self.draw_decision()
            seq = self.transforms[int(d["policy"])]
            for j, chance in enumerate(seq.layers):
                items[j, n] = chance.dispatch_item(h, w, apply=bool(d["apply"][j]), negate=bool(d["negate"][j]))
        return items

    def items_plan(self, input_shape, decisions=None):
        """The elementwise call resolved to per-image op records (kernels.AugItemsPlan) - what ViTEngine.forward(..., augment=plan)
        evaluates inside its normalise + patchify pass."""
        from .. import kernels as K
        return K.AugItemsPlan(self.elementwise_items(int(input_shape[0]), int(input_shape[1]), int(input_shape[2]), decisions))

    def compute_output_shape(self, input_shape):
        return self._transform.compute_output_shape(input_shape)

    def get_config(self):
        config = {"elementwise": self.elementwise}
        base_config = super(AutoAugment, self).get_config()
        return dict(list(base_config.items()) + list(config.items()))


@register_keras_serializable(package="Chambers")
class RandAugment(Layer):
    _OPS = ["AutoContrast", "Equalize", "Invert", "Brightness", "Contrast", "Color", "Sharpness", "ShearX", "ShearY",
            "TranslateX", "TranslateY", "Posterize", "Solarize", "SolarizeAdd", "CutOut", "Rotate"]

    def __init__(self, n_transforms, magnitude, elementwise=False, name=None, **kwargs):
        super(RandAugment, self).__init__(name=name, **kwargs)
        self.n_transforms = n_transforms
        self.magnitude = magnitude
        self.elementwise = elementwise
        self.transforms = [_get_transform(op, magnitude) for op in self._OPS]
        self._transform = image_augmentations.RandomChoice(self.transforms, n_transforms=n_transforms, elementwise=elementwise)
        self.input_spec = InputSpec(ndim=4, dtype=torch.uint8)

    def _sublayers(self):
        return [self._transform]

    def call(self, inputs, training=None, decisions=None, **kwargs):
        """decisions (optional): list of n_transforms dicts {"op": index, "negate": bool, "centers": [B,2]}
        (a list of such lists, one per image, when elementwise=True)."""
        if not _is_training(training):
            return inputs
        if decisions is None:
            return self._transform(inputs)

        def split(ds):
            choices = [int(d["op"]) for d in ds]
            kws = [{k: v for k, v in d.items() if k != "op"} for d in ds]
            return choices, kws

        if self.elementwise:
            pairs = [split(ds) for ds in decisions]
            return self._transform(inputs, choices=[p[0] for p in pairs], slot_kwargs=[p[1] for p in pairs])
        choices, kws = split(decisions)
        return self._transform(inputs, choices=choices, slot_kwargs=kws)

    def items_plan(self, input_shape, decisions=None):
        """The elementwise call resolved to per-image op records (kernels.AugItemsPlan; decisions: one list of n_transforms dicts
        per image, centers = (cy, cx)) - what ViTEngine.forward(..., augment=plan) evaluates inside its normalise + patchify pass."""
        from .. import kernels as K
        b, h, w = int(input_shape[0]), int(input_shape[1]), int(input_shape[2])
        if decisions is None:
            return K.AugItemsPlan(self._transform.elementwise_items(b, h, w))
        choices = [[int(d["op"]) for d in ds] for ds in decisions]
        kws = [[{k: v for k, v in d.items() if k != "op"} for d in ds] for ds in decisions]
        return K.AugItemsPlan(self._transform.elementwise_items(b, h, w, choices, kws))

    def plan(self, input_shape, decisions=None):
        """The batch-shared call resolved to op records (kernels.AugPlan) - what ViTEngine.forward(..., augment=plan) fuses
        into its normalise + patchify pass."""
        if self.elementwise:
            raise ValueError("plan() describes the batch-shared mode; elementwise=True resolves to per-image records: items_plan()")
        if decisions is None:
            return self._transform.plan(input_shape)
        return self._transform.plan(input_shape, [int(d["op"]) for d in decisions], [{k: v for k, v in d.items() if k != "op"} for d in decisions])

    def compute_output_shape(self, input_shape):
        return self._transform.compute_output_shape(input_shape)

    def get_config(self):
        config = {"n_transforms": self.n_transforms, "magnitude": self.magnitude, "elementwise": self.elementwise}
        base_config = super(RandAugment, self).get_config()
        return dict(list(base_config.items()) + list(config.items()))
