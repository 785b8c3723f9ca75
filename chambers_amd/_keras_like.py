"""The small part of the Keras Layer/Model contract the chambers API surface relies on
(SURVEY §8b): auto-naming, build-on-first-call, get_config/from_config, weight lists in
creation order with get_weights/set_weights, and the `register_keras_serializable`
registry used by RandomChance / RandomChoice (image_augmentations.py:534-545,588-604).
Tensors are torch tensors; this module does no arithmetic.
"""
import re

import numpy as np
import torch

_REGISTRY = {}
_NAME_COUNTS = {}


def register_keras_serializable(package="Custom", name=None):
    def deco(cls):
        _REGISTRY[package + ">" + (name or cls.__name__)] = cls
        _REGISTRY.setdefault(name or cls.__name__, cls)
        cls._keras_package = package
        return cls
    return deco


def serialize(layer):
    """tf.keras.layers.serialize."""
    return {"class_name": getattr(layer, "_keras_package", "Custom") + ">" + type(layer).__name__, "config": layer.get_config()}


def deserialize(config):
    """tf.keras.layers.deserialize."""
    cls = _REGISTRY.get(config["class_name"]) or _REGISTRY.get(config["class_name"].split(">")[-1])
    if cls is None:
        raise ValueError("Unknown layer: " + str(config["class_name"]))
    return cls.from_config(dict(config["config"]))


def _snake(name):
    s = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    return re.sub("([a-z])([A-Z])", r"\1_\2", s).lower()


def reset_name_counts():
    _NAME_COUNTS.clear()


def default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class Variable:
    """A named weight; `value` is a torch tensor that the owning layer (or the training
    engine, which may re-point it at a slice of its flat buffers) reads at call time.  On the GPU a trainable variable is an
    autograd leaf (requires_grad): the stand-alone layers' Functions (layers/autograd.py) leave its gradient in `value.grad`."""

    def __init__(self, name, value, trainable=True):
        self.name = name
        self.value = value
        self.trainable = bool(trainable)
        if self.trainable and value.is_cuda and value.is_floating_point():
            self.value.requires_grad_(True)

    @property
    def grad(self):
        return self.value.grad

    def zero_grad(self):
        self.value.grad = None

    @property
    def shape(self):
        return tuple(self.value.shape)

    def numpy(self):
        return self.value.detach().float().cpu().numpy()

    def assign(self, array):
        array = np.asarray(array)
        if tuple(array.shape) != self.shape:
            raise ValueError("Layer weight shape %s not compatible with provided weight shape %s" % (self.shape, tuple(array.shape)))
        with torch.no_grad():
            self.value.copy_(torch.as_tensor(array, dtype=self.value.dtype))


class InputSpec:
    def __init__(self, ndim=None, dtype=None):
        self.ndim = ndim
        self.dtype = dtype


class Layer:
    def __init__(self, name=None, dtype=None, trainable=True, **kwargs):
        if kwargs:
            raise TypeError("Keyword argument not understood: " + ", ".join(kwargs))
        if name is None:
            base = _snake(type(self).__name__)
            n = _NAME_COUNTS.get(base, 0)
            _NAME_COUNTS[base] = n + 1
            name = base if n == 0 else "%s_%d" % (base, n)
        self.name = name
        self.trainable = trainable
        self._dtype = dtype or "float32"
        self.built = False
        self._weights = []
        self._version = 0  # bumped whenever weights change (operand caches key on it)
        if not hasattr(self, "input_spec"):
            self.input_spec = None

    # ---- weights
    def add_weight(self, name, shape, initializer=None, dtype=None):
        from . import initializers
        value = initializers.get(initializer)(tuple(int(s) for s in shape))
        var = Variable(self.name + "/" + name + ":0", torch.as_tensor(value, dtype=torch.float32).to(default_device()), trainable=self.trainable)
        self._weights.append(var)
        return var

    def _sublayers(self):
        return []

    @property
    def weights(self):
        out = list(self._weights)
        for sub in self._sublayers():
            out.extend(sub.weights)
        return out

    trainable_weights = weights

    def named_weights(self, scope=""):
        """[(name as Keras scopes it - 'encoder/encoder_layer/dense1/kernel:0' - , Variable)] in the order of `weights`."""
        out = [(scope + v.name, v) for v in self._weights]
        for sub in self._sublayers():
            out.extend(sub.named_weights(scope + self.name + "/"))
        return out

    def get_weights(self):
        return [w.numpy() for w in self.weights]

    def set_weights(self, weights):
        ws = self.weights
        if len(ws) != len(weights):
            raise ValueError('You called `set_weights(weights)` on layer "%s" with a weight list of length %d, but the layer was '
                             "expecting %d weights." % (self.name, len(weights), len(ws)))
        for var, arr in zip(ws, weights):
            var.assign(arr)
        self._bump()

    def _bump(self):
        self._version += 1
        for sub in self._sublayers():
            sub._bump()

    def count_params(self):
        return int(sum(int(np.prod(w.shape)) for w in self.weights))

    # ---- call protocol
    def build(self, input_shape):
        self.built = True

    def _check_input_spec(self, inputs):
        spec = self.input_spec
        if spec is None or not isinstance(inputs, torch.Tensor):
            return
        if spec.ndim is not None and inputs.dim() != spec.ndim:
            raise ValueError('Input 0 of layer "%s" is incompatible with the layer: expected ndim=%d, found ndim=%d. Full shape '
                             "received: %s" % (self.name, spec.ndim, inputs.dim(), tuple(inputs.shape)))
        if spec.dtype is not None and inputs.dtype != spec.dtype:
            raise ValueError('Input 0 of layer "%s" is incompatible with the layer: expected dtype=%s, found dtype=%s'
                             % (self.name, spec.dtype, inputs.dtype))

    def __call__(self, inputs, *args, **kwargs):
        self._check_input_spec(inputs)
        if not self.built:
            if isinstance(inputs, (list, tuple)):
                shape = [tuple(t.shape) for t in inputs]
            else:
                shape = tuple(inputs.shape)
            self.build(shape)
            self.built = True
        return self.call(inputs, *args, **kwargs)

    def call(self, inputs, **kwargs):
        return inputs

    def compute_output_shape(self, input_shape):
        return input_shape

    # ---- config
    def get_config(self):
        return {"name": self.name, "trainable": self.trainable, "dtype": self._dtype}

    @classmethod
    def from_config(cls, config):
        return cls(**config)


class Sequential(Layer):
    """tf.keras.Sequential as used by the reference (patch_embeddings, AutoAugment sub-policies)."""

    def __init__(self, layers=None, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.layers = list(layers or [])

    def _sublayers(self):
        return self.layers

    def get_layer(self, name):
        for layer in self.layers:
            if layer.name == name:
                return layer
        raise ValueError("No such layer: " + name)

    def call(self, inputs, **kwargs):
        x = inputs
        for layer in self.layers:
            x = layer(x, **kwargs) if _accepts_kwargs(layer) else layer(x)
        return x

    def compute_output_shape(self, input_shape):
        for layer in self.layers:
            input_shape = layer.compute_output_shape(input_shape)
        return input_shape

    def get_config(self):
        return {"name": self.name, "layers": [serialize(layer) for layer in self.layers]}

    @classmethod
    def from_config(cls, config):
        return cls([deserialize(c) for c in config["layers"]], name=config.get("name"))


def _accepts_kwargs(layer):
    return getattr(layer, "_forward_kwargs", False)
