"""Keras core layers the chambers ViT graph is assembled from (tf.keras.layers.{Dense, LayerNormalization,
Dropout, Conv2D, Reshape} as constructed in chambers/layers/transformer.py:38-49 and
chambers/models/backbones/vision_transformer.py:235-283), executing through the HIP C ABI.

Numerics follow the build's mixed-bf16 mode: fp32 variables, bf16 MFMA operands, fp32 accumulation.  Layer
outputs are float32 torch tensors unless stated; `training=True` dropout uses a key derived from
chambers_amd.rng (seed, call counter, layer id).

Every layer is TRAINABLE: `call` goes through the torch.autograd.Function wrappers of layers/autograd.py (forward and backward
are both calls into the HIP library), so a model composed from these layers can be differentiated with `loss.backward()` and
stepped with `chambers_amd.optimizers.AdamW.apply_gradients` (the reference's layers are ordinary Keras layers: fit works on any
composition of them)."""
import torch

from .. import kernels as K
from .. import rng
from .._keras_like import Layer, register_keras_serializable
from ..activations import gelu as _gelu_fn
from . import autograd as AG

_call_counter = [0]


def _next_key(site):
    _call_counter[0] += 1
    return rng.site_key(0x5EED, _call_counter[0], site)


def _as2d(x):
    return x.reshape(-1, x.shape[-1])


def _bf16(x):
    return K.cast_bf16(x) if x.dtype in (torch.bfloat16, torch.float32) else x.to(torch.bfloat16)


@register_keras_serializable(package="Chambers")
class Dense(Layer):
    def __init__(self, units, activation=None, kernel_initializer="glorot_uniform", name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.units, self.activation, self.kernel_initializer = int(units), activation, kernel_initializer

    def build(self, input_shape):
        self.kernel = self.add_weight("kernel", (input_shape[-1], self.units), self.kernel_initializer)
        self.bias = self.add_weight("bias", (self.units,), "zeros")

    def call(self, inputs, **kwargs):
        if inputs.shape[-1] % 64:
            raise ValueError("Dense on MI355X needs in_features % 64 == 0 (got %d -> %d)" % (inputs.shape[-1], self.units))
        act = self.activation
        if act is _gelu_fn or act == "gelu":
            return AG.LinearFn.apply(inputs, self.kernel.value, self.bias.value, "gelu", False)
        if act in (None, "linear", "tanh"):                # tanh: the `feature` head (vision_transformer.py:275-278)
            return AG.LinearFn.apply(inputs, self.kernel.value, self.bias.value, None if act == "linear" else act, False)
        if act == "softmax":
            # classifier_activation of the stand-alone layer (inference form; training consumes logits through the fused loss)
            return K.softmax_rows(AG.LinearFn.apply(inputs, self.kernel.value, self.bias.value, None, False).detach())
        raise ValueError("unsupported activation %r" % (act,))

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[:-1]) + (self.units,)

    def get_config(self):
        act = self.activation
        act = "gelu" if act is _gelu_fn else act
        return dict(super().get_config(), units=self.units, activation=act, kernel_initializer=self.kernel_initializer)


@register_keras_serializable(package="Chambers")
class LayerNormalization(Layer):
    def __init__(self, epsilon=1e-3, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.epsilon = epsilon

    def build(self, input_shape):
        self.gamma = self.add_weight("gamma", (input_shape[-1],), "ones")
        self.beta = self.add_weight("beta", (input_shape[-1],), "zeros")

    def call(self, inputs, **kwargs):
        return AG.LayerNormFn.apply(inputs, self.gamma.value, self.beta.value, self.epsilon)      # bf16: the compute dtype under the mixed policy

    def get_config(self):
        return dict(super().get_config(), epsilon=self.epsilon)


@register_keras_serializable(package="Chambers")
class Dropout(Layer):
    _site = 9000

    def __init__(self, rate, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.rate = rate

    def call(self, inputs, training=None, key=None, **kwargs):
        if not training or self.rate == 0.0:
            return inputs
        key = _next_key(self._site) if key is None else key
        return AG.DropoutFn.apply(inputs, self.rate, key)

    def get_config(self):
        return dict(super().get_config(), rate=self.rate)


@register_keras_serializable(package="Chambers")
class Conv2D(Layer):
    """Only the form the reference uses: kernel_size == strides, padding 'valid' (patch embedding)."""

    def __init__(self, filters, kernel_size, strides, padding="valid", name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        if kernel_size != strides or padding != "valid":
            raise ValueError("the MI355X patch-embedding path implements Conv2D(kernel_size == strides, padding='valid')")
        self.filters, self.kernel_size, self.strides, self.padding = int(filters), int(kernel_size), int(strides), padding

    def build(self, input_shape):
        p = self.kernel_size
        self.kernel = self.add_weight("kernel", (p, p, input_shape[-1], self.filters), "glorot_uniform")
        self.bias = self.add_weight("bias", (self.filters,), "zeros")

    def call(self, inputs, **kwargs):
        p = self.kernel_size
        b, h, w, _ = inputs.shape
        # Conv2D with kernel == stride is a GEMM on the gathered patch rows (the image is data, not a differentiated input)
        patches = K.normalize_patchify(inputs, p, "tf") if inputs.dtype == torch.uint8 else K.patchify_f32(inputs.detach().to(torch.float32), p)
        out = AG.LinearFn.apply(patches, self.kernel.value.reshape(-1, self.filters), self.bias.value, None, False)
        return out.reshape(b, h // p, w // p, self.filters)

    def compute_output_shape(self, input_shape):
        p = self.kernel_size
        return (input_shape[0], input_shape[1] // p, input_shape[2] // p, self.filters)

    def get_config(self):
        return dict(super().get_config(), filters=self.filters, kernel_size=self.kernel_size, strides=self.strides, padding=self.padding)


@register_keras_serializable(package="Chambers")
class Reshape(Layer):
    def __init__(self, target_shape, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.target_shape = list(target_shape)

    def call(self, inputs, **kwargs):
        return inputs.reshape(inputs.shape[0], *self.target_shape)

    def get_config(self):
        return dict(super().get_config(), target_shape=self.target_shape)
